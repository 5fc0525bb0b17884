"""2-D occupancy-grid variant of the frame loop (BASELINE configs 0 and 1): the reference's
PFMotionUpdate / PFMeasurementUpdate / PFUpdateMap / PFResample stages in the order of particleFilter().
CPU part: the oracle step against its own stage functions; GPU part: pfslam_step_grid vs the oracle step."""
import ctypes as C
import zlib

import numpy as np
import pytest

import oracle_lib as O


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def _drive(pkg, frames, seed=5):
    return pkg.synth.corridor_sequence(frames, seed=seed)[1]


def test_oracle_grid_step_first_frame_is_flat_score(pkg):
    """Frame 1 on the all -100 grid: every particle scores the same, so no weight changes, best = slot 0 and the
    pose is particle 0 after dispersion; the grid equals PFUpdateMap at that pose."""
    frames = _drive(pkg, 2)
    n = 64
    o = O.Slam(n)
    o.step_grid(1, frames[0][1])
    t = o.trace()
    assert t["best"] == 0 and t["resampled"] == 0 and t["neff"] == float(n)
    p = O.make_particles(n, 0.0, 0.0, 0.0)
    O.add_noise(p, frame=1)
    assert (bits(o.pose) == bits([p["x"][0], p["y"][0], p["theta"][0]])).all()
    got = o.particles()
    for fld in ("x", "y", "theta", "w"):
        assert (bits(got[fld]) == bits(p[fld])).all()
    g = np.full((1600, 1600), -100, np.int8)
    patch = O.default_patch()
    O.lib().orc_update_map_grid(O.P(g), 1600, 1600, C.byref(patch), O.P(o.pose), O.P(frames[0][1]), 1081)
    assert (o.grid == g).all() and (g != -100).sum() > 1000


def test_oracle_grid_step_equals_its_stages(pkg):
    """Six frames: the step is exactly motion -> score/weights -> map update -> resample of the stage functions."""
    frames = _drive(pkg, 6)
    n = 200
    o = O.Slam(n, strict_host_mirror=0)
    p = O.make_particles(n, 0.0, 0.0, 0.0)
    g = np.full((1600, 1600), -100, np.int8)
    patch = O.default_patch()
    for f, (_, scan) in enumerate(frames, start=1):
        o.step_grid(f, scan)
        O.add_noise(p, frame=f)
        fit = np.zeros(n, np.int32)
        O.lib().orc_score_grid(O.P(g), 1600, 1600, C.byref(patch), O.P(p), n, O.P(scan), 1081, O.P(fit))
        imin, imax = C.c_int(), C.c_int()
        O.lib().orc_minmax_first_i32(O.P(fit), n, C.byref(imin), C.byref(imax))
        rng = int(fit[imax.value]) - int(fit[imin.value])
        if rng > 0:
            O.lib().orc_update_weights_i32(O.P(p), n, O.P(fit), float(np.float32(1) / np.float32(rng)), int(fit[imin.value]))
        best = imax.value
        robot = np.array([p["x"][best], p["y"][best], p["theta"][best]], np.float32)
        assert o.trace()["best"] == best and (bits(o.pose) == bits(robot)).all()
        O.lib().orc_update_map_grid(O.P(g), 1600, 1600, C.byref(patch), O.P(robot), O.P(scan), 1081)
        neff = C.c_float()
        did = O.lib().orc_resample(O.P(p), n, f, C.byref(neff), None)
        assert did == o.trace()["resampled"]
    assert (o.grid == g).all()
    got = o.particles()
    for fld in ("x", "y", "theta", "w"):
        assert (bits(got[fld]) == bits(p[fld])).all()


def test_oracle_grid_step_tracks_the_drive(pkg):
    """The 2-D filter follows the synthetic drive (heading within 0.02 rad, lateral error under 0.1 m)."""
    frames = _drive(pkg, 30)
    o = O.Slam(300)
    for f, (pose, scan) in enumerate(frames, start=1):
        o.step_grid(f, scan)
    est = o.pose
    assert abs(est[2] - frames[-1][0][2]) < 0.02 and abs(est[1] - frames[-1][0][1]) < 0.1
    g = o.grid
    assert g.max() > 0 and (g == -113).sum() > 10000  # walls gaining confidence (+4 per hit from -100), free space saturated


@pytest.mark.gpu
@pytest.mark.parametrize("n,strict,nframes", [(50, 1, 16), (10000, 1, 10), (1000, 0, 12)])
def test_gpu_grid_step_replay_matches_oracle(pkg, n, strict, nframes):
    """BASELINE config 0 (50 particles) and config 1 (10 k particles): every frame's best index, resample
    decision, Neff and pose, and the final grid and particles, bit-identical to the oracle."""
    assert pkg.device_count() > 0
    frames = _drive(pkg, nframes)
    o = O.Slam(n, strict_host_mirror=strict)
    h = pkg.PfSlam(n, strict_host_mirror=strict)
    did = 0
    for f, (_, scan) in enumerate(frames, start=1):
        o.step_grid(f, scan)
        h.step_grid(f, scan)
        to, tg = o.trace(), h.trace()
        assert tg == to, (f, tg, to)
        assert (bits(h.pose) == bits(o.pose)).all(), f
        did += to["resampled"]
        if f in (1, 2, nframes):
            assert zlib.crc32(h.grid().tobytes()) == zlib.crc32(o.grid.tobytes()), f
    assert did >= 2
    got, want = h.particles(), o.particles()
    for fld in ("x", "y", "theta", "w"):
        assert (bits(got[fld]) == bits(want[fld])).all(), fld
    h.close(); o.close()


@pytest.mark.gpu
def test_gpu_grid_step_from_a_preset_grid(pkg):
    """pfslam_set_grid seeds the map; out-of-map robot cells (pose near the edge) are clipped like the reference."""
    assert pkg.device_count() > 0
    rng = np.random.RandomState(3)
    grid = rng.randint(-113, 114, (1600, 1600)).astype(np.int8)
    frames = _drive(pkg, 4)
    n = 300
    o = O.Slam(n); h = pkg.PfSlam(n)
    o.set_grid(grid); h.set_grid(grid)
    p = O.make_particles(n, 19.6, -19.7, 0.4)  # the scan reaches beyond the map on two sides
    h.set_particles(p); o.set_particles(p)
    for f, (_, scan) in enumerate(frames, start=1):
        o.step_grid(f, scan); h.step_grid(f, scan)
        assert h.trace() == o.trace()
        assert (bits(h.pose) == bits(o.pose)).all()
    assert (h.grid() == o.grid).all()
    h.close(); o.close()


def test_oracle_cpu_branch_first_frame_accounting(pkg):
    """The reference's CPU branches of the 2-D loop (kernel.cu:340-369, 578-620, 487-508; H7), first frame on the flat
    -100 grid, against an independent numpy accounting: free cells -1 once per cell, wall cells +4 once per BEAM (the
    GPU branch adds +4 once per cell), no 20 m reject but rays only to end cells inside the map, best = first slot."""
    frames = _drive(pkg, 3)
    scan = frames[0][1].copy()
    scan[10:20] = 29.0          # beyond the GPU branch's 20 m reject, but inside the 40 m map from the origin? no: 29 m > 20 m half-extent
    scan[500:510] = 19.5        # inside the map
    n = 50
    o = O.Slam(n)
    o.step_grid_cpu(1, scan)
    t = o.trace()
    assert t["best"] == 0 and t["resampled"] == 0
    p = O.make_particles(n)
    O.add_noise(p, frame=1)
    robot = np.array([p["x"][0], p["y"][0], p["theta"][0]], np.float32)
    assert (bits(o.pose) == bits(robot)).all()
    res = np.float32(0.025)
    cx = int(np.round(np.float32(800.0) + robot[0] / res + res / np.float32(2)))
    cy = int(np.round(np.float32(800.0) + robot[1] / res + res / np.float32(2)))
    hits = np.zeros((1600, 1600), np.int32)
    free = np.zeros(1600 * 1600, np.uint8)
    s, c = O.sincosf(((np.float32(-135.0) + np.arange(1081, dtype=np.float32) * np.float32(.25)) * np.float32(np.pi)) / np.float32(180.0) + robot[2])
    # np.float32(np.pi) == the reference's PI truncated to float
    traced = 0
    for j in range(1081):
        wx = np.float32(np.round(scan[j] * c[j] / res)) + np.float32(cx)
        wy = np.float32(np.round(scan[j] * s[j] / res)) + np.float32(cy)
        if 0 <= wx < 1600 and 0 <= wy < 1600:
            O.lib().orc_trace_ray(cx, cy, int(wx), int(wy), 1600, 1600, O.P(free))
            hits[int(wx), int(wy)] += 1
            traced += 1
    assert 300 < traced < 1081    # beams whose end cell is outside the map (the 29-30 m ones) trace nothing
    want = np.full((1600, 1600), -100, np.int32) - free.reshape(1600, 1600) + 4 * hits
    want = np.clip(want, -113, 113).astype(np.int8)
    assert (o.grid == want).all()
    assert hits.max() >= 2        # several beams per wall cell: where the two branches differ
    # the GPU branch on the same input: same free mask semantics inside 20 m, +4 once per cell
    g = O.Slam(n)
    g.step_grid(1, scan)
    diff = (g.grid != o.grid)
    assert diff.any() and (g.grid[hits >= 2] <= o.grid[hits >= 2]).all()
    # a few more frames run and resample (sequential in-place copies, one engine)
    did = 0
    for f in (2, 3):
        o.step_grid_cpu(f, frames[f - 1][1])
        did += o.trace()["resampled"]
        pp = o.particles()
        assert np.isfinite(pp["x"]).all() and (pp["w"] >= 0).all()
    o.close(); g.close()
