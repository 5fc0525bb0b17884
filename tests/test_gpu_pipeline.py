"""The frame pipeline of pfslam_step: frames are enqueued and booked `lag` steps later; the insert of a frame's new walls
(KDTree::InsertNode, kdtree.cpp:69-105, in list order kernel.cu:1512-1517) and the resample decision (kernel.cu:474) are taken
on the device.  Nothing of that may be visible in the results: every lag gives the oracle's frames, trees and particles;
deferred errors surface at the call that books the frame."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


@pytest.mark.parametrize("lag", [0, 1, 2])
def test_every_lag_replays_the_oracle(pkg, lag):
    """26 frames from an empty map (seed scan, hundreds of inserts in the first frames, a re-balance at frame 5 and at 12,
    resamples), the trace read only every 4th frame so that frames really are in flight in between."""
    n, period = 640, 7
    _, frames = pkg.synth.corridor_sequence(26, seed=9)
    o = O.Slam(n, kd_capacity=1 << 16, balance_period=period)
    h = pkg.PfSlam(n, kd_capacity=1 << 16, balance_period=period)
    h.set_lag(lag)
    want = []
    for f, (_, scan) in enumerate(frames, start=1):
        o.step(f, scan)
        want.append((o.trace(), o.pose.copy()))
        h.step(f, scan)
        if f % 4 == 0 or f == len(frames):
            assert h.trace() == want[-1][0], (lag, f)
            assert (bits(h.pose) == bits(want[-1][1])).all(), (lag, f)
    assert sum(t["resampled"] for t, _ in want) > 0 and sum(t["n_insert"] for t, _ in want) > 500
    assert h.kd_size == o.kd_size
    assert h.map().tobytes() == o.tree().tobytes()
    got, ref = h.particles(), o.particles()
    for fld in ("x", "y", "theta", "w"):
        assert (bits(got[fld]) == bits(ref[fld])).all(), (lag, fld)
    h.close(); o.close()


def test_device_insert_of_sorted_walls_builds_the_reference_chains(pkg, small_world):
    """New walls arrive sorted by cell index, so whole runs of them fall off the same link of the tree and hang below each
    other: the device insert resolves them in list order (claim rounds in LDS).  A scan that sees a long unmapped wall from
    a map that knows nothing near it: hundreds of inserts in one frame, tree byte-identical to the host InsertNode loop."""
    tree = small_world["tree"]
    segs = np.array([[6.0, -9.0, 6.0, 9.0], [-9.0, 7.5, 9.0, 7.5], [-9.0, -8.5, 9.0, -8.5]], np.float32)  # walls the map lacks
    scan = pkg.synth.make_scan(segs, (0.0, 0.0, 0.0), seed=3)
    o = O.Slam(64, kd_capacity=len(tree) + 4096)
    h = pkg.PfSlam(64, kd_capacity=len(tree) + 4096)
    o.set_map(tree); h.set_map(tree)
    inserted = 0
    for f in (6, 7, 8):
        o.step(f, scan); h.step(f, scan)
        to, tg = o.trace(), h.trace()
        assert tg == to, (f, tg, to)
        inserted += to["n_insert"]
    assert inserted > 300
    a, b = h.map(), o.tree()
    assert a.tobytes() == b.tobytes()
    # the chains are really there: some inserted node sits more than 12 levels below the last sorted node
    depth = np.zeros(len(b), np.int32)
    for i in range(len(tree), len(b)):
        p = b["parent"][i]
        depth[i] = depth[p] + 1 if p >= len(tree) else 1
    assert depth.max() > 12
    h.close(); o.close()


def test_deferred_capacity_error_is_reported_by_the_booking_call(pkg):
    """A frame whose new walls do not fit inserts nothing and fails -- at the call that books it: with one frame in flight
    that is pfslam_synchronize (or the next step / any getter), with lag 0 the step itself."""
    _, frames = pkg.synth.corridor_sequence(3, seed=5)
    h = pkg.PfSlam(64)
    h.step(1, frames[0][1])
    seeded = h.kd_size
    h.close()
    for lag in (0, 1):
        h = pkg.PfSlam(64, kd_capacity=seeded + 2)
        h.set_lag(lag)
        h.step(1, frames[0][1])                      # seeds the map: booked at once
        assert h.kd_size == seeded
        if lag == 0:
            with pytest.raises(pkg.PfSlamError, match="kd_capacity exhausted"):
                h.step(2, frames[1][1])
        else:
            h.step(2, frames[1][1])                  # enqueued; nothing has been booked yet
            with pytest.raises(pkg.PfSlamError, match="kd_capacity exhausted"):
                h.synchronize()
        assert h.kd_size == seeded                   # nothing was inserted, the handle is still usable
        h.close()


def test_scans_of_2000_beams_insert_through_a_large_lds_window(pkg):
    """k_test_new keeps a frame's new walls in LDS: 32 bytes per beam, i.e. more than the default 48 KB dynamic limit from
    1537 beams on (raised at the first launch).  2000 beams over 360 degrees, replayed against the oracle; more than 4096 is
    refused at create."""
    nb = 2000
    with pytest.raises(pkg.PfSlamError, match="4096 beams"):
        pkg.PfSlam(8, n_beams=5000)
    rng = np.random.RandomState(4)
    o = O.Slam(96, n_beams=nb, kd_capacity=1 << 16)
    h = pkg.PfSlam(96, n_beams=nb, kd_capacity=1 << 16)
    inserted = 0
    for f in range(1, 8):
        ang = np.deg2rad(-135.0 + 0.25 * np.arange(nb))
        # a rounded room: the range varies smoothly with the LIDAR_ANGLE of kernel.cu:42 (beams past 270 degrees wrap around)
        scan = (6.0 + 1.5 * np.cos(3.0 * ang + 0.05 * f) + rng.uniform(-0.01, 0.01, nb)).astype(np.float32)
        o.step(f, scan); h.step(f, scan)
        to, tg = o.trace(), h.trace()
        assert tg == to, (f, tg, to)
        assert (bits(h.pose) == bits(o.pose)).all(), f
        inserted += to["n_insert"]
    assert inserted > 200
    assert h.map().tobytes() == o.tree().tobytes()
    h.close(); o.close()


def test_kd_and_grid_frames_alternate_without_settling(pkg):
    """pfslam_step leaves its map update running on the aux stream; a pfslam_step_grid right behind it uses the same masks,
    pose, counts and sums (the host layer's particleFilterPC switches between the two loops).  No getter in between: the
    frames stay in flight.  Trees, grid, particles and poses must equal the oracle's, which runs the same alternation."""
    assert pkg.device_count() > 0
    n, nframes = 3000, 26
    segs, frames = pkg.synth.corridor_sequence(nframes, seed=11)
    o = O.Slam(n, kd_capacity=1 << 16)
    h = pkg.PfSlam(n, kd_capacity=1 << 16)
    for f, (_, scan) in enumerate(frames, start=1):
        grid_frame = f > 2 and f % 3 != 0           # KD, KD, then two grid frames after every KD frame
        (o.step_grid if grid_frame else o.step)(f, scan)
        (h.step_grid if grid_frame else h.step)(f, scan)
        if f % 9 == 0:                              # an occasional look: books the frames in flight
            assert (bits(h.pose) == bits(o.pose)).all(), f
    assert (bits(h.pose) == bits(o.pose)).all()
    assert h.map().tobytes() == o.tree().tobytes()
    assert (h.grid() == o.grid).all()
    got, want = h.particles(), o.particles()
    for fld in ("x", "y", "theta", "w"):
        assert (bits(got[fld]) == bits(want[fld])).all(), fld
    h.close(); o.close()


def test_headings_beyond_the_angle_addition_bound_in_the_frame_loop(pkg):
    """The scan-match kernel of the cell rows runs WITHOUT the direct form of CleanLidarScan's cos / sin while the host's bound on
    |heading| is below 512 rad (csrc/pf_math.h, sincos_sum_spec<GUARD>; the bound follows set_particles, the odometry shifts and
    every frame's header).  Headings pushed beyond 1024 rad -- by a shift between two frames in flight, and by set_particles -- must
    switch the guarded kernel in: poses, maps and particles stay bit-identical to the oracle."""
    assert pkg.device_count() > 0
    n, nframes = 5000, 14
    segs, frames = pkg.synth.corridor_sequence(nframes, seed=13)
    o = O.Slam(n, kd_capacity=1 << 16)
    h = pkg.PfSlam(n, kd_capacity=1 << 16)
    h.set_variant(3)    # the cell rows whatever the cloud's spread
    for f, (_, scan) in enumerate(frames, start=1):
        if f == 6:      # between two enqueued frames: + 164 full turns, heading ~ 1030 rad, the scans still fit
            d = np.array([0.0, 0.0, 164 * 2 * np.pi], np.float32)
            o.shift_particles(d); h.shift_particles(d)
        if f == 10:     # and back below the bound through set_particles
            p = o.particles().copy()
            p["theta"] = (p["theta"] - np.float32(164 * 2 * np.pi)).astype(np.float32)
            o.set_particles(p); h.set_particles(p)
        o.step(f, scan); h.step(f, scan)
        if f in (5, 7, 9, 11, 14):
            assert (bits(h.pose) == bits(o.pose)).all(), f
    st = h.cell_stats()
    assert st["rows"] > 0      # the cell rows were in use
    assert h.map().tobytes() == o.tree().tobytes()
    got, want = h.particles(), o.particles()
    for fld in ("x", "y", "theta", "w"):
        assert (bits(got[fld]) == bits(want[fld])).all(), fld
    h.close(); o.close()


def test_cell_row_bookkeeping_is_deterministic(pkg):
    """The persistent cell rows are built by kernels on three streams (marking and walks beside the scan-match kernel, the update behind
    the insert).  Whatever the timing, the bookkeeping must come out the same: cells claimed, rows, pool slots and extensions of two
    handles stepping through the same 24 frames at 100 000 particles are identical.  (Round 4: the next frame's marking pass bumped the
    list counter before writing the entries while k_cells_update<true> was reading the list -- the number of claimed cells then differed
    from run to run, by thousands, and every parity test still passed.)"""
    assert pkg.device_count() > 0
    pts, segs = pkg.synth.make_map_points(100000, seed=1)
    tree = pkg.kd_create(pts)
    scans = [pkg.synth.make_scan(segs, (0.002 * i, 0.001 * i, 0.0004 * i), seed=2000 + i) for i in range(24)]
    seen = []
    import os
    os.environ["PFSLAM_STABLE_ORDER"] = "1"  # (inside a Hilbert cell the counting sort's lane order is atomic arrival order: canonical here)
    for rep in range(3):
        h = pkg.PfSlam(100000, kd_capacity=100000 + (1 << 18))
        os.environ.pop("PFSLAM_STABLE_ORDER", None) if rep == 2 else None
        h.set_map(tree)
        for f in range(1, 6):
            h.motion_update(f)
        for i, s in enumerate(scans):
            h.step(6 + i, s)
        st = h.cell_stats()
        seen.append((st["cells"], st["rows"], st["pool_slots"], st["extended"], st["walked_from_root"], st["claimed"], tuple(h.pose.view(np.int32))))
        assert st["flags"] == 0 and st["walked_from_root"] == st["cells"] == st["claimed"] > 20000
        h.close()
    assert seen[0] == seen[1] == seen[2], seen
