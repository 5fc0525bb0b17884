"""Edge cases of the domain on the GPU path: NaN / Inf / negative ranges, duplicate map points (collisions),
single-particle and million-particle handles, missing map, capacity limits -- against the oracle where it applies."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def test_nan_inf_negative_ranges(pkg, small_world):
    tree = small_world["tree"]
    scan = small_world["scan"].copy()
    scan[::7] = np.nan
    scan[3::11] = np.inf
    scan[5::13] = -np.inf
    scan[2::17] = -3.5          # a negative range is a valid number for the reference: it points backwards
    n = 200
    p = O.make_particles(n, 0.2, 0.1, -0.7)
    O.add_noise(p, 4)
    h = pkg.PfSlam(n, kd_capacity=len(tree) + 4000)
    h.set_map(tree); h.set_particles(p); h.set_scan(scan)
    assert (bits(h.score_kd()) == bits(O.score_kd(tree, p, scan))).all()
    # the same scan through the map update: rejected beams mark nothing
    h.set_pose((0.2, 0.1, -0.7))
    h.update_map_kd()
    fm, wm = O.get_walls(scan, 800, 800, np.float32(-0.7))
    assert (h.cells(0) == np.flatnonzero(wm)).all() and (h.cells(1) == np.flatnonzero(fm)).all()
    h.close()


def test_duplicate_map_points_and_exact_hits(pkg):
    """Collisions: the same point many times (ties in every sort), and queries that coincide with nodes (s == 0)."""
    rng = np.random.RandomState(2)
    base = np.zeros((300, 4), np.float32)
    base[:, 0] = rng.randint(-40, 40, 300).astype(np.float32) * np.float32(0.025)
    base[:, 1] = rng.randint(-40, 40, 300).astype(np.float32) * np.float32(0.025)
    pts = np.concatenate([base, base[:150], base[:50], base[:50]])
    pts[:, 3] = rng.randint(-100, 114, len(pts))
    tree = pkg.kd_create(pts)
    assert tree.tobytes() == O.kd_create(pts).tobytes()
    q = np.zeros((len(pts) + 500, 3), np.float32)
    q[:len(pts), :2] = pts[:, :2]
    q[len(pts):, :2] = rng.uniform(-1.2, 1.2, (500, 2))
    h = pkg.PfSlam(64, kd_capacity=4096)
    h.set_map(tree)
    want, _ = O.traverse_batch(tree, q)
    assert (h.traverse(q) == want).all()
    h.close()


def test_single_particle_and_missing_map(pkg, small_world):
    h = pkg.PfSlam(1)
    with pytest.raises(pkg.PfSlamError):
        h.set_scan(small_world["scan"]); h.score_kd()          # no map loaded: loud error
    h.set_map(small_world["tree"])
    p = O.make_particles(1, 0.3, 0.3, 0.3)
    h.set_particles(p)
    assert (bits(h.score_kd()) == bits(O.score_kd(small_world["tree"], p, small_world["scan"]))).all()
    best, fmin, fmax = h.measurement_update()
    assert best == 0 and fmin == fmax
    did, neff = h.resample(1)
    assert did == 0 and neff == 1.0
    h.close()
    with pytest.raises(pkg.PfSlamError):
        pkg.PfSlam(0)
    with pytest.raises(pkg.PfSlamError):
        big = np.zeros(100, pkg.NODE_DTYPE)
        big["left"] = big["right"] = big["parent"] = -1
        g = pkg.PfSlam(8, kd_capacity=10)
        g.set_map(big)                                          # larger than kd_capacity


def test_full_map_is_a_loud_error_and_refused_maps_leave_the_handle_intact(pkg, small_world):
    """kd_capacity exhausted by the inserts of a frame -> pfslam_step fails with a message (the reference appends into a
    static array without any check, kernel.cu:79, 1515); a map that fails validation does not replace the loaded one."""
    segs, frames = pkg.synth.corridor_sequence(8, seed=5)
    h = pkg.PfSlam(64)
    h.step(1, frames[0][1])
    seeded = h.kd_size
    h.close()
    h = pkg.PfSlam(64, kd_capacity=seeded + 3)        # room for the first scan and three more walls
    with pytest.raises(pkg.PfSlamError, match="kd_capacity exhausted"):
        for f, (_, scan) in enumerate(frames, start=1):
            h.step(f, scan)
    h.close()
    tree, scan = small_world["tree"], small_world["scan"]
    h = pkg.PfSlam(32, kd_capacity=len(tree) + 8)
    h.set_map(tree)
    p = O.make_particles(32, 0.1, 0.1, 0.1)
    h.set_particles(p); h.set_scan(scan)
    before = h.score_kd()
    bad = tree.copy()
    bad["left"][len(bad) // 2] = len(bad) + 5          # out-of-range link
    with pytest.raises(pkg.PfSlamError):
        h.set_map(bad)
    assert h.kd_size == len(tree) and h.map().tobytes() == tree.tobytes()
    assert (bits(h.score_kd()) == bits(before)).all()
    with pytest.raises(pkg.PfSlamError):
        pkg.PfSlam(8, kd_capacity=1 << 27)            # beyond the 32-bit byte offsets of the map records
    h.close()


def test_one_million_particles(pkg, small_world):
    """Maximum bench size: 1 M particles in one handle.  Sampled parity + permutation invariance + exact Neff path."""
    tree, scan = small_world["tree"], small_world["scan"]
    n = 1 << 20
    p = O.make_particles(n, 0.1, -0.2, 0.3)
    rng = np.random.RandomState(0)
    p["x"] += rng.normal(0, 0.03, n).astype(np.float32)
    p["y"] += rng.normal(0, 0.03, n).astype(np.float32)
    p["theta"] += rng.normal(0, 0.02, n).astype(np.float32)
    h = pkg.PfSlam(n)
    h.set_map(tree); h.set_particles(p); h.set_scan(scan)
    fit = h.score_kd()
    sub = rng.choice(n, 96, replace=False)
    assert (bits(fit[sub]) == bits(O.score_kd(tree, p[sub], scan))).all()
    best, fmin, fmax = h.measurement_update()
    assert best == int(np.argmax(fit)) and fmin == fit.min() and fmax == fit.max()
    did, neff = h.resample(3)
    w = h.particles()["w"] if not did else None
    # canonical Neff over 256 tiles equals the oracle's
    import ctypes as C
    pw = p.copy()
    O.lib().orc_update_weights_f32(O.P(pw), n, O.P(fit), float(np.float32(1) / (np.float32(fmax) - np.float32(fmin))), int(fmin))
    ne = C.c_float()
    did_o = O.lib().orc_resample(O.P(pw), n, 3, C.byref(ne), None)
    assert did == did_o and np.float32(neff) == np.float32(ne.value)
    if did:
        got = h.particles()
        assert (bits(got["x"]) == bits(pw["x"])).all() and (bits(got["theta"]) == bits(pw["theta"])).all()
    h.close()


def test_3d_queries_on_a_planar_map(pkg, small_world):
    """pfslam_traverse with z != 0 queries against a planar map takes the generic kernel, which has to undo the
    planar link trick of the device layout."""
    tree = small_world["tree"]
    rng = np.random.RandomState(11)
    q = rng.uniform(-15, 15, (5000, 3)).astype(np.float32)
    q[:, 2] = rng.uniform(-0.5, 0.5, 5000)
    h = pkg.PfSlam(64, kd_capacity=len(tree) + 64)
    h.set_map(tree)
    want, _ = O.traverse_batch(tree, q)
    assert (h.traverse(q) == want).all()
    h.close()


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_maps_scans_particles(pkg, seed):
    """Random small worlds: degenerate maps (collinear points, heavy duplicates, tiny trees), random poses far outside the
    map, random scans.  Traversal and score must equal the oracle bit for bit with integer weights and with arbitrary float
    weights alike."""
    rng = np.random.RandomState(100 + seed)
    kind = seed % 3
    n_pts = int(rng.choice([1, 2, 3, 7, 64, 500, 3000]))
    pts = np.zeros((n_pts, 4), np.float32)
    if kind == 0:      # collinear
        pts[:, 0] = rng.randint(-700, 700, n_pts).astype(np.float32) * np.float32(0.025)
    elif kind == 1:    # few distinct values, many duplicates
        pts[:, 0] = rng.randint(-5, 5, n_pts).astype(np.float32) * np.float32(0.025)
        pts[:, 1] = rng.randint(-5, 5, n_pts).astype(np.float32) * np.float32(0.025)
    else:              # generic, not grid-snapped
        pts[:, :2] = rng.uniform(-20, 20, (n_pts, 2))
    pts[:, 3] = rng.randint(-113, 114, n_pts)
    tree = pkg.kd_create(pts)
    assert tree.tobytes() == O.kd_create(pts).tobytes()
    q = np.zeros((3000, 3), np.float32)
    q[:, :2] = rng.uniform(-30, 30, (3000, 2))
    n = int(rng.choice([1, 65, 700]))
    p = O.make_particles(n)
    p["x"], p["y"], p["theta"] = rng.uniform(-25, 25, n), rng.uniform(-25, 25, n), rng.uniform(-7, 7, n)
    scan = rng.uniform(0, 25, 1081).astype(np.float32)
    h = pkg.PfSlam(n, kd_capacity=max(8, n_pts))
    h.set_map(tree); h.set_particles(p); h.set_scan(scan)
    want_idx, _ = O.traverse_batch(tree, q)
    assert (h.traverse(q) == want_idx).all()
    assert (bits(h.score_kd()) == bits(O.score_kd(tree, p, scan))).all()
    # arbitrary float weights
    t2 = tree.copy()
    t2["w"] = rng.uniform(-113, 113, n_pts).astype(np.float32)
    h.set_map(t2)
    # non-integral weights: the library detects them at upload and scores in ONE beam chunk, i.e. in the reference's own
    # sequential beam order (kernEvaluateParticlesKD) -- bit-exact again, just slower
    got, want = h.score_kd(), O.score_kd(t2, p, scan)
    assert (bits(got) == bits(want)).all()
    h.close()


def test_non_default_map_and_beam_count(pkg):
    """Nothing is hard-wired to 1081 beams / 40 m / 0.025 m: a 30 x 30 m map of 0.05 m cells and 721 beams (-135..+45 deg)
    replays bit-identically, KD path and 2-D path; a non-square map is refused (the reference's x * dim.x + y cell index)."""
    import ctypes as C
    segs, frames = pkg.synth.corridor_sequence(12, seed=7)
    patch = O.Patch(30.0, 30.0, 0.05, 0.05)
    kw = dict(n_beams=721, kd_capacity=1 << 16)
    o = O.Slam(300, patch=patch, **kw)
    h = pkg.PfSlam(300, map_scale=(30.0, 30.0), map_res=(0.05, 0.05), **kw)
    for f, (_, scan) in enumerate(frames, start=1):
        scan = np.ascontiguousarray(scan[:721])
        o.step(f, scan); h.step(f, scan)
        assert h.trace() == o.trace(), f
        assert (h.pose.view(np.int32) == o.pose.view(np.int32)).all()
        assert (h.cells(0) == o.cells(0)).all() and (h.cells(1) == o.cells(1)).all()
    assert h.map().tobytes() == o.tree().tobytes() and o.kd_size > 100
    h.close(); o.close()
    o = O.Slam(300, patch=patch, **kw)
    h = pkg.PfSlam(300, map_scale=(30.0, 30.0), map_res=(0.05, 0.05), **kw)
    for f, (_, scan) in enumerate(frames, start=1):
        scan = np.ascontiguousarray(scan[:721])
        o.step_grid(f, scan); h.step_grid(f, scan)
        assert h.trace() == o.trace(), f
    assert (h.grid() == o.grid).all() and h.grid().shape == o.grid.shape
    h.close(); o.close()
    with pytest.raises(pkg.PfSlamError):
        pkg.PfSlam(8, map_scale=(40.0, 30.0))


@pytest.mark.parametrize("res,scale", [(0.03, 36.0), (0.1, 40.0)])
def test_cell_rows_at_other_resolutions(pkg, res, scale):
    """The lattice-cell rows (csrc/kd_cells.hip.inc) are exact for |k| < 2^20 cells per axis at ANY resolution (PF_LATTICE_KMAX): a
    3 cm and a 10 cm map, rows forced on at 300 particles, replay bit-identically; a map with one point beyond the range is scored
    without rows (the round-2 plan), bit-identically too."""
    segs, frames = pkg.synth.corridor_sequence(10, seed=9)
    patch = O.Patch(scale, scale, res, res)
    kw = dict(kd_capacity=1 << 16)
    o = O.Slam(300, patch=patch, **kw)
    h = pkg.PfSlam(300, map_scale=(scale, scale), map_res=(res, res), **kw)
    h.set_variant(3)
    for f, (_, scan) in enumerate(frames, start=1):
        o.step(f, scan); h.step(f, scan)
        assert h.trace() == o.trace(), f
        assert (h.pose.view(np.int32) == o.pose.view(np.int32)).all()
    assert h.map().tobytes() == o.tree().tobytes() and o.kd_size > 100
    st = h.cell_stats()
    assert st["rows"] > 100 and st["flags"] == 0
    # the same map with one point pushed beyond 2^20 cells: off the range -> no rows, same scores
    tree = o.tree().copy()
    leaf = int(np.where((tree["left"] < 0) & (tree["right"] < 0))[0][0])
    p = o.particles().copy()
    scan = frames[-1][1]
    far = tree.copy()
    far["x"][leaf] = np.float32(res) * np.float32(2 ** 20 + 5)
    h2 = pkg.PfSlam(300, map_scale=(scale, scale), map_res=(res, res), **kw)
    h2.set_variant(3)
    h2.set_map(far); h2.set_particles(p); h2.set_scan(scan)
    got = h2.score_kd()
    assert h2.cell_stats()["rows"] == 0 and h2.plan_stats()["rows"] > 0
    o2 = O.Slam(300, patch=patch, **kw)
    assert (bits(got) == bits(O.score_kd(far, p, scan))).all()
    h.close(); h2.close(); o.close(); o2.close()


def test_differential_fuzz_of_the_frame_loop():
    """20 s of tests/fuzz_step.py: random particle counts / beam counts / map geometry / parity flags / balance periods,
    adversarial scans (NaN, Inf, zero, negative, out of range), clouds at the map edge, tiny capacity headroom -- KD and
    2-D frame loops stay bit-identical to the oracle frame by frame."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_step.py"), "20", "7"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    # and once more with the shared-prefix plan of the score kernel on at EVERY particle count (by default it starts at ~6 k
    # particles): tiny waves, partly filled waves, NaN / Inf scans and clouds at the map edge all go through the planning kernel
    # (lattice-cell rows -- the SLAM step's own maps lie on the lattice --, then the round-2 shared-prefix plan)
    # ... and with the persistent cell rows forced on and their list / pool capacities cut to a few hundred entries: the overflow paths
    # (cells left without rows, wipes asked for in the frame header, suspension after the second overflow) change no result
    for extra, seed in (({}, "11"), ({"PFSLAM_VARIANT": "4"}, "12"),
                        ({"PFSLAM_VARIANT": "3", "PFSLAM_CELL_LIST_CAP": "200", "PFSLAM_CELL_POOL_CAP": "1500"}, "13")):
        env = dict(os.environ, PFSLAM_PLAN_MIN_N="1", **extra)
        out = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_step.py"), "15", seed], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
