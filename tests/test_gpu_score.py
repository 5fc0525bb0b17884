"""GPU parity, first slice: pf_math on the device, the reference's KD traversal, the scan-match score
(A5) and the dispersion (A3) -- HIP path through the C-ABI vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(pkg):
    assert pkg.device_count() > 0, "these tests need a GPU; the library has no CPU fallback"
    return pkg


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def test_device_math_matches_oracle_bitwise(gpu):
    h = gpu.PfSlam(64)
    L = O.lib()
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.uniform(-10, 10, 200000), rng.uniform(-1e-3, 1e-3, 1000), [0.0, -0.0, 100.5, -3000.25]]).astype(np.float32)
    sc = h.debug_math(0, x)
    s, c = O.sincosf(x)
    assert (bits(sc[:, 0]) == bits(s)).all() and (bits(sc[:, 1]) == bits(c)).all()
    # erfcinv on the p-grid the normal sampler produces, plus random (0, 1]
    y = np.concatenate([rng.uniform(2.0 ** -31, 1.0, 100000), (np.arange(1, 2000) * 2.0 ** -31)]).astype(np.float32)
    e = h.debug_math(1, y)
    want = np.array([L.orc_erfcinvf(float(v)) for v in y], np.float32)
    assert (bits(e) == bits(want)).all()
    a = rng.uniform(-1, 1, 50000).astype(np.float32)
    assert (bits(h.debug_math(2, a)) == bits(np.array([L.orc_asinf(float(v)) for v in a], np.float32))).all()
    r = rng.uniform(1e-6, 1e6, 50000).astype(np.float32)
    assert (bits(h.debug_math(3, r)) == bits(np.array([L.orc_rsqrtf(float(v)) for v in r], np.float32))).all()
    assert (bits(h.debug_math(4, r)) == bits(np.sqrt(r))).all()          # correctly rounded sqrt
    assert (bits(h.debug_math(5, x)) == bits(x / np.float32(0.025))).all()  # correctly rounded divide
    h.close()


def test_beam_end_points_for_any_heading(gpu):
    """CleanLidarScan (kernel.cu:182-187) on the device -- angle addition in double below 1024 rad of heading, the direct form above
    (headings are never normalised) -- against the oracle, bit for bit, and against the direct definition cos / sin(fl(angle + theta))."""
    import ctypes as C
    h = gpu.PfSlam(64)
    L = O.lib()
    rng = np.random.RandomState(11)
    th = np.concatenate([rng.uniform(-1, 1, 4000) * m for m in (1.0, 30.0, 1000.0, 1023.9, 1025.0, 1e4, 1e6)] + [[0.0, 1024.0, -1024.0, np.nan]]).astype(np.float32)
    got = h.debug_math(7, th)
    want = np.zeros_like(got)
    x, y = C.c_float(), C.c_float()
    for i, t in enumerate(th):
        L.orc_clean_lidar_scan(i % 1081, C.c_float(1.0), C.c_float(float(t)), C.byref(x), C.byref(y))
        want[i] = (x.value, y.value)
    assert (bits(got) == bits(want)).all()
    PI = np.float32(3.1415926535897932384626422832795028841971)
    ang = ((np.float32(-135.0) + (np.arange(len(th)) % 1081).astype(np.float32) * np.float32(.25)) * PI / np.float32(180.0)).astype(np.float32)
    s, c = O.sincosf((ang + th).astype(np.float32))
    ok = np.isfinite(th)
    assert (bits(got[ok, 0]) == bits(c[ok])).all() and (bits(got[ok, 1]) == bits(s[ok])).all()
    h.close()


def test_sub_cell_index_shortcut_agrees_with_the_exact_index(gpu):
    """The score kernel takes a query's sub-cell index from floor(q * 2 / res) unless q is within a margin of a sub-cell edge
    (csrc/kd_cells.hip.inc, sub_index_fast); the exact index compares with the float edges fl(k res) and fl(fl(k res) + res / 2).
    Both are evaluated on the device here, a few ulps either side of EVERY edge within +-105 m and on random points, and the
    exact one is re-derived in numpy from the sorted edge list."""
    h = gpu.PfSlam(64)
    res = np.float32(0.025)
    k = np.arange(-4200, 4201)
    lo = k.astype(np.float32) * res
    mid = lo + np.float32(0.5) * res
    edges = np.empty(2 * len(k), np.float32)      # edge of sub-cell S = 2 k and S = 2 k + 1
    edges[0::2], edges[1::2] = lo, mid
    assert (np.diff(edges) > 0).all()
    # +-40 ulps around every edge
    near = [edges]
    a = b = edges
    for _ in range(40):
        a = np.nextafter(a, np.float32(np.inf)); b = np.nextafter(b, np.float32(-np.inf))
        near += [a, b]
    rng = np.random.RandomState(3)
    rnd = rng.uniform(-104.9, 104.9, 1000000).astype(np.float32)
    q = np.concatenate(near + [rnd]).astype(np.float32)
    got = h.debug_math(6, q)
    want = np.searchsorted(edges, q, side="right") - 1 + 2 * int(k[0])   # max S with edge(S) <= q
    inside = (q >= edges[0]) & (q < edges[-1])
    assert (got[inside, 0] == want[inside]).all()
    safe = got[:, 1] != np.int32(-2 ** 31)
    assert (got[safe, 1] == got[safe, 0]).all()
    assert not safe[: len(edges)].any()                      # a query ON an edge is never taken by the shortcut
    unsafe_rnd = 1.0 - safe[-len(rnd):].mean()
    assert unsafe_rnd < 5e-3, unsafe_rnd                      # ... and almost every other query is
    h.close()


def test_traversal_matches_oracle_including_root_parent_case(gpu, small_world):
    tree = small_world["tree"]
    h = gpu.PfSlam(64)
    h.set_map(tree)
    rng = np.random.RandomState(1)
    q = np.zeros((60000, 3), np.float32)
    q[:30000, :2] = rng.uniform(-21, 21, (30000, 2))
    idx = rng.randint(0, len(tree), 30000)
    q[30000:, 0] = tree["x"][idx] + rng.normal(0, 0.015, 30000)
    q[30000:, 1] = tree["y"][idx] + rng.normal(0, 0.015, 30000)
    q[-1, :2] = (tree["x"][0], tree["y"][0])  # exactly the root point: best stays the root (H1)
    want, visits = O.traverse_batch(tree, q)
    got = h.traverse(q)
    assert (got == want).all()
    assert (want == 0).any(), "the H1 case (best == root, parent == -1) must be exercised"
    h.close()


def test_traversal_generic_3d_tree(gpu):
    rng = np.random.RandomState(5)
    pts = rng.uniform(-5, 5, (3000, 4)).astype(np.float32)
    tree = gpu.kd_create(pts)
    h = gpu.PfSlam(64)
    h.set_map(tree)
    q = rng.uniform(-6, 6, (20000, 3)).astype(np.float32)
    want, _ = O.traverse_batch(tree, q)
    assert (h.traverse(q) == want).all()
    h.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4097])
def test_score_kd_matches_oracle_bit_exact(gpu, small_world, n):
    tree, scan = small_world["tree"], small_world["scan"]
    p = O.make_particles(n, 0.1, -0.2, 0.3)
    O.add_noise(p, frame=7)
    want = O.score_kd(tree, p, scan)
    h = gpu.PfSlam(n)
    h.set_map(tree)
    h.set_particles(p)
    h.set_scan(scan)
    got = h.score_kd()
    assert (bits(got) == bits(want)).all()
    h.close()


def test_score_kd_edge_scans(gpu, small_world):
    tree = small_world["tree"]
    n = 300
    p = O.make_particles(n, -1.0, 2.0, -2.5)
    O.add_noise(p, frame=3)
    h = gpu.PfSlam(n)
    h.set_map(tree)
    h.set_particles(p)
    for scan in (gpu.synth.make_weird_scan(3), np.full(1081, 1000.0, np.float32), np.zeros(1081, np.float32)):
        h.set_scan(scan)
        assert (bits(h.score_kd()) == bits(O.score_kd(tree, p, scan))).all()
    # all beams out of range -> every fit is exactly +0
    h.set_scan(np.full(1081, 1000.0, np.float32))
    assert (bits(h.score_kd()) == 0).all()
    h.close()


def test_score_queries_exactly_on_cell_edges(gpu, small_world):
    """The rows of the scan-match kernel are built for a sub-cell WITHOUT its edges (corner test, csrc/kd_cells.hip.inc): a query
    exactly on a low edge must take the generic traversal.  Beam 540 looks along the heading (LIDAR_ANGLE(540) = 0), so with
    theta = 0 its end point is (x + r, y) EXACTLY: particles sit on lattice edges, on sub-cell edges and one float either side of
    them, and the beam's range is a lattice multiple, so that x + r lands on edges too."""
    tree = small_world["tree"]
    res = np.float32(0.025)
    k = np.arange(-30, 31)
    lo = k.astype(np.float32) * res
    edges = np.concatenate([lo, lo + np.float32(0.5) * res])
    vals = np.concatenate([edges, np.nextafter(edges, np.float32(np.inf)), np.nextafter(edges, np.float32(-np.inf))])
    rng = np.random.RandomState(5)
    n = 6000
    p = O.make_particles(n, 0.0, 0.0, 0.0)
    p["y"] = vals[rng.randint(0, len(vals), n)]
    p["x"] = vals[rng.randint(0, len(vals), n)]
    p["theta"] = np.where(rng.rand(n) < 0.8, np.float32(0.0), rng.uniform(-0.01, 0.01, n).astype(np.float32))
    h = gpu.PfSlam(n)
    h.set_map(tree)
    h.set_particles(p)
    h.set_variant(3)  # the organised kernel whatever the particle count
    for r540 in (np.float32(80) * res, np.float32(123) * res + np.float32(0.5) * res, np.float32(2.0)):
        scan = small_world["scan"].copy()
        scan[540] = r540
        scan[538:543:2] = r540  # neighbours: almost on the edges
        h.set_scan(scan)
        assert (bits(h.score_kd()) == bits(O.score_kd(tree, p, scan))).all()
    st = h.cell_stats()
    assert st["rows"] > 0
    h.close()
    # A wall along x with a different weight on every point: beam 540 of a particle at (0, y), theta = 0, ends at (r, y) exactly, and
    # r = fl(fl(k res) + res / 2) is the edge between two sub-cells AND the bisector of the wall points k and k + 1 -- the one place
    # where the row of the sub-cell [mid, ...) (which holds k + 1 only) would give the wrong node when k wins the tie.
    kk = np.arange(0, 400)
    wall = np.zeros((len(kk), 4), np.float32)
    wall[:, 0] = kk.astype(np.float32) * res
    wall[:, 3] = (kk % 200) - 100
    rng.shuffle(wall)
    wtree = gpu.kd_create(wall)
    n = 4800
    q = O.make_particles(n, 0.0, 0.0, 0.0)
    ys = np.array([0.0, 0.0125, -0.0125, 0.025, 1e-9, -1e-9, 0.006, -0.02], np.float32)
    q["y"] = ys[np.arange(n) % len(ys)]
    q["x"] = np.float32(0.0)
    q["theta"] = np.float32(0.0)
    h = gpu.PfSlam(n)
    h.set_map(wtree)
    h.set_particles(q)
    h.set_variant(3)
    base = np.full(1081, 1000.0, np.float32)  # every other beam out of range: the fit IS the weight of beam 540's nearest node
    for kq in range(3, 390, 7):
        lo_k = np.float32(kq) * res
        mid_k = lo_k + np.float32(0.5) * res
        # on the bisector, on a wall point, and ONE float either side of the bisector: the corners [lo+, hi-] of the corner test
        for r540 in (mid_k, lo_k, np.nextafter(mid_k, np.float32(0)), np.nextafter(mid_k, np.float32(np.inf)), np.nextafter(lo_k, np.float32(np.inf))):
            scan = base.copy()
            scan[540] = r540
            h.set_scan(scan)
            assert (bits(h.score_kd()) == bits(O.score_kd(wtree, q, scan))).all(), (kq, float(r540))
    h.close()


def test_score_kd_after_inserts_unbalanced_tree(gpu, small_world):
    """Leaf-appended nodes (KDTree::InsertNode) break the pre-order 'left = i+1' pattern."""
    base = small_world["tree"]
    cap = len(base) + 500
    t = np.zeros(cap, gpu.NODE_DTYPE)
    t[:len(base)] = base
    rng = np.random.RandomState(8)
    for k in range(500):
        p4 = np.array([np.float32(rng.randint(-700, 700)) * np.float32(0.025), np.float32(rng.randint(-700, 700)) * np.float32(0.025), 0, -100], np.float32)
        gpu.kd_insert_node(t, len(base) + k, p4)
    n = 500
    p = O.make_particles(n, 0.3, 0.1, 1.0)
    O.add_noise(p, frame=2)
    h = gpu.PfSlam(n)
    h.set_map(t)
    h.set_particles(p)
    h.set_scan(small_world["scan"])
    assert (bits(h.score_kd()) == bits(O.score_kd(t, p, small_world["scan"]))).all()
    h.close()


def test_motion_update_matches_oracle_bit_exact(gpu):
    n = 5000
    p = O.make_particles(n, 1.5, -0.5, 0.25)
    want = O.add_noise(p.copy(), frame=42, idx0=0)
    h = gpu.PfSlam(n)
    h.set_particles(p)
    h.motion_update(42)
    got = h.particles()
    for f in ("x", "y", "theta", "w"):
        assert (bits(got[f]) == bits(want[f])).all(), f
    h.close()
    # sharded: global indices key the RNG, so a shard reproduces the matching slice
    h2 = gpu.PfSlam(1000, global_offset=3000, global_n=n)
    h2.set_particles(p[3000:4000])
    h2.motion_update(42)
    got2 = h2.particles()
    assert (bits(got2["x"]) == bits(want["x"][3000:4000])).all()
    h2.close()


def test_score_at_full_bench_size_properties(gpu):
    """BASELINE size (100k-point map, 10k particles): size-independent properties instead of a full oracle run --
    (1) a sampled subset of particles matches the oracle bit-exactly, (2) the score is invariant to the
    particle order (permutation), (3) scores are integers bounded by 1081 * 113."""
    pts, segs = gpu.synth.make_map_points(100000, seed=1)
    tree = gpu.kd_create(pts)
    scan = gpu.synth.make_scan(segs, (0, 0, 0), seed=2)
    n = 10000
    p = O.make_particles(n)
    O.add_noise(p, frame=1)
    h = gpu.PfSlam(n, kd_capacity=1 << 18)
    h.set_map(tree)
    h.set_particles(p)
    h.set_scan(scan)
    fit = h.score_kd()
    sub = np.random.RandomState(0).choice(n, 64, replace=False)
    assert (bits(fit[sub]) == bits(O.score_kd(tree, p[sub], scan))).all()
    perm = np.random.RandomState(1).permutation(n)
    h.set_particles(p[perm])
    assert (bits(h.score_kd()) == bits(fit[perm])).all()
    assert (fit == np.round(fit)).all() and np.abs(fit).max() <= 1081 * 113
    h.close()


@pytest.mark.parametrize("variant", [0, 1])
def test_lane_order_variants_are_bit_identical(gpu, small_world, variant):
    """variant 0 = lanes along a Hilbert curve (counting sort over cells of the cloud), 1 = identity lane order: the order
    only decides which lane scores which particle.  Also on a tree with leaf inserts; and the
    counting instantiation of the kernel (pfslam_score_census) returns the same scores and a plausible census."""
    base = small_world["tree"]
    cap = len(base) + 400
    t = np.zeros(cap, gpu.NODE_DTYPE)
    t[:len(base)] = base
    rng = np.random.RandomState(3)
    for k in range(400):
        p4 = np.array([np.float32(rng.randint(-700, 700)) * np.float32(0.025), np.float32(rng.randint(-700, 700)) * np.float32(0.025), 0, -100], np.float32)
        gpu.kd_insert_node(t, len(base) + k, p4)
    n = 3000
    p = O.make_particles(n, 0.1, -0.2, 0.3)
    for f in (1, 2, 3):
        O.add_noise(p, frame=f)
    for tree in (base, t):
        want, visits, valid = O.score_kd(tree, p, small_world["scan"], stats=True)
        h = gpu.PfSlam(n, kd_capacity=cap)
        h.set_variant(variant)
        h.set_map(tree); h.set_particles(p); h.set_scan(small_world["scan"])
        assert (bits(h.score_kd()) == bits(want)).all()
        c = h.score_census()
        # the oracle counts every node the reference traversal reads; the product reads fewer (shared-prefix plan, skipped
        # no-op re-descents) and the census says how many
        assert 0 < c["visits"] <= visits and c["trips"] * 64 >= c["visits"] and c["tests"] * 64 >= c["test_lanes"] > 0
        assert c["redescents_noop"] <= c["redescents"]
        # 2 = the plain per-lane traversal, 3 = lattice-cell rows forced on (the synthetic map lies on the lattice), 4 = the round-2
        # shared-prefix plan forced on: same scores
        for v in (2, 3, 4):
            h.set_variant(v)
            assert (bits(h.score_kd()) == bits(want)).all(), v
            stats, cells = h.plan_stats(), h.cell_stats()
            assert (stats["rows"] > 0) == (v == 4) and (cells["rows"] > 0) == (v == 3)
            if v == 4:
                # 3000 particles: wide wave boxes, many rows straddle early -- but there is a plan, and it prunes
                assert stats["waves"] == (n + 63) // 64 and 0.0 < stats["candidates"] < stats["path_len"] and stats["path_len"] > 3
            if v == 3:
                # every marked lattice cell got the rows of its four sub-cells, a couple of candidates each
                assert 4 * cells["cells"] >= cells["rows"] > 100 and 1.0 <= cells["candidates"] <= 12.0 and cells["cells_without_row"] < 0.01 * cells["rows"]
                c3 = h.score_census()
                assert c3["prefix_trips"] >= valid                                                # every in-range query found its row
        h.set_variant(variant)
        assert (bits(h.score_kd()) == bits(want)).all()
        h.close()
