"""Pin the CPU oracle (and the product's host-side tree code) against what can be run of the
reference in this container: its own kdtree.cpp (oracle/_ref/libkdtree_ref.so), the image's
rocThrust for the RNG, and libm for the transcendentals."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib as O


def grid_points(n, seed, span=400):
    rng = np.random.RandomState(seed)
    pts = np.zeros((n, 4), np.float32)
    pts[:, 0] = rng.randint(-span, span, n).astype(np.float32) * np.float32(0.025)
    pts[:, 1] = rng.randint(-span, span, n).astype(np.float32) * np.float32(0.025)
    pts[:, 3] = rng.randint(-100, 114, n)
    return pts


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 1), (3, 2), (17, 3), (1000, 4), (5000, 5)])
def test_kd_create_matches_reference_kdtree_cpp(pkg, oracle, n, seed):
    ref = O.ref_kdtree()
    if ref is None:
        pytest.skip("oracle/_ref/libkdtree_ref.so not built (needs /root/reference)")
    assert ref.ref_node_size() == 32
    pts = grid_points(n, seed, span=60)  # heavy key ties: the case std::sort's instability matters
    want = np.zeros(n, O.NODE_DTYPE)
    ref.ref_kd_create(O.P(pts), n, O.P(want))
    got_oracle = O.kd_create(pts)
    got_product = pkg.kd_create(pts)
    assert got_oracle.tobytes() == want.tobytes()
    assert got_product.tobytes() == want.tobytes()


@pytest.mark.parametrize("n,seed,span,zmode", [(40000, 1, 300, 0), (65537, 2, 500, 1), (65537, 3, 500, 2), (65537, 4, 500, 3),
                                               (100003, 5, 800, 0), (50000, 6, 2, 0)])
def test_kd_create_large_and_tied_levels_match_reference(pkg, oracle, n, seed, span, zmode):
    """Sizes at which the product's build leaves the plain std::sort call (its sorts run on several threads above 32768 points,
    csrc/kd_host.cpp), on planar maps (every z level: all keys tied) and not.  zmode 1: two z values (real sorts on the z levels);
    2: one -0.0 among the zeros (still tied); 3: one odd point."""
    ref = O.ref_kdtree()
    if ref is None:
        pytest.skip("no _ref")
    pts = grid_points(n, seed, span=span)
    rng = np.random.RandomState(seed + 100)
    if zmode == 1:
        pts[:, 2] = rng.randint(0, 2, n)
    elif zmode == 2:
        pts[n // 2, 2] = -0.0
    elif zmode == 3:
        pts[n // 3, 2] = 1.0
    want = np.zeros(n, O.NODE_DTYPE)
    ref.ref_kd_create(O.P(pts), n, O.P(want))
    assert pkg.kd_create(pts).tobytes() == want.tobytes()
    if n <= 65537:
        assert O.kd_create(pts).tobytes() == want.tobytes()


def test_kd_create_3d_points_match_reference(pkg, oracle):
    ref = O.ref_kdtree()
    if ref is None:
        pytest.skip("no _ref")
    rng = np.random.RandomState(9)
    pts = rng.uniform(-5, 5, (777, 4)).astype(np.float32)
    want = np.zeros(len(pts), O.NODE_DTYPE)
    ref.ref_kd_create(O.P(pts), len(pts), O.P(want))
    assert O.kd_create(pts).tobytes() == want.tobytes()
    assert pkg.kd_create(pts).tobytes() == want.tobytes()


def test_kd_insert_and_balance_match_reference(pkg, oracle):
    ref = O.ref_kdtree()
    if ref is None:
        pytest.skip("no _ref")
    base = grid_points(500, 21, span=80)
    extra = grid_points(300, 22, span=80)
    extra[:, 3] = -100
    cap = 800
    trees = []
    for kind in ("ref", "oracle", "product"):
        t = np.zeros(cap, O.NODE_DTYPE)
        if kind == "ref":
            ref.ref_kd_create(O.P(base), 500, O.P(t))
            for k in range(300):
                ref.ref_kd_insert_node(O.P(extra[k]), O.P(t), 500 + k)
        elif kind == "oracle":
            O.lib().orc_kd_create(O.P(base), 500, O.P(t))
            for k in range(300):
                O.kd_insert(t, 500 + k, extra[k])
        else:
            t[:500] = pkg.kd_create(base)
            for k in range(300):
                pkg.kd_insert_node(t, 500 + k, extra[k])
        trees.append(t.copy())
    assert trees[1].tobytes() == trees[0].tobytes()
    assert trees[2].tobytes() == trees[0].tobytes()
    # Balance = rebuild from the current values (kdtree.cpp:31-40)
    a, b, c = trees[0].copy(), trees[1].copy(), trees[2].copy()
    ref.ref_kd_balance(O.P(a), cap)
    O.lib().orc_kd_balance(O.P(b), cap)
    pkg.kd_balance(c, cap)
    assert b.tobytes() == a.tobytes()
    assert c.tobytes() == a.tobytes()


def test_tree_invariants(pkg):
    pts = grid_points(3000, 31)
    t = pkg.kd_create(pts)
    n = len(t)
    assert t["parent"][0] == -1 and t["axis"][0] == 0
    for i in range(n):
        for side in ("left", "right"):
            c = t[side][i]
            if c >= 0:
                assert t["parent"][c] == i
                assert t["axis"][c] == (t["axis"][i] + 1) % 3
        if t["left"][i] >= 0:
            assert t["left"][i] == i + 1  # pre-order layout
    # every input point appears exactly once
    got = np.sort(np.stack([t["x"], t["y"], t["z"], t["w"]], 1).view("f4,f4,f4,f4").ravel())
    want = np.sort(pts.view("f4,f4,f4,f4").ravel())
    assert (got == want).all()


def test_minstd_and_uniform_match_rocthrust(oracle):
    T = O.thrust_probe()
    if T is None:
        pytest.skip("thrust probe not built")
    L = O.lib()
    for seed in (1, 12345, 2147483646, 2147483647, 0, 4000000000):
        out = np.zeros(64, np.uint32)
        T.tp_minstd(C.c_uint(seed), 64, O.P(out))
        s = seed % 2147483647 or 1
        st = C.c_uint32(s)
        mine = [L.orc_minstd_next(C.byref(st)) for _ in range(64)]
        assert mine == out.tolist()
    assert L.orc_minstd_next(C.byref(C.c_uint32(12345))) == 595905495  # SURVEY.md 8c known answer
    o = np.zeros(500, np.float32)
    T.tp_uniform(777, 0.0, 3.5, 500, O.P(o))
    st = C.c_uint32(777)
    mine = np.array([L.orc_uniform_real(C.byref(st), 0.0, 3.5) for _ in range(500)], np.float32)
    assert (mine == o).all()


def test_normal_distribution_structure_matches_rocthrust(oracle):
    """rocThrust evaluates mean + sd*S3*erfcinv(2p) with a double erfcinv (Cephes ndtri); replaying that
    with the oracle's ndtri/log restatement must reproduce its samples exactly, and the oracle's own
    (CUDA-like, float erfcinv) samples must agree to 1 ulp."""
    T = O.thrust_probe()
    if T is None:
        pytest.skip("thrust probe not built")
    L = O.lib()
    S1, S2 = np.float32(2.0 ** -31), np.float32(2.0 ** -32)
    sds = (np.float32(0.015), np.float32(0.015), np.float32(0.01))
    worst = 0
    for k in range(3000):
        seed = L.orc_engine_seed(k % 97 + 1, k, 0)
        o = np.zeros(3, np.float32)
        T.tp_normal3(seed, 0.015, 0.015, 0.01, O.P(o))
        st = C.c_uint32(seed)
        st2 = C.c_uint32(seed)
        for j, sd in enumerate(sds):
            u = L.orc_minstd_next(C.byref(st)) - 1
            S3 = np.float32(-1.4142135623730950488)
            if u > 2147483645 // 2:
                u = 2147483645 - u
                S3 = -S3
            p = np.float32(np.float32(u) * S1 + S2)
            e = -L.orc_ndtri(0.5 * float(np.float32(2) * p)) * (1 / math.sqrt(2.0))
            assert np.float32(float(np.float32(sd * S3)) * e) == o[j]
            mine = np.float32(L.orc_normal(C.byref(st2), 0.0, float(sd)))
            worst = max(worst, abs(int(mine.view(np.int32)) - int(o[j].view(np.int32))))
    assert worst <= 1


def test_transcendentals_are_correctly_rounded_and_close_to_libm(oracle):
    L = O.lib()
    m = C.CDLL("libm.so.6")
    for f in ("sinf", "cosf", "asinf"):
        getattr(m, f).restype = C.c_float
        getattr(m, f).argtypes = [C.c_float]
    x = np.random.RandomState(0).uniform(-8, 8, 20000).astype(np.float32)
    s, c = O.sincosf(x)
    assert (s == np.sin(x.astype(np.float64)).astype(np.float32)).all()
    assert (c == np.cos(x.astype(np.float64)).astype(np.float32)).all()
    gs = np.array([m.sinf(float(v)) for v in x], np.float32)
    gc = np.array([m.cosf(float(v)) for v in x], np.float32)
    ulp = lambda a, b: np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    assert ulp(s, gs).max() <= 1 and ulp(c, gc).max() <= 1
    assert (s != gs).mean() < 0.03 and (c != gc).mean() < 0.03
    xa = np.random.RandomState(2).uniform(-1, 1, 20000).astype(np.float32)
    a = np.array([L.orc_asinf(float(v)) for v in xa], np.float32)
    assert (a == np.arcsin(xa.astype(np.float64)).astype(np.float32)).all()
    for v in np.random.RandomState(3).uniform(1e-12, 50, 5000):
        assert abs(L.orc_log(float(v)) - math.log(v)) <= 4e-16 * max(1.0, abs(math.log(v)))


def test_utilhash_and_seed_known_answers(oracle):
    L = O.lib()

    def utilhash(a):
        M = 0xFFFFFFFF
        a = ((a + 0x7ed55d16) + (a << 12)) & M
        a = ((a ^ 0xc761c23c) ^ (a >> 19)) & M
        a = ((a + 0x165667b1) + (a << 5)) & M
        a = ((a + 0xd3a2646c) ^ (a << 9)) & M
        a = ((a + 0xfd7046c5) + (a << 3)) & M
        a = ((a ^ 0xb55a4f09) ^ (a >> 16)) & M
        return a

    for a in (0, 1, 12345, 0x80000001, 0xFFFFFFFF):
        assert L.orc_utilhash(a) == utilhash(a)
    for it, idx, dep in ((1, 0, 0), (7, 999, 0), (650, 3, 5), (650, 3, 5 + 512), (650, 3, 5 + 1024)):
        key = (0x80000000 | ((dep << 22) & 0xFFFFFFFF) | it) & 0xFFFFFFFF
        h = utilhash(key) ^ utilhash(idx)
        assert L.orc_engine_seed(it, idx, dep) == (h % 2147483647 or 1)
    # H5: the thread index only contributes its low 9 bits (bit 31 is forced by 1 << 31)
    assert L.orc_engine_seed(650, 3, 5) == L.orc_engine_seed(650, 3, 5 + 512) == L.orc_engine_seed(650, 3, 5 + 1024)


def test_angle_addition_sincos_equals_the_direct_form(oracle):
    """CleanLidarScan's cos / sin of rot = fl(angle + theta) (oracle/pfslam_oracle.c, orc_sincos_sum) are formed by angle addition
    in double below 1024 rad of heading and directly above: either way the result must be the correctly rounded one of the direct
    definition, orc_sincosf(rot) -- for every beam and headings of any size (they are never normalised)."""
    import ctypes as C
    L = O.lib()
    PI = np.float32(3.1415926535897932384626422832795028841971)
    rng = np.random.RandomState(5)
    x, y, s, c = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    for mag in (1.0, 50.0, 1000.0, 1024.5, 1e4, 1e6):
        for t in (rng.uniform(-1, 1, 60) * mag).astype(np.float32):
            for j in range(0, 1081, 9):
                ang = np.float32(np.float32(np.float32(-135.0) + np.float32(j) * np.float32(.25)) * PI) / np.float32(180.0)
                rot = np.float32(ang + t)
                L.orc_clean_lidar_scan(j, C.c_float(1.0), C.c_float(float(t)), C.byref(x), C.byref(y))
                L.orc_sincosf(C.c_float(float(rot)), C.byref(s), C.byref(c))
                assert (x.value, y.value) == (c.value, s.value), (j, float(t))


def test_parallel_host_sort_is_std_sort_on_tied_keys(pkg):
    """csrc/kd_host.cpp restates libstdc++'s introsort (its own pivot / heap / insertion pieces, a branch-free form of its partition,
    the recursion on several threads, tied ranges permuted by a cached gather); the permutation among tied keys IS the tree topology.
    Its start-up self-check (tied arrays, restated form vs std::sort, byte for byte) must have passed wherever that form is in use, and
    a tree far above the 32 768-point threshold, built from heavily tied points, must equal the one a child process builds with
    PFSLAM_PLAIN_SORT=1 (every sort the library's std::sort call, one thread) and the one built on ONE thread of the restated form."""
    import subprocess, sys, os
    L = pkg.binding.load()
    assert L.pfslam_kd_sort_threads() >= 1
    rng = np.random.RandomState(9)
    n = 70001
    pts = np.zeros((n, 4), np.float32)
    pts[:, 0] = rng.randint(-40, 40, n).astype(np.float32) * np.float32(0.025)   # 80 distinct x, 60 distinct y: ties everywhere
    pts[:, 1] = rng.randint(-30, 30, n).astype(np.float32) * np.float32(0.025)
    pts[:, 3] = np.arange(n, dtype=np.float32)                                  # w tells tied points apart
    pts[n // 2:, 0] = np.float32(0.5)   # half of the map is ONE wall: whole sub-ranges tied on x as well as on z
    tree = pkg.kd_create(pts)
    code = ("import importlib, numpy as np, sys; pkg = importlib.import_module('gpu-icp-slam_amd'); "
            "pts = np.load(sys.argv[1]); t = pkg.kd_create(pts); assert pkg.binding.load().pfslam_kd_parallel_sort() == 0; "
            "sys.stdout.buffer.write(t.tobytes())")
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "pf_tied_%d.npy" % os.getpid())
    np.save(path, pts)
    try:
        for env in ({"PFSLAM_PLAIN_SORT": "1"}, {"PFSLAM_SORT_THREADS": "1"}):
            out = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **env),
                                 cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, check=True).stdout
            assert out == tree.tobytes(), env
    finally:
        os.remove(path)


def test_restated_host_sort_equals_std_sort_on_many_shapes(pkg):
    """The restated introsort of csrc/kd_host.cpp (branch-free partition with its exact meeting-point rule, rank sort of the <= 16-element
    pieces, tied ranges by a cached permutation) against the library call (PFSLAM_PLAIN_SORT=1 in a child process): trees of point sets
    whose levels exercise the partition's tail (sizes around 2 x 64), sorted / reversed / organ-pipe / few-valued / all-equal keys,
    -0.0 beside +0.0, denormals, and a NaN coordinate (which must send the whole build to the library call)."""
    import subprocess, sys, os
    rng = np.random.RandomState(77)
    cases = []
    for n in (1, 2, 16, 17, 18, 33, 63, 64, 65, 127, 128, 129, 130, 131, 200, 257, 300, 401, 1000, 2049, 5000, 33000):
        for kind in range(7):
            p = np.zeros((n, 4), np.float32)
            i = np.arange(n)
            if kind == 0:
                p[:, 0] = rng.randint(-50, 50, n) * np.float32(0.05); p[:, 1] = rng.randint(-50, 50, n) * np.float32(0.05)
            elif kind == 1:
                p[:, 0] = i * np.float32(0.05); p[:, 1] = (n - i) * np.float32(0.05)                 # sorted / reversed
            elif kind == 2:
                p[:, 0] = np.minimum(i, n - i) * np.float32(0.05); p[:, 1] = (i % 3) * np.float32(0.05)  # organ pipe / three values
            elif kind == 3:
                p[:, 0] = np.float32(1.25); p[:, 1] = rng.randint(0, 2, n) * np.float32(0.05)       # one wall
            elif kind == 4:
                p[:, 0] = rng.uniform(-3, 3, n); p[:, 1] = rng.uniform(-3, 3, n); p[:, 2] = rng.randint(0, 3, n)  # untied, 3-D
            elif kind == 5:
                p[:, 0] = np.where(rng.randint(0, 2, n) == 0, np.float32(0.0), np.float32(-0.0)); p[:, 1] = rng.randint(0, 4, n) * np.float32(1e-41)
            else:
                p[:, 0] = rng.randint(-5, 5, n); p[:, 1] = rng.randint(-5, 5, n)
                if n > 2: p[n // 2, 1] = np.nan
            p[:, 3] = i
            cases.append(p)
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "pf_sortshapes_%d.npz" % os.getpid())
    np.savez(path, *cases)
    code = ("import importlib, numpy as np, sys, hashlib; pkg = importlib.import_module('gpu-icp-slam_amd'); z = np.load(sys.argv[1]); "
            "print(' '.join(hashlib.md5(pkg.kd_create(z['arr_%d' % k]).tobytes()).hexdigest() for k in range(len(z.files))))")
    try:
        out = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, PFSLAM_PLAIN_SORT="1"),
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, check=True, text=True).stdout.split()
    finally:
        os.remove(path)
    import hashlib
    assert len(out) == len(cases)
    for k, p in enumerate(cases):
        assert hashlib.md5(pkg.kd_create(p).tobytes()).hexdigest() == out[k], (k, len(p), k % 7)
