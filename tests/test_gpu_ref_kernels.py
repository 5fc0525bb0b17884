"""The REFERENCE's own kernels as a second checker (round 5).

oracle/_ref/kernel_ref.hsaco is /root/reference/src/kernel.cu compiled for gfx950, device code only, from a scratch copy made at
build time by sed (byte-order mark, the blank inside the launch chevrons) and hipify-perl (oracle/kernel_ref_wrap.cpp has the
recipe and what it does NOT cover: the transcendental functions come from ROCm's device library, not CUDA's).  These tests run
the reference's kernels on the MI355X and compare

  * the restatement (oracle/pfslam_oracle.c) -- bit for bit wherever no transcendental function is involved; and, where one is
    (CleanLidarScan's cos / sin, normal_distribution's erfcinv), bit for bit BEHIND it, by feeding the reference's own end
    points to the restatement's traversal / ray / cell code ("hybrid"), with the agreement rate of the trigonometry itself
    reported beside it;
  * the product (C-ABI, libpfslam_hip.so) against the same kernels.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import ref_kernels as R
from ref_kernels import ptr, i32, f32, boolean, ivec2, vec3, patch

pytestmark = pytest.mark.gpu
N, NB = R.PARTICLE_COUNT, R.LIDAR_SIZE


def report(line):
    """The agreement numbers of this file, printed AND appended to gpurun_out/ref_kernel_agreement.txt (copied to profiles/rNN_ref_kernel_agreement.txt)."""
    import os
    print(line)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "ref_kernel_agreement.txt"), "a") as fh:
            fh.write(line + "\n")
    except OSError:
        pass


@pytest.fixture()
def rk(pkg):
    assert pkg.device_count() > 0
    k = R.RefKernels()
    yield k
    k.close()


def aged_tree(pkg, small_world, frames=12, n=256):
    """the synthetic map after a few SLAM frames of the restatement: inserted lattice nodes, weights moved by the update passes"""
    s = O.Slam(n, kd_capacity=len(small_world["tree"]) + 40000)
    s.set_map(small_world["tree"])
    pose = np.array([0.1, -0.2, 0.3], np.float32)
    for f in range(1, frames + 1):
        pose = pose + np.array([0.02, 0.01, 0.004], np.float32)
        s.step(f, pkg.synth.make_scan(small_world["segs"], tuple(pose), seed=100 + f))
    return s.tree().copy()


def queries(tree, rng, n):
    """query points: uniform, near nodes (ties with lattice points and hyperplanes included), exactly on nodes, far outside"""
    k = rng.randint(0, len(tree), n)
    near = np.stack([tree["x"][k], tree["y"][k]], 1) + rng.choice([0.0, 0.0125, -0.0125, 0.025, 1e-3, -1e-3], (n, 2)).astype(np.float32)
    uni = rng.uniform(-22, 22, (n, 2))
    far = rng.uniform(-500, 500, (n // 8, 2))
    q = np.concatenate([near, uni, far]).astype(np.float32)
    return np.concatenate([q, np.zeros((len(q), 1), np.float32)], 1)


def ref_traverse(rk, tree, q):
    tb, t0 = rk.tree_dev(tree)
    pts = np.zeros((len(q), 4), np.float32)
    pts[:, :3] = q
    dp, dc = rk.dev(pts), rk.zeros(len(q), np.int32)
    rk.launch("findCorrespondenceIndexKD", len(q), 128, i32(len(q)), ptr(dc), ptr(dp), ptr(t0))
    return dc.get()


def test_traversal_findCorrespondenceIndexKD(pkg, rk, small_world):
    """kernel.cu:924-972 (the same loop as 1239-1276 of EvaluateParticleKD) against orc_kd_traverse and the product's pfslam_traverse,
    on the built tree and on the tree aged by inserts: identical node indices for every query"""
    rng = np.random.RandomState(5)
    for tree in (small_world["tree"], aged_tree(pkg, small_world)):
        q = queries(tree, rng, 20000)
        want = ref_traverse(rk, tree, q)
        got, _ = O.traverse_batch(tree, q)
        assert (got == want).all(), "restatement differs from the reference's traversal in %d of %d queries" % ((got != want).sum(), len(q))
        h = pkg.PfSlam(64)
        h.set_map(tree)
        prod = h.traverse(q)
        h.close()
        assert (np.asarray(prod).ravel() == want).all(), "product traversal differs from the reference kernel"


def test_traversal_root_is_best_reads_sentinel(rk):
    """H1: with the root as best node the reference reads tree[-1]; with the sentinel in front of the array the kernel stops there,
    which is the restatement's definition"""
    tree = O.kd_create(np.array([[0, 0, 0, 1], [1, 0, 0, 1], [-1, 0, 0, 1], [0, 1, 0, 1], [0, -1, 0, 1]], np.float32))
    q = np.array([[0.01, 0.01, 0], [0.3, 0.0, 0], [-0.49, 0.2, 0]], np.float32)
    want = ref_traverse(rk, tree, q)
    got, _ = O.traverse_batch(tree, q)
    assert (got == want).all()


def test_findCorrespondenceKD_values(pkg, rk, small_world):
    """kernel.cu:874-922: the matched node's value"""
    tree = aged_tree(pkg, small_world)
    q = queries(tree, np.random.RandomState(6), 4000)
    _, t0 = rk.tree_dev(tree)
    pts = np.zeros((len(q), 4), np.float32)
    pts[:, :3] = q
    dp, dc = rk.dev(pts), rk.zeros((len(q), 4), np.float32)
    rk.launch("findCorrespondenceKD", len(q), 128, i32(len(q)), ptr(dc), ptr(dp), ptr(t0))
    best, _ = O.traverse_batch(tree, q)
    want = np.stack([tree["x"][best], tree["y"][best], tree["z"][best], tree["w"][best]], 1)
    assert (dc.get().view(np.int32) == want.view(np.int32)).all()


def test_getHyperplaneDist(rk):
    rng = np.random.RandomState(8)
    n = 4096
    a = rng.normal(0, 3, (n, 4)).astype(np.float32)
    b = rng.normal(0, 3, (n, 4)).astype(np.float32)
    b[::7] = a[::7]  # ties: on the hyperplane
    axis = rng.randint(0, 3, n).astype(np.int32)
    dd, db = rk.zeros(n, np.float32), rk.zeros(n, np.int32)
    rk.launch("ref_probe_hyperplane", n, 128, ptr(rk.dev(a)), ptr(rk.dev(b)), ptr(rk.dev(axis)), ptr(dd), ptr(db), i32(n))
    pa, pb = a[np.arange(n), axis], b[np.arange(n), axis]
    assert (dd.get().view(np.int32) == np.abs(pa - pb).view(np.int32)).all()
    assert (db.get() == (pa < pb)).all()  # sortFuncX/Y/Z: strict less (kernel.cu:821-832)


def hybrid_score(rk, tree, p, scan):
    """EvaluateParticleKD with the REFERENCE's end points (its CleanLidarScan, run on the GPU) and the restatement's traversal"""
    n = len(p)
    beam = np.tile(np.arange(NB, dtype=np.int32), n)
    e = rk.clean_lidar_scan(beam, np.tile(scan, n), np.repeat(p["theta"], NB)).reshape(n, NB, 2)
    ok = (np.abs(e[:, :, 0]) < np.float32(20.0)) & (np.abs(e[:, :, 1]) < np.float32(20.0))
    wx = e[:, :, 0] + p["x"][:, None]
    wy = e[:, :, 1] + p["y"][:, None]
    q = np.stack([wx[ok], wy[ok], np.zeros(ok.sum(), np.float32)], 1)
    best, _ = O.traverse_batch(tree, q)
    w = np.zeros((n, NB), np.float32)
    w[ok] = tree["w"][best]
    fit = np.zeros(n, np.float32)
    for j in range(NB):  # the beam order of the reference's loop, one float addition each
        fit = np.where(ok[:, j], fit + w[:, j], fit)
    return fit, e, ok


def ref_score(rk, tree, p, scan):
    _, t0 = rk.tree_dev(tree)
    dp, ds, df = rk.dev(p), rk.dev(np.ascontiguousarray(scan, np.float32)), rk.zeros(N, np.float32)
    rk.launch("kernEvaluateParticlesKD", N, 128, ptr(0), ivec2(1600, 1600), patch(), ptr(dp), vec3(0, 0, 0), ptr(ds), ptr(df), ptr(t0),
              i32(len(tree)))
    return df.get()


@pytest.mark.parametrize("fractional", [False, True])
def test_kernEvaluateParticlesKD(pkg, rk, small_world, fractional):
    """kernel.cu:1198-1308, the hot loop itself, 1000 particles x 1081 beams on the aged map.
    (1) hybrid (reference end points + restated traversal + restated sum) == reference kernel, bit for bit, every particle;
    (2) the restatement's own trigonometry (fp64 specification) against ROCm's cosf / sinf: rate reported, scores compared;
    (3) the product against the reference kernel."""
    tree = aged_tree(pkg, small_world)
    if fractional:  # uploaded maps may carry any weight: the sum's order then matters
        tree["w"] = (tree["w"] * np.float32(0.37) + np.float32(0.11)).astype(np.float32)
    rng = np.random.RandomState(9)
    p = O.make_particles(N)
    p["x"] = (0.3 + rng.normal(0, 0.05, N)).astype(np.float32)
    p["y"] = (-0.1 + rng.normal(0, 0.05, N)).astype(np.float32)
    p["theta"] = (0.35 + rng.normal(0, 0.03, N)).astype(np.float32)
    p["theta"][:8] = [0.0, 3.0, -3.0, 40.0, -100.0, 1000.0, 5000.0, 1e5]  # headings are never normalised
    scan = pkg.synth.make_scan(small_world["segs"], (0.3, -0.1, 0.35), seed=21).astype(np.float32)
    scan[5], scan[6], scan[700] = 29.0, 0.0, 31.0  # out of range / degenerate beams
    want = ref_score(rk, tree, p, scan)
    hyb, e, ok = hybrid_score(rk, tree, p, scan)
    assert (hyb.view(np.int32) == want.view(np.int32)).all(), "hybrid differs from kernEvaluateParticlesKD in %d particles" % (hyb != want).sum()
    # the restatement as a whole (its own cos / sin)
    mine = O.score_kd(tree, p, scan)
    same = mine.view(np.int32) == want.view(np.int32)
    ex = np.zeros((N, NB, 2), np.float32)
    L = O.lib()
    x, y = C.c_float(), C.c_float()
    for i in range(0, N, 50):  # a sample of the end points themselves
        for j in range(NB):
            L.orc_clean_lidar_scan(j, float(scan[j]), float(p["theta"][i]), C.byref(x), C.byref(y))
            ex[i, j] = (x.value, y.value)
    smp = slice(0, N, 50)
    trig_same = (ex[smp].view(np.int32) == e[smp].view(np.int32)).all(axis=2).mean()
    ulp = np.abs(ex[smp].view(np.int32).astype(np.int64) - e[smp].view(np.int32).astype(np.int64))
    report("kernEvaluateParticlesKD (fractional weights %s): end points of the specification identical to ROCm's cosf/sinf: %.4f (max %d ulp); restatement's scores identical: %d of %d, max |diff| %.3f"
           % (fractional, trig_same, ulp[np.isfinite(e[smp])].max(), same.sum(), N, np.abs(mine - want).max()))
    assert ulp[np.abs(p["theta"][smp]) < 1e4].max() <= 2, "specified cos/sin further than 2 ulp from the device library's"
    assert same.mean() >= 0.9
    assert np.abs(mine - want).max() <= 0.02 * np.abs(want).max() + 16
    # the product
    h = pkg.PfSlam(N)
    h.set_map(tree)
    h.set_particles(p)
    h.set_scan(scan)
    got = h.score_kd()
    h.close()
    assert (got.view(np.int32) == mine.view(np.int32)).all()
    report("kernEvaluateParticlesKD (fractional weights %s): product (specification) scores identical to the reference kernel: %d of %d" % (fractional, (got.view(np.int32) == want.view(np.int32)).sum(), N))
    # the device library's trigonometry (pfslam_set_trig): the product IS the reference kernel, every particle, every bit -- the
    # headings beyond any bound included (no specification, no guard: cosf / sinf of the float sum, as the reference's text says)
    h = pkg.PfSlam(N)
    h.set_trig(1)
    h.set_map(tree)
    h.set_particles(p)
    h.set_scan(scan)
    dev = h.score_kd()
    h.close()
    bad = int((dev.view(np.int32) != want.view(np.int32)).sum())
    report("kernEvaluateParticlesKD (fractional weights %s): product (device-library trigonometry) scores differing from the reference kernel: %d of %d" % (fractional, bad, N))
    assert bad == 0


def test_traceRay(rk):
    """kernel.cu:190-240: every octant, degenerate rays, rays leaving the map, on a 48 x 48 grid, cell for cell"""
    rng = np.random.RandomState(3)
    dim, n = 48, 6000
    se = rng.randint(-10, dim + 10, (n, 4)).astype(np.int32)
    se[:200, 2:] = se[:200, :2]                      # zero length
    se[200:400, 3] = se[200:400, 1]                  # horizontal
    se[400:600, 2] = se[400:600, 0]                  # vertical
    d = rng.randint(1, 20, 200)
    se[600:800, 2], se[600:800, 3] = se[600:800, 0] + d, se[600:800, 1] + d  # diagonal: |dx| == |dy|
    masks = rk.zeros((n, dim * dim), np.uint8)
    rk.launch("ref_probe_trace_ray", n, 64, ptr(rk.dev(se)), i32(n), i32(dim), i32(dim), ptr(masks))
    want = masks.get()
    L = O.lib()
    got = np.zeros_like(want)
    for i in range(n):
        L.orc_trace_ray(int(se[i, 0]), int(se[i, 1]), int(se[i, 2]), int(se[i, 3]), dim, dim, O.P(got[i]))
    assert ((got != 0) == (want != 0)).all(), "restated traceRay differs in %d rays" % ((got != 0) != (want != 0)).any(axis=1).sum()


def roundf(v):  # C roundf on float32 values (half away from zero), exact in double
    v = v.astype(np.float64)
    return (np.sign(v) * np.floor(np.abs(v) + 0.5)).astype(np.float32)


@pytest.mark.parametrize("theta,center", [(0.3, (800, 800)), (-2.0, (811, 795)), (1.0, (30, 1580))])
def test_kernGetWalls(pkg, rk, small_world, theta, center):
    """kernel.cu:524-549 on the real 1600 x 1600 masks.  Hybrid: the reference's end points -> the restatement's rounding, ray and
    wall-cell code == the reference's masks, cell for cell; and the restatement / product as a whole beside it."""
    dim = 1600
    scan = pkg.synth.make_scan(small_world["segs"], (0.1, -0.2, theta), seed=31).astype(np.float32)
    scan[3], scan[500] = 40.0, 0.0
    fm, wm = rk.zeros(dim * dim, np.uint8), rk.zeros(dim * dim, np.uint8)
    rk.launch("kernGetWalls", NB, 128, ptr(rk.dev(scan)), ivec2(*center), f32(theta), ptr(fm), ptr(wm), ivec2(dim, dim), patch())
    want_f, want_w = fm.get() != 0, wm.get() != 0
    e = rk.clean_lidar_scan(np.arange(NB), scan, np.full(NB, theta, np.float32))
    ok = (np.abs(e[:, 0]) < np.float32(20.0)) & (np.abs(e[:, 1]) < np.float32(20.0))
    res = np.float32(0.025)
    wx = roundf(e[:, 0] / res) + np.float32(center[0])
    wy = roundf(e[:, 1] / res) + np.float32(center[1])
    hf, hw = np.zeros(dim * dim, np.uint8), np.zeros(dim * dim, np.uint8)
    L = O.lib()
    for j in np.nonzero(ok)[0]:
        L.orc_trace_ray(center[0], center[1], int(wx[j]), int(wy[j]), dim, dim, O.P(hf))
        if 0 <= wx[j] < dim and 0 <= wy[j] < dim:
            hw[int(np.float32(wx[j] * np.float32(dim)) + wy[j])] = 1
    assert ((hf != 0) == want_f).all() and ((hw != 0) == want_w).all(), "hybrid masks differ from kernGetWalls"
    of, ow = O.get_walls(scan, center[0], center[1], theta)
    report("kernGetWalls theta %.1f centre %s: restatement as a whole: free cells differing %d of %d, wall cells differing %d of %d"
           % (theta, center, ((of != 0) != want_f).sum(), want_f.sum(), ((ow != 0) != want_w).sum(), want_w.sum()))
    assert ((ow != 0) != want_w).sum() <= 0.02 * want_w.sum() + 2  # only where cos/sin round differently at a cell edge


def test_kernGetWallsKD(pkg, rk, small_world):
    """kernel.cu:974-991: the ICP targets (the comma expressions of lines 984-985 make them pose + end point, unrounded)"""
    scan = pkg.synth.make_scan(small_world["segs"], (0.1, -0.2, 0.3), seed=41).astype(np.float32)
    scan[10] = 35.0
    out = rk.zeros((NB, 4), np.float32)
    rk.launch("kernGetWallsKD", NB, 128, ptr(rk.dev(scan)), vec3(0.1, -0.2, 0.3), ptr(out), patch())
    e = rk.clean_lidar_scan(np.arange(NB), scan, np.full(NB, 0.3, np.float32))
    ok = (np.abs(e[:, 0]) < np.float32(20.0)) & (np.abs(e[:, 1]) < np.float32(20.0))
    want = np.zeros((NB, 4), np.float32)  # H2: entries of rejected beams stay as allocated (zero here)
    want[ok, 0] = np.float32(0.1) + e[ok, 0]
    want[ok, 1] = np.float32(-0.2) + e[ok, 1]
    want[ok, 3] = 4.0
    assert (out.get().view(np.int32) == want.view(np.int32)).all()


def test_rng_hash_engine_uniform(rk):
    """utilhash, makeSeededRandomEngine, the engine's raw outputs and uniform_real_distribution (kernel.cu:89-102, thrust): bit for bit"""
    rng = np.random.RandomState(4)
    n = 5000
    it = rng.randint(0, 100000, n).astype(np.int32)
    ix = rng.randint(0, 2000000, n).astype(np.int32)
    dp = rng.randint(0, 1000, n).astype(np.int32)
    it[:4], ix[:4], dp[:4] = [0, 1, 7, 700], [0, 0, 5, 1081], [0, 0, 999, 3]
    out, outf = rk.zeros((n, 4), np.uint32), rk.zeros((n, 4), np.float32)
    rk.launch("ref_probe_rng", n, 128, ptr(rk.dev(it)), ptr(rk.dev(ix)), ptr(rk.dev(dp)), ptr(out), ptr(outf), i32(n))
    out, outf = out.get(), outf.get()
    L = O.lib()
    nd = 0
    for k in range(n):
        assert L.orc_utilhash(int(ix[k])) == out[k, 0]
        st = C.c_uint32(L.orc_engine_seed(int(it[k]), int(ix[k]), int(dp[k])))
        st2, st3 = C.c_uint32(st.value), C.c_uint32(st.value)
        for c in range(3):
            assert L.orc_minstd_next(C.byref(st)) == out[k, 1 + c], "engine output %d of seed %d" % (c, k)
        u = [L.orc_uniform_real(C.byref(st2), 0.0, 3.5) for _ in range(2)]
        assert np.array(u, np.float32).view(np.int32).tolist() == outf[k, :2].view(np.int32).tolist()
        g = np.array([L.orc_normal(C.byref(st3), 0.0, 0.015) for _ in range(2)], np.float32)
        nd += int((g.view(np.int32) != outf[k, 2:].view(np.int32)).sum())
        assert np.abs(g - outf[k, 2:]).max() <= 4e-9 + 1e-6 * np.abs(g).max()  # erfcinv: ROCm's device library vs the specification
    report("normal_distribution draws of the specification differing in the last place from ROCm's erfcinv: %d of %d" % (nd, 2 * n))


def test_kernAddNoise(rk):
    """kernel.cu:375-397 on 1000 particles: same engine draws; the normal variates agree to the last places of erfcinv"""
    p = O.make_particles(N, 0.5, -0.25, 0.125)
    want = O.add_noise(p.copy(), 17)
    dp = rk.dev(p)
    rk.launch("kernAddNoise", N, 128, ptr(dp), i32(17))
    got = dp.get()
    for f in ("x", "y", "theta"):
        assert np.abs(got[f] - want[f]).max() <= 2e-7
    same = sum((got[f].view(np.int32) == want[f].view(np.int32)).sum() for f in ("x", "y", "theta"))
    report("kernAddNoise: dispersed coordinates of the specification identical: %d of %d" % (same, 3 * N))
    assert same >= 0.97 * 3 * N
    assert (got["w"] == want["w"]).all()


def test_kernUpdateWeights_and_copy(rk):
    """kernel.cu:297-304 (float fit, int min) and 420-427"""
    rng = np.random.RandomState(12)
    p = O.make_particles(N)
    p["w"] = rng.uniform(0.1, 2.0, N).astype(np.float32)
    fit = rng.randint(-30000, 4000, N).astype(np.float32)
    mn, c = int(fit.min()), 1.0 / float(fit.max() - fit.min())
    want = p.copy()
    O.lib().orc_update_weights_f32(O.P(want), N, O.P(fit), c, mn)
    dp = rk.dev(p)
    rk.launch("kernUpdateWeights", N, 128, i32(N), ptr(dp), ptr(rk.dev(fit)), f32(c), i32(mn), also="Pffi")
    assert (dp.get()["w"].view(np.int32) == want["w"].view(np.int32)).all()
    for sq in (False, True):
        dw = rk.zeros(N, np.float32)
        rk.launch("kernCopyWeights", N, 128, ptr(dp), ptr(dw), boolean(sq))
        w = want["w"] * want["w"] if sq else want["w"]
        assert (dw.get().view(np.int32) == w.view(np.int32)).all()


def test_kernWeightedSample(rk):
    """kernel.cu:429-444.  The kernel gathers in place (H3: a race between workgroups); one 64-lane wavefront reads before it writes,
    so a single-wave launch gives the drawn index of threads 0..63 exactly -- for many (Neff, frame) seeds; the full 1000-thread
    launch must then be consistent with the snapshot semantics up to that race (every result lies on the chain of its draw)."""
    rng = np.random.RandomState(14)
    L = O.lib()
    for trial in range(24):
        w = rng.uniform(0, 1, N).astype(np.float32) ** 6
        cdf = np.zeros(N, np.float32)
        L.orc_inclusive_scan_f32(O.P(w), N, O.P(cdf))
        neff, frame = float(rng.uniform(1, 700)), int(rng.randint(0, 20000))
        src = np.zeros(N, np.int32)
        L.orc_weighted_sample_indices(O.P(cdf), N, neff, frame, 0, N, O.P(src))
        p = O.make_particles(N)
        p["x"] = np.arange(N, dtype=np.float32)
        dp, dw = rk.dev(p), rk.dev(cdf)
        rk.launch("kernWeightedSample", 64, 64, ptr(dp), ptr(dw), f32(cdf[-1]), f32(neff), i32(frame))
        got = dp.get()
        assert (got["x"][:64].astype(np.int32) == src[:64]).all(), "drawn indices differ (Neff %.2f, frame %d)" % (neff, frame)
        assert (got["w"][:64] == 1.0).all() and (got["x"][64:] == p["x"][64:]).all()
        if trial < 4:
            dp2 = rk.dev(p)
            rk.launch("kernWeightedSample", N, 128, ptr(dp2), ptr(dw), f32(cdf[-1]), f32(neff), i32(frame))
            g2 = dp2.get()["x"].astype(np.int32)
            for i in range(N):
                j, chain = src[i], set()
                while j not in chain:
                    chain.add(j)
                    j = src[j]
                assert g2[i] in chain
            report("kernWeightedSample full launch: %d of %d results equal the snapshot semantics (the rest read an already overwritten slot)" % ((g2 == src).sum(), N))


def test_kernUpdateMapKD_and_TestCorrespondance(pkg, rk, small_world):
    """kernel.cu:1350-1380 with distinct target nodes (the reference's read-modify-write is not atomic: H4)"""
    tree = aged_tree(pkg, small_world)
    rng = np.random.RandomState(15)
    n = 3000
    idx = rng.permutation(len(tree))[:n].astype(np.int32)
    pts = np.zeros((n, 4), np.float32)
    off = rng.choice([0.0, 0.0125, 0.02, 0.0353, 0.0354, 0.05], (n, 2)).astype(np.float32)
    pts[:, 0], pts[:, 1] = tree["x"][idx] + off[:, 0], tree["y"][idx] + off[:, 1]
    tree["w"][idx[:50]] = 112.0  # the clamp at +-113
    tree["w"][idx[50:100]] = -113.0
    for val in (-1, 4):
        tb, t0 = rk.tree_dev(tree)
        rk.launch("kernUpdateMapKD", n, 128, i32(n), ptr(t0), ptr(rk.dev(pts)), ptr(rk.dev(idx)), i32(val), patch())
        want = tree.copy()
        pa = O.default_patch()
        O.lib().orc_update_map_kd(O.P(want), O.P(pts), O.P(idx), n, val, C.byref(pa))
        assert (tb.get()[1:]["w"].view(np.int32) == want["w"].view(np.int32)).all()
    _, t0 = rk.tree_dev(tree)
    dd = rk.zeros(n, np.uint8)
    rk.launch("kernTestCorrespondance", n, 128, i32(n), ptr(t0), ptr(rk.dev(pts)), ptr(rk.dev(idx)), ptr(dd), patch())
    create = np.zeros(n, np.uint8)
    pa = O.default_patch()
    O.lib().orc_test_correspondence(O.P(tree), O.P(pts), O.P(idx), n, O.P(create), C.byref(pa))
    assert ((dd.get() != 0) == (create != 0)).all()


def test_grid_kernels(pkg, rk, small_world):
    """2-D path: kernEvaluateParticles (kernel.cu:257-284; hybrid through the reference's end points) and kernUpdateMap (513-522)"""
    dim = 1600
    rng = np.random.RandomState(16)
    grid = rng.randint(-100, 100, dim * dim).astype(np.int8)
    p = O.make_particles(N)
    p["x"] = rng.normal(0.2, 0.05, N).astype(np.float32)
    p["y"] = rng.normal(-0.1, 0.05, N).astype(np.float32)
    p["theta"] = rng.normal(0.3, 0.02, N).astype(np.float32)
    p["x"][0], p["y"][1] = 19.5, -19.9  # end points leaving the grid
    scan = pkg.synth.make_scan(small_world["segs"], (0.2, -0.1, 0.3), seed=51).astype(np.float32)
    fit = rk.zeros(N, np.int32)
    rk.launch("kernEvaluateParticles", N, 128, ptr(rk.dev(grid)), ivec2(dim, dim), patch(), ptr(rk.dev(p)), vec3(0, 0, 0), ptr(rk.dev(scan)), ptr(fit))
    want = fit.get()
    e = rk.clean_lidar_scan(np.tile(np.arange(NB, dtype=np.int32), N), np.tile(scan, N), np.repeat(p["theta"], NB)).reshape(N, NB, 2)
    res, half = np.float32(0.025), np.float32(0.5) * np.float32(40.0) / np.float32(0.025)
    wx = roundf(half + (e[:, :, 0] + p["x"][:, None]) / res)
    wy = roundf(half + (e[:, :, 1] + p["y"][:, None]) / res)
    ok = (wx >= 0) & (wx < dim) & (wy >= 0) & (wy < dim)
    idx = np.where(ok, wx.astype(np.int64) * dim + wy.astype(np.int64), 0)
    hyb = np.where(ok, grid[idx].astype(np.int32), 0).sum(axis=1).astype(np.int32)
    assert (hyb == want).all(), "hybrid grid score differs from kernEvaluateParticles in %d particles" % (hyb != want).sum()
    mine = np.zeros(N, np.int32)
    pa = O.default_patch()
    O.lib().orc_score_grid(O.P(grid), dim, dim, C.byref(pa), O.P(p), N, O.P(scan), NB, O.P(mine))
    report("kernEvaluateParticles: grid scores of the restatement identical: %d of %d (max |diff| %d)" % ((mine == want).sum(), N, np.abs(mine - want).max()))
    h = pkg.PfSlam(N)
    h.set_trig(1)
    h.set_grid(grid.reshape(dim, dim)); h.set_particles(p); h.set_scan(scan)
    dev = h.score_grid()
    h.close()
    report("kernEvaluateParticles: product (device-library trigonometry) grid scores differing from the reference kernel: %d of %d" % ((dev != want).sum(), N))
    assert (dev == want).all()
    assert (mine == want).mean() >= 0.8 and np.abs(mine - want).max() <= 400
    mask = (rng.uniform(0, 1, dim * dim) < 0.3).astype(np.uint8)
    for val in (-1, 4):
        dg = rk.dev(grid)
        rk.launch("kernUpdateMap", dim * dim, 128, i32(dim * dim), ptr(dg), ptr(rk.dev(mask)), i32(val))
        v = np.clip(grid.astype(np.int32) + val, -113, 113)
        assert (dg.get() == np.where(mask != 0, v, grid).astype(np.int8)).all()


def test_bench_workload_scan_match_vs_reference_kernel(pkg, rk):
    """BASELINE configs[2]'s workload as bench.py builds it (100 000 particles, 100 000-point map, a dozen SLAM frames so that the
    lattice-cell rows exist and the map has aged): the product's scan-match pass -- k_score_kd_cells, the kernel the bench line is
    about -- against the reference's own kernEvaluateParticlesKD on 12 batches of 1000 particles of the same cloud, same tree,
    same scan.  Every score must be bit-identical unless the two cos / sin implementations round an end point across a
    nearest-node tie (reported; hybrid-checked on those particles)."""
    n = 100000
    pts, segs = pkg.synth.make_map_points(100000, seed=1)
    tree0 = pkg.kd_create(pts)
    h = pkg.PfSlam(n, kd_capacity=100000 + (1 << 18))
    h.set_map(tree0)
    for f in range(1, 6):
        h.motion_update(f)
    for f in range(6, 18):
        h.step(f, pkg.synth.make_scan(segs, (0.002 * f, 0.001 * f, 0.0004 * f), seed=2000 + f))
    h.synchronize()
    p, tree = h.particles().copy(), h.map().copy()
    scan = pkg.synth.make_scan(segs, (0.002 * 18, 0.001 * 18, 0.0004 * 18), seed=2018).astype(np.float32)
    h.set_particles(p)  # (whatever the host mirror holds: both sides score exactly this cloud)
    h.set_scan(scan)
    got = h.score_kd()
    stats = h.cell_stats() if hasattr(h, "cell_stats") else {}
    h.close()
    assert len(tree) > 100000  # inserts happened
    tb, t0 = rk.tree_dev(tree)
    ds = rk.dev(scan)
    rng = np.random.RandomState(2)
    starts = [0, n - N] + list(rng.randint(0, n - N, 10))
    diff = 0
    for s in starts:
        pb = np.ascontiguousarray(p[s:s + N])
        dp, df = rk.dev(pb), rk.zeros(N, np.float32)
        rk.launch("kernEvaluateParticlesKD", N, 128, ptr(0), ivec2(1600, 1600), patch(), ptr(dp), vec3(0, 0, 0), ptr(ds), ptr(df), ptr(t0), i32(len(tree)))
        want = df.get()
        dp.free(); df.free()
        bad = np.nonzero(got[s:s + N].view(np.int32) != want.view(np.int32))[0]
        diff += len(bad)
        if len(bad):  # a tie decided differently by ROCm's cosf / sinf: the reference's end points + restated traversal must give the kernel's score
            hyb, _, _ = hybrid_score(rk, tree, pb[bad], scan)
            assert (hyb.view(np.int32) == want[bad].view(np.int32)).all()
            assert np.abs(got[s:s + N][bad] - want[bad]).max() <= 4 * 226  # a few beams changing sides between a wall node (<= +113) and a free node (>= -113)
    report("bench workload (specification): %d of %d product scores identical to the reference kernel (cell rows: %s)" % (len(starts) * N - diff, len(starts) * N, stats))
    assert diff <= 0.01 * len(starts) * N


def test_contraction_variant_reported(pkg, small_world):
    """nvcc contracts a * b + c into fma by default where its optimiser chooses to; which operations that hits is a property of
    nvcc's code generation and cannot be reproduced here.  kernel_ref_fma.hsaco is the same reference text built with clang's
    contraction on: this test REPORTS how far that moves the hot loop's results (glm::distance's dot product becomes fma chains, so
    nearest-node ties can fall the other way) -- it is information about the bound of the parity claim, not a parity check."""
    a, b = R.RefKernels(), R.RefKernels(fma=True)
    tree = aged_tree(pkg, small_world)
    rng = np.random.RandomState(9)
    p = O.make_particles(N)
    p["x"] = (0.3 + rng.normal(0, 0.05, N)).astype(np.float32)
    p["y"] = (-0.1 + rng.normal(0, 0.05, N)).astype(np.float32)
    p["theta"] = (0.35 + rng.normal(0, 0.03, N)).astype(np.float32)
    scan = pkg.synth.make_scan(small_world["segs"], (0.3, -0.1, 0.35), seed=21).astype(np.float32)
    fa, fb = ref_score(a, tree, p, scan), ref_score(b, tree, p, scan)
    q = queries(tree, np.random.RandomState(5), 20000)
    ia, ib = ref_traverse(a, tree, q), ref_traverse(b, tree, q)
    report("contraction on vs off: scores identical %d of %d (max |diff| %.1f); traversal indices identical %d of %d"
           % ((fa == fb).sum(), N, np.abs(fa - fb).max(), (ia == ib).sum(), len(q)))
    a.close(); b.close()
    assert (fa == fb).mean() > 0.5


# ---- the device library's transcendentals (pfslam_set_trig): the product against the reference's kernels with ZERO tolerance -----------------
def test_devlib_bench_workload_every_particle_equals_reference_kernel(pkg, rk):
    """BASELINE configs[2]'s workload, as above, with the handle's transcendentals switched to the device library's (what the
    reference's text compiles to here) from the first frame on: the product's scan-match pass -- k_score_kd_cells over the lattice-cell
    rows, Hilbert lane order, 16-bit beam-chunk partials -- against the reference's own kernEvaluateParticlesKD for ALL 100 000 particles
    (100 launches of PARTICLE_COUNT = 1000): not one score may differ.  This is the cell-row memoisation against the reference's
    traversal with no hybrid and no tolerance."""
    n = 100000
    pts, segs = pkg.synth.make_map_points(100000, seed=1)
    tree0 = pkg.kd_create(pts)
    h = pkg.PfSlam(n, kd_capacity=100000 + (1 << 18))
    h.set_trig(1)
    h.set_map(tree0)
    for f in range(1, 6):
        h.motion_update(f)
    for f in range(6, 18):
        h.step(f, pkg.synth.make_scan(segs, (0.002 * f, 0.001 * f, 0.0004 * f), seed=2000 + f))
    h.synchronize()
    assert h.frame_mode()["round5_frame"]
    p, tree = h.particles().copy(), h.map().copy()
    scan = pkg.synth.make_scan(segs, (0.002 * 18, 0.001 * 18, 0.0004 * 18), seed=2018).astype(np.float32)
    h.set_particles(p)
    h.set_scan(scan)
    got = h.score_kd()
    stats = h.cell_stats()
    h.close()
    assert len(tree) > 100000 and stats["rows"] > 0
    tb, t0 = rk.tree_dev(tree)
    ds = rk.dev(scan)
    diff = 0
    for s in range(0, n, N):
        pb = np.ascontiguousarray(p[s:s + N])
        dp, df = rk.dev(pb), rk.zeros(N, np.float32)
        rk.launch("kernEvaluateParticlesKD", N, 128, ptr(0), ivec2(1600, 1600), patch(), ptr(dp), vec3(0, 0, 0), ptr(ds), ptr(df), ptr(t0), i32(len(tree)))
        want = df.get()
        dp.free(); df.free()
        diff += int((got[s:s + N].view(np.int32) != want.view(np.int32)).sum())
    report("bench workload (device-library trigonometry): %d of %d product scores differ from the reference kernel (cell rows %d, cells %d)"
           % (diff, n, stats["rows"], stats["cells"]))
    assert diff == 0


@pytest.mark.parametrize("theta", [0.3, -2.0, 1.0])
def test_devlib_get_walls_masks_equal_kernGetWalls(pkg, rk, small_world, theta):
    """k_get_walls (+ the ordered cell lists) with the device library's trigonometry against the reference's kernGetWalls on the real
    1600 x 1600 masks, at the KD path's centre (kernel.cu:1408-1411): the occupancy-cell indices, wall and free, cell for cell."""
    dim, center = 1600, (800, 800)
    scan = pkg.synth.make_scan(small_world["segs"], (0.1, -0.2, theta), seed=31).astype(np.float32)
    scan[3], scan[500] = 40.0, 0.0
    fm, wm = rk.zeros(dim * dim, np.uint8), rk.zeros(dim * dim, np.uint8)
    rk.launch("kernGetWalls", NB, 128, ptr(rk.dev(scan)), ivec2(*center), f32(theta), ptr(fm), ptr(wm), ivec2(dim, dim), patch())
    want_f, want_w = np.flatnonzero(fm.get()), np.flatnonzero(wm.get())
    tree = small_world["tree"]
    h = pkg.PfSlam(64, kd_capacity=len(tree) + 4000)
    h.set_trig(1)
    h.set_map(tree); h.set_scan(scan); h.set_pose((0.1, -0.2, theta))
    h.update_map_kd()
    got_w, got_f = h.cells(0), h.cells(1)
    h.close()
    report("kernGetWalls theta %.1f: product (device-library trigonometry) wall cells %d / %d, free cells %d / %d, differing: %d"
           % (theta, len(got_w), len(want_w), len(got_f), len(want_f), len(np.setxor1d(got_w, want_w)) + len(np.setxor1d(got_f, want_f))))
    assert (got_w == want_w).all() and (got_f == want_f).all()


def test_devlib_dispersion_equals_kernAddNoise(pkg, rk):
    """k_motion with the device library's erfcinvf against the reference's kernAddNoise (thrust's normal_distribution): every coordinate
    of every particle, bit for bit -- and the frame loop's dispersion kernel (k_motion_count) gives the same particles."""
    p = O.make_particles(N, 0.5, -0.25, 0.125)
    dp = rk.dev(p)
    rk.launch("kernAddNoise", N, 128, ptr(dp), i32(17))
    want = dp.get()
    h = pkg.PfSlam(N)
    h.set_trig(1)
    h.set_particles(p)
    h.motion_update(17)
    got = h.particles().copy()
    h.close()
    bad = sum(int((got[f].view(np.int32) != want[f].view(np.int32)).sum()) for f in ("x", "y", "theta"))
    report("kernAddNoise: product (device-library erfcinv) coordinates differing from the reference kernel: %d of %d" % (bad, 3 * N))
    assert bad == 0
