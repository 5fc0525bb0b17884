"""Differential fuzz of the whole frame loop (KD and 2-D variants) against the oracle: random configurations and adversarial
scans (NaN / Inf / zero / out-of-range beams, robot driven to the map edge, tiny capacity headroom).  Run on the GPU box:
    python tests/fuzz_step.py [seconds] [seed]
Exits non-zero at the first divergence and prints the case."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # this script lives in tests/: it drives the oracle
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O
pkg = importlib.import_module("gpu-icp-slam_amd")
os.environ.setdefault("ORC_THREADS", str(min(64, os.cpu_count() or 1)))

# python tests/fuzz_step.py --replay FILE : the case a diverging run saved (gpurun_out/fuzz_case.npz), stepped again with a look at every frame
REPLAY = None
if len(sys.argv) > 2 and sys.argv[1] == "--replay":
    REPLAY = np.load(sys.argv[2], allow_pickle=True)
    sys.argv = [sys.argv[0], "1e9", "1"]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cases = frames_total = refused = 0


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def mutate(scan, rng):
    s = scan.copy()
    k = rng.randint(0, 7)
    n = len(s)
    if k == 1: s[rng.randint(0, n, rng.randint(1, 60))] = np.nan
    if k == 2: s[rng.randint(0, n, rng.randint(1, 60))] = np.inf
    if k == 3: s[rng.randint(0, n, rng.randint(1, 200))] = 0.0
    if k == 4: s[rng.randint(0, n, rng.randint(1, 200))] = rng.uniform(20, 60)
    if k == 5: s[:] = rng.uniform(0.05, 29.0, n).astype(np.float32)
    if k == 6: s[rng.randint(0, n, rng.randint(1, 30))] = -rng.uniform(0.1, 5.0)
    return s.astype(np.float32)


while time.time() < t_end:
    n = int(rng.choice([1, 2, 50, 63, 64, 65, 300, 1000, 1025, 2049, 5000]))
    nb = int(rng.choice([1081, 1081, 1081, 721, 361, 64, 1]))
    res = float(rng.choice([0.025, 0.025, 0.05, 0.1]))
    scale = float(rng.choice([40.0, 40.0, 30.0, 20.0]))
    strict, bug = int(rng.randint(0, 2)), int(rng.randint(0, 2))
    period = int(rng.choice([100, 7, 0]))
    grid_mode = rng.rand() < 0.3
    nframes = int(rng.randint(3, 14))
    seed = int(rng.randint(0, 1 << 30))
    _, frames = pkg.synth.corridor_sequence(nframes, seed=seed % 1000)
    cap = int(rng.choice([1 << 16, 1 << 16, 3000]))
    desc = dict(n=n, nb=nb, res=res, scale=scale, strict=strict, bug=bug, period=period, grid=grid_mode, nframes=nframes, seed=seed, cap=cap)
    patch = O.Patch(scale, scale, res, res)
    try:
        o = O.Slam(n, n_beams=nb, kd_capacity=cap, strict_host_mirror=strict, free_upload_bug=bug, balance_period=period, patch=patch)
        h = pkg.PfSlam(n, n_beams=nb, kd_capacity=cap, strict_host_mirror=strict, free_upload_bug=bug, balance_period=period,
                       map_scale=(scale, scale), map_res=(res, res))
    except pkg.PfSlamError as e:
        print("create refused", desc, e); continue
    drift = rng.rand() < 0.2  # push the particle cloud towards the map edge
    lag = int(rng.choice([0, 1, 1, 2]))   # frames pfslam_step may leave in flight
    stride = int(rng.choice([1, 1, 3]))  # the product's trace is read every `stride` frames: in between, frames really are in flight
    h.set_lag(lag)
    desc.update(lag=lag, stride=stride)
    ok = True
    used = []
    for f, (_, scan) in enumerate(frames, start=1):
        scan = mutate(np.ascontiguousarray(scan[:nb]), rng)
        used.append(scan)
        if drift and f == 2:
            p = O.make_particles(n, scale / 2 - 0.3, -scale / 2 + 0.2, 1.0)
            o.set_particles(p); h.set_particles(p)
        try:
            if grid_mode:
                o.step_grid(f, scan); h.step_grid(f, scan)
            else:
                o.step(f, scan); h.step(f, scan)
            if f % stride == 0 or f == len(frames):
                h.synchronize()  # pfslam_step only enqueues the frame: a deferred error is reported by the call that books it
        except pkg.PfSlamError as e:
            # the only legitimate refusal: map capacity exhausted -- the product fails loudly where the oracle (like the
            # reference, which has no bound check at all) just stops inserting; the case ends there
            if "kd_capacity exhausted" in str(e) or "kd_capacity too small" in str(e):
                ok = False
                break
            print("UNEXPECTED ERROR", desc, f, e); sys.exit(2)
        if not (f % stride == 0 or f == len(frames)):
            frames_total += 1
            continue
        try:
            to, tg = o.trace(), h.trace()
        except pkg.PfSlamError as e:
            if "kd_capacity exhausted" in str(e):
                ok = False
                break
            print("UNEXPECTED ERROR", desc, f, e); sys.exit(2)
        same = (tg == to) or (np.isnan(to["neff"]) and np.isnan(tg["neff"]) and {k: v for k, v in tg.items() if k != "neff"} == {k: v for k, v in to.items() if k != "neff"})
        if not same or not (bits(h.pose) == bits(o.pose)).all():
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_case.npz"), desc=np.array([desc], dtype=object), scans=np.array(used), drift=drift, frame=f)
            print("DIVERGED", desc, "frame", f, tg, to, h.pose, o.pose); sys.exit(1)
        frames_total += 1
    if not ok:
        h.close(); o.close()
        refused += 1
        continue
    if grid_mode:
        if not (h.grid() == o.grid).all():
            print("DIVERGED grid", desc); sys.exit(1)
    elif o.kd_size == h.kd_size and o.kd_size > 0:
        if h.map().tobytes() != o.tree().tobytes():
            print("DIVERGED tree", desc); sys.exit(1)
    got, want = h.particles(), o.particles()
    for fld in ("x", "y", "theta", "w"):
        if not (bits(got[fld]) == bits(want[fld])).all():
            print("DIVERGED particles", fld, desc); sys.exit(1)
    h.close(); o.close()
    cases += 1
print("fuzz ok: %d cases, %d frames (%d cases ended by a loud kd_capacity refusal)" % (cases, frames_total, refused))
