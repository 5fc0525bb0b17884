#!/usr/bin/env python3
"""Replay a case tests/fuzz_step.py saved (gpurun_out/fuzz_case.npz) against the oracle, frame by frame; prints the first diverging frame.
Environment (PFSLAM_VARIANT, PFSLAM_SERIAL, PFSLAM_GATES, capacities ...) is taken as given.  usage: replay_case.py case.npz [repeats]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
pkg = importlib.import_module("gpu-icp-slam_amd")
d = np.load(sys.argv[1], allow_pickle=True)
desc, scans, drift = d["desc"][0], d["scans"], bool(d["drift"])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.int32)
bad = 0
for rep in range(reps):
    patch = O.Patch(desc["scale"], desc["scale"], desc["res"], desc["res"])
    kw = dict(n_beams=desc["nb"], kd_capacity=desc["cap"], strict_host_mirror=desc["strict"], free_upload_bug=desc["bug"], balance_period=desc["period"])
    o = O.Slam(desc["n"], patch=patch, **kw)
    h = pkg.PfSlam(desc["n"], map_scale=(desc["scale"],) * 2, map_res=(desc["res"],) * 2, **kw)
    h.set_lag(desc["lag"])
    first = None
    for f, scan in enumerate(scans, start=1):
        if drift and f == 2:
            p = O.make_particles(desc["n"], desc["scale"] / 2 - 0.3, -desc["scale"] / 2 + 0.2, 1.0)
            o.set_particles(p); h.set_particles(p)
        o.step(f, scan); h.step(f, scan)
        if f % desc["stride"] == 0 or f == len(scans):
            h.synchronize()
            to, tg = o.trace(), h.trace()
            if tg != to or not (bits(h.pose) == bits(o.pose)).all():
                first = (f, tg, to); break
    chk = h.check_cells() if hasattr(h, "check_cells") else None
    print("rep %d: %s  cells %s" % (rep, "ok" if first is None else "DIVERGED at frame %d %s vs %s" % first, {k: v for k, v in (chk or {}).items() if k.startswith("v") and v} if chk else None))
    bad += first is not None
    h.close(); o.close()
print("diverged in %d of %d replays" % (bad, reps))
