#!/usr/bin/env python3
"""fuzz_replay.py FILE [repeats]: a case saved by tests/fuzz_step.py on divergence, stepped again -- product vs oracle after EVERY frame
(trace, pose, the scores of the frame through the stage API on the same state) -- so that a divergence shows where it starts."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O
pkg = importlib.import_module("gpu-icp-slam_amd")
z = np.load(sys.argv[1], allow_pickle=True)
d = z["desc"][0]; scans = z["scans"]; drift = bool(z["drift"])
print(d, "diverged at frame", int(z["frame"]), "scans", scans.shape)
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.int32)
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    patch = O.Patch(d["scale"], d["scale"], d["res"], d["res"])
    kw = dict(n_beams=d["nb"], kd_capacity=d["cap"], strict_host_mirror=d["strict"], free_upload_bug=d["bug"], balance_period=d["period"])
    o = O.Slam(d["n"], patch=patch, **kw)
    h = pkg.PfSlam(d["n"], map_scale=(d["scale"], d["scale"]), map_res=(d["res"], d["res"]), **kw)
    h.set_lag(d["lag"])
    look_every = int(os.environ.get("LOOK_EVERY", "0")) or d["stride"]
    for f, scan in enumerate(scans, start=1):
        if drift and f == 2:
            p = O.make_particles(d["n"], d["scale"] / 2 - 0.3, -d["scale"] / 2 + 0.2, 1.0)
            o.set_particles(p); h.set_particles(p)
        o.step(f, scan); h.step(f, scan)
        if f % look_every == 0 or f == len(scans):
            to, tg = o.trace(), h.trace()
            po, pg = o.particles(), h.particles()
            bad = [fld for fld in ("x", "y", "theta", "w") if not (bits(po[fld]) == bits(pg[fld])).all()]
            same_tree = o.kd_size == h.kd_size and (o.kd_size == 0 or h.map().tobytes() == o.tree().tobytes())
            print("rep", rep, "frame", f, "trace ok" if to == tg else ("TRACE %s vs %s" % (tg, to)), "particles", bad or "ok", "tree", "ok" if same_tree else "DIFF", h.cell_stats() if (bad or to != tg) else "")
            if bad and "w" in bad:
                idx = np.where(bits(po["w"]) != bits(pg["w"]))[0]
                print("   weights differ for", len(idx), "particles:", idx[:10], pg["w"][idx[:5]], po["w"][idx[:5]])
    h.close(); o.close()
