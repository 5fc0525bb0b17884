#!/usr/bin/env python3
"""tools/sigma_sweep.py [particles]  (GPU box) -- the scoring pass over clouds of different spread, every organisation of the
scan-match kernel.  VERDICT r03 #5: what the lattice-cell rows cost is the number of cells under the beam ends, i.e. the cloud's
spread x beam length, not the particle count.  For each sigma (x, y: sigma; heading: sigma / 8 m) the map is aged by 25 frames of
the bench drive, the cloud is replaced by a Gaussian one and pfslam_time_score_kd times the whole pass (lane order + marking /
planning + scan-match + reduce; the cell rows persist, so the repeated pass is the steady state of a stationary cloud: records are
looked at, nothing is walked) for variant 3 (cell rows), 4 (round-2 plan), 2 (plain traversal) and 0 (default choice).
Also the frame loop itself: 12 frames of pfslam_step from the wide cloud (the filter contracts it), default organisation, with the
cell rows' flags -- a cloud too wide for the table must end up SUSPENDED (round-2 plan), not wiped every frame."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
pkg = importlib.import_module("gpu-icp-slam_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pts, segs = pkg.synth.make_map_points(100000, seed=1)
tree = pkg.kd_create(pts)
scans = [pkg.synth.make_scan(segs, (0.002 * i, 0.001 * i, 0.0004 * i), seed=2000 + i) for i in range(40)]
out = {"particles": n, "sweep": []}
for sigma in (0.015, 0.05, 0.2, 0.5, 0.7, 1.0):
    row = {"sigma_m": sigma}
    for variant in (3, 4, 2, 0):
        h = pkg.PfSlam(n, kd_capacity=100000 + (1 << 18))
        h.set_map(tree); h.set_variant(variant)
        for f in range(1, 6):
            h.motion_update(f)
        for i in range(25):
            h.step(6 + i, scans[i])
        p = h.particles().copy()
        rng = np.random.RandomState(7)
        mx, my, mt = [float(np.mean(p[k])) for k in ("x", "y", "theta")]
        p["x"] = (mx + rng.normal(0, sigma, n)).astype(np.float32)
        p["y"] = (my + rng.normal(0, sigma, n)).astype(np.float32)
        p["theta"] = (mt + rng.normal(0, sigma / 8.0, n)).astype(np.float32)
        h.set_particles(p); h.set_scan(scans[25])
        h.score_kd()                       # first pass: new cells are walked
        ms = h.time_score_kd(20)
        cs = h.cell_stats()
        key = {3: "cells", 4: "plan", 2: "plain", 0: "default"}[variant]
        row[key + "_ms"] = ms
        if variant in (0, 3):
            row[key + "_cells"] = {k: cs[k] for k in ("cells", "rows", "cells_without_row", "pool_slots", "flags", "suspended")}
        if variant == 0:                   # ... and the frame loop from this cloud
            t0 = time.perf_counter()
            for i in range(12):
                h.step(31 + i, scans[26 + i])
            h.synchronize()
            cs = h.cell_stats()
            row["loop_12_frames_ms_per_frame"] = (time.perf_counter() - t0) / 12 * 1e3
            row["loop_cells_after"] = {k: cs[k] for k in ("cells", "flags", "wipes", "suspended")}
        h.close()
    best = min(row[k] for k in ("cells_ms", "plan_ms", "plain_ms"))
    row["default_over_best"] = row["default_ms"] / best
    out["sweep"].append(row)
    print(json.dumps(row), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "sigma_sweep.json"), "w"), indent=1)
