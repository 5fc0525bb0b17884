#!/usr/bin/env python3
"""Loop-trip census of the scan-match traversal (build with PFSLAM_EXTRA_FLAGS=-DPF_EXP_COUNT):
per wave-query: trips of the descent loop, mean active lanes, parent tests.  Run on the GPU box."""
import importlib, sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
pkg = importlib.import_module("gpu-icp-slam_amd")
n = 100000
pts, segs = pkg.synth.make_map_points(100000, seed=1)
h = pkg.PfSlam(n, kd_capacity=1 << 18)
h.set_map(pkg.kd_create(pts))
for f in range(1, 6):
    h.motion_update(f)
frame = 6
for k in range(25):
    scan = pkg.synth.make_scan(segs, (0.002 * k, 0.001 * k, 0.0004 * k), seed=2000 + k)
    if k in (0, 24):
        h.debug_census(reset=True)
        h.set_scan(scan); h.score_kd(fetch=False)   # extra scoring pass on the current state, census only
        c = h.debug_census(reset=True)
        wq = (n / 64.0) * 1081
        print("frame %d: descent-loop trips per wave-query %.1f, mean active lanes %.1f (%.0f%%), lane visits per query %.1f, "
              "parent tests per wave-query %.2f (lanes %.1f)" % (frame, c[0] / wq, c[1] / max(c[0], 1), 100 * c[1] / max(c[0], 1) / 64,
                                                                 c[1] / (n * 1081.0), c[2] / wq, c[3] / max(c[2], 1)))
    h.step(frame, scan); frame += 1
