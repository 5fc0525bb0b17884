#!/usr/bin/env python3
"""Scoring pass (pfslam_time_score_kd, 40 launches) on the aged bench state, printed 3 times: a steadier A/B figure than the 20-frame bench
window when two builds (PFSLAM_EXTRA_FLAGS) are compared on one box."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("gpu-icp-slam_amd")
pts, segs = pkg.synth.make_map_points(100000, seed=1)
h = pkg.PfSlam(100000, kd_capacity=100000 + (1 << 18))
h.set_map(pkg.kd_create(pts))
for f in range(1, 6):
    h.motion_update(f)
for i in range(25):
    h.step(6 + i, pkg.synth.make_scan(segs, (0.002 * i, 0.001 * i, 0.0004 * i), seed=2000 + i))
h.score_kd()
print(os.environ.get("PFSLAM_EXTRA_FLAGS", ""), " ".join("%.4f" % h.time_score_kd(40) for _ in range(3)), flush=True)
