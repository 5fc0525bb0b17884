#!/usr/bin/env python3
"""Scoring-pass time with and without the shared-prefix plan over particle counts (GPU box): where does the plan start to pay?"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ["PFSLAM_PLAN_MIN_N"] = "1"
pkg = importlib.import_module("gpu-icp-slam_amd")
pts, segs = pkg.synth.make_map_points(100000, seed=1)
tree = pkg.kd_create(pts)
for n in (1000, 2000, 5000, 10000, 20000, 50000):
    res = {}
    for variant in (0, 2):
        h = pkg.PfSlam(n, kd_capacity=100000 + (1 << 18))
        h.set_map(tree); h.set_variant(variant)
        for f in range(1, 6):
            h.motion_update(f)
        for i in range(25):
            h.step(6 + i, pkg.synth.make_scan(segs, (0.002 * i, 0.001 * i, 0.0004 * i), seed=2000 + i))
        res[variant] = h.time_score_kd(10)
        h.close()
    print("n %6d  plan %.4f ms  plain %.4f ms  ratio %.2f" % (n, res[0], res[2], res[2] / res[0]))
