#!/usr/bin/env python3
"""Shared-prefix plan statistics on the bench workload (run on the GPU box): python tools/plan_probe.py [particles] [map_points] [variant]"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("gpu-icp-slam_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pts, segs = pkg.synth.make_map_points(k, seed=1)
h = pkg.PfSlam(n, kd_capacity=k + (1 << 18))
h.set_map(pkg.kd_create(pts))
h.set_variant(variant)
for f in range(1, 6):
    h.motion_update(f)
for i in range(25):
    h.step(6 + i, pkg.synth.make_scan(segs, (0.002 * i, 0.001 * i, 0.0004 * i), seed=2000 + i))
    if i in (0, 4, 24):
        c = h.score_census(); s = h.plan_stats()
        q = s["rows"] or 1
        print("frame", 6 + i, "ms", round(h.time_score_kd(5), 4), "trips/query", round(c["trips"] / q, 2), "lanes/trip", round(c["visits"] / max(c["trips"], 1), 1),
              "tests/query", round(c["tests"] / q, 2), "uniform", round(c["uniform_trips"] / max(c["trips"], 1), 3), "redesc/lane-query", round(c["redescents"] / (q * 64), 3), "noop", round(c["redescents_noop"] / max(c["redescents"], 1), 3), json.dumps({a: round(b, 5) for a, b in s.items()}))
