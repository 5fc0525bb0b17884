#!/usr/bin/env python3
"""tools/experiments/r05/clock_ramp.py [--spin MS] -- ON THE GPU BOX.  Per-block frame time of the bench workload (10-frame blocks, frames 6..),
with and without GPU work in front of it: is a frame early in the process slower because of what it computes (the map / the cloud evolve)
or because of when it runs (clocks)?"""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("gpu-icp-slam_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--spin", type=float, default=0.0, help="ms of frames on a second handle first")
ap.add_argument("--frames", type=int, default=100)
a = ap.parse_args()
pts, segs = pkg.synth.make_map_points(100000, seed=1)
tree = pkg.kd_create(pts)
scans = [pkg.synth.make_scan(segs, (0.002 * f, 0.001 * f, 0.0004 * f), seed=2000 + f) for f in range(a.frames)]
def engine():
    e = pkg.PfSlam(100000, kd_capacity=100000 + (1 << 18))
    e.set_map(tree)
    for f in range(1, 6):
        e.motion_update(f)
    return e
if a.spin > 0:
    w = engine()
    t0 = time.perf_counter(); k = 0
    while (time.perf_counter() - t0) * 1e3 < a.spin:
        w.step(6 + k % 90, scans[k % 90]); k += 1
        if k % 10 == 0: w.synchronize()
    w.synchronize(); w.close()
e = engine()
e.synchronize()
out = []
for b in range(0, a.frames - 10, 10):
    t0 = time.perf_counter()
    for k in range(b, b + 10):
        e.step(6 + k, scans[k])
    e.synchronize()
    out.append((time.perf_counter() - t0) * 100)
print("spin %.0f ms: ms/frame per 10-frame block:" % a.spin, " ".join("%.3f" % v for v in out))
