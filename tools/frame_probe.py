#!/usr/bin/env python3
"""tools/frame_probe.py [--particles N] [--map-points K] [--frames F] [--serial]  -- ON THE GPU BOX.
Where a round-5 frame's time goes, read off the frame's own kernels (pfslam_set_probe: the first thread of every launch stores the
100 MHz wall clock -- no profiler attached, no extra launches, no events).  Prints, per launch, the mean start time relative to the
start of the frame's scan-match kernel, and the two numbers the frame is judged by:
  chain = start of the next frame's scan-match kernel - start of this frame's reduce (what sits between two scan-match kernels)
  frame = start of the next frame's scan-match kernel - start of this frame's
Same workload as bench.py (synthetic 1081-beam scans, 100 k particles, 100 k-point map, frames 11-30 of the run)."""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("gpu-icp-slam_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=100000)
    ap.add_argument("--map-points", type=int, default=100000)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--serial", action="store_true")
    ap.add_argument("--variant", type=int, default=0)
    a = ap.parse_args()
    pts, segs = pkg.synth.make_map_points(a.map_points, seed=1)
    tree = pkg.kd_create(pts)
    n = a.warmup + a.frames
    scans = [pkg.synth.make_scan(segs, (0.002 * f, 0.001 * f, 0.0004 * f), seed=2000 + f) for f in range(n)]
    e = pkg.PfSlam(a.particles, kd_capacity=a.map_points + (1 << 18))
    e.set_map(tree)
    if a.variant:
        e.set_variant(a.variant)
    if a.serial:
        e.set_serial(1)
    for f in range(1, 6):
        e.motion_update(f)
    e.set_probe(n + 8)
    for k in range(n):
        e.step(6 + k, scans[k])
    e.synchronize()
    names, t, last = e.probe(n)
    sc, rd = names.index("C scan-match"), names.index("C reduce")
    t = t[-a.frames:]
    ok = (t[:, sc] > 0) & (t[:, rd] > 0)
    print("frames with stamps: %d of %d (ticket %d last)" % (int(ok.sum()), len(t), last))
    rel = np.where(t > 0, t - t[:, sc:sc + 1], np.nan)
    print("%-20s %10s %10s" % ("launch", "start us", "(min..max)"))
    order = np.argsort(np.nanmean(rel, axis=0))
    for k in order:
        col = rel[:, k]
        if np.all(np.isnan(col)):
            continue
        print("%-20s %+10.1f   %+.1f .. %+.1f" % (names[k], np.nanmean(col), np.nanmin(col), np.nanmax(col)))
    nxt = t[1:, sc] - t[:-1, sc]
    chain = t[1:, sc] - t[:-1, rd]
    good = ok[1:] & ok[:-1]
    print("frame  (scan-match start to next scan-match start): mean %.1f us  min %.1f  max %.1f" % (nxt[good].mean(), nxt[good].min(), nxt[good].max()))
    print("chain  (reduce start to next scan-match start):     mean %.1f us  min %.1f  max %.1f" % (chain[good].mean(), chain[good].min(), chain[good].max()))
    print("scan-match kernel (start to reduce start, incl. its launch gap): mean %.1f us" % (t[:, rd] - t[:, sc])[ok].mean())
    print("cell stats:", {k: v for k, v in e.cell_stats().items() if k in ("cells", "rows", "pool_slots", "walked_from_root", "extended", "updates", "wipes", "flags")})
    print("cell check:", e.check_cells())
    print("frame mode:", e.frame_mode())
    e.close()


if __name__ == "__main__":
    main()
