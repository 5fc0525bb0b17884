#!/usr/bin/env python3
"""tools/experiments/r05/frame_components.py [particles] [map_points] [frames] -- ON THE GPU BOX.  Per 10-frame block of one run (frames 6 ..): what a
frame is made of, read off the frame probe -- scan-match kernel (start -> reduce start), reduce + walls + insert (reduce start -> cells update
start), publishing pass (start -> last workgroup's end), the whole frame (scan-match start -> next) -- and the cell statistics' deltas."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
pkg = importlib.import_module("gpu-icp-slam_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
mp = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 150
pts, segs = pkg.synth.make_map_points(mp, seed=1)
tree = pkg.kd_create(pts)
h = pkg.PfSlam(n, kd_capacity=mp + (1 << 18))
h.set_map(tree)
for f in range(1, 6):
    h.motion_update(f)
scans = [pkg.synth.make_scan(segs, (0.002 * f, 0.001 * f, 0.0004 * f), seed=2000 + f) for f in range(frames)]
h.set_probe(frames + 8)
stats = []
for k in range(frames):
    h.step(6 + k, scans[k])
    if k % 10 == 9:
        h.synchronize()
        stats.append(dict(h.cell_stats()))
h.synchronize()
names, tp, last = h.probe(frames)
ix = {nm: i for i, nm in enumerate(names)}
sc, rd, cu, cue = ix["C scan-match"], ix["C reduce"], ix["C cells update"], ix["C cells update: last wg ends"]
first_frame = 6 + frames - len(tp)
print("block(frames)   frame_us  scan-match  reduce..update  cells-update  | extended/frame  claimed/frame  cells")
prev = None
for b in range(0, len(tp) - 10, 10):
    rows = tp[b:b + 11]
    ok = (rows[:-1, sc] > 0) & (rows[1:, sc] > 0) & (rows[:-1, rd] > 0) & (rows[:-1, cu] > 0)
    if not ok.any():
        continue
    fr = (rows[1:, sc] - rows[:-1, sc])[ok]
    sm = (rows[:-1, rd] - rows[:-1, sc])[ok]
    mid = (rows[:-1, cu] - rows[:-1, rd])[ok]
    upd = (rows[:-1, cue] - rows[:-1, cu])[ok]
    st = stats[(first_frame - 6 + b) // 10] if (first_frame - 6 + b) // 10 < len(stats) else None
    d = ""
    if st is not None:
        if prev is not None and st.get("wipes") == prev.get("wipes"):
            d = "%8.0f %12.0f %10.0f  cand/row %.3f  redesc/row %.3f  rows %.0f  norow %.0f  slots %.0f" % (
                (st["extended"] - prev["extended"]) / 10, (st["claimed"] - prev["claimed"]) / 10, st["cells"], st["candidates"], st["redescent_candidates"],
                st["rows"], st["cells_without_row"], st["pool_slots"])
        else:
            d = "   (wipe in or before this block)   cells %.0f  cand/row %.3f  redesc/row %.3f  rows %.0f  norow %.0f  slots %.0f" % (
                st["cells"], st["candidates"], st["redescent_candidates"], st["rows"], st["cells_without_row"], st["pool_slots"])
        prev = st
    print("%3d..%3d  %9.1f %10.1f %12.1f %12.1f   | %s" % (first_frame + b, first_frame + b + 9, np.median(fr), np.median(sm), np.median(mid), np.median(upd), d))
