#!/usr/bin/env python3
"""Generates tests/golden/golden_v2.npz -- vectors for the rows added after golden_v1: the 2-D occupancy-grid frame
loop (BASELINE configs[0-1]) and the topology graph / loop-closure proposal (kernel.cu:623-795).

Expected values come from the CPU oracle (the reference cannot execute these here: no nvcc / GPU; they are not even
called by its shipped step).  Inputs: golden_v1's replay scans and a seeded closed loop.
Run from the repo root:  python tests/golden/make_golden_v2.py
"""
import importlib
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402

pkg = importlib.import_module("gpu-icp-slam_amd")
v1 = np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))
out = {}

# ---- 2-D frame loop: 200 particles over golden_v1's 10 replay scans
scans = v1["replay_scans"]
o = O.Slam(200)
rows = []
for f, scan in enumerate(scans, start=1):
    o.step_grid(f, scan)
    t = o.trace()
    rows.append([t["best"], t["resampled"], int(np.float32(t["neff"]).view(np.int32))] + [int(v) for v in o.pose.view(np.int32)])
out["grid_replay_trace"] = np.array(rows, np.int64)
g = o.grid
out["grid_replay_crc"] = np.uint32(zlib.crc32(g.tobytes()))
out["grid_replay_free_count"] = np.int32((g < -100).sum())
cells = np.flatnonzero(g.ravel() > -100).astype(np.int32)   # the wall side explicitly; the whole grid through the crc
out["grid_replay_wall_cells"], out["grid_replay_wall_vals"] = cells, g.ravel()[cells]
p = o.particles()
out["grid_replay_particles"] = np.stack([p["x"], p["y"], p["theta"], p["w"]], 1)
o.close()

# ---- topology graph on a closed 12 m square driven twice, closure pairs against a grid with the square's walls
topo = O.Topology()
path = []
for k in range(0, 97):
    s = (k % 48) / 12.0
    side, u = int(s), (s - int(s)) * 12.0
    xy = [(u - 6, -6), (6, u - 6), (6 - u, 6), (-6, 6 - u)][side]
    path.append((xy[0], xy[1], 0.1 * k))
path = np.array(path, np.float32)
created = [topo.update(r) for r in path]
out["topo_path"], out["topo_created"] = path, np.array(created, np.int32)
out["topo_nodes"] = topo.nodes()
grid = np.full((1600, 1600), -100, np.int8)
grid[800 - 80:800 + 80, 800 - 80] = 113     # an inner wall segment between two sides of the loop
grid[800 - 80:800 + 80, 800 + 80] = 113
out["topo_grid_cells"] = np.flatnonzero(grid.ravel() != -100).astype(np.int32)
out["topo_pairs"] = topo.loop_closure(grid, path[-1])
out["topo_walls_a"] = np.array([0.0, -6.0], np.float32)
out["topo_walls_b"] = np.array([0.0, 6.0], np.float32)
out["topo_walls_n"] = np.int32(O.find_walls(grid, out["topo_walls_a"], out["topo_walls_b"]))

dst = os.path.join(ROOT, "tests", "golden", "golden_v2.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, {k: getattr(v, "shape", None) for k, v in out.items()})
