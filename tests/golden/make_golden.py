#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz -- small input/expected-output vectors for every row of the hot path.

Sources of the expected values:
  * tree_ref_*     : the REFERENCE's own src/kdtree.cpp (compiled unmodified into oracle/_ref/libkdtree_ref.so by
                     oracle/Makefile) run in this container -- real reference output.
  * everything else: the CPU oracle (oracle/pfslam_oracle.c), the restatement of the device-only functions the
                     reference cannot execute here (no nvcc / GPU / libmat) -- see DESIGN.md "Oracle" for what is pinned.
Inputs come from the seeded synthetic generator (gpu-icp-slam_amd/synth.py); the reference's data/train_lidar*.mat are
absent from its checkout.  Run from the repo root:  python tests/golden/make_golden.py
"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402
import oracle_engine as OE  # noqa: E402

pkg = importlib.import_module("gpu-icp-slam_amd")
L = O.lib()
out = {}

# ---- A15 tree build: reference kdtree.cpp
ref = O.ref_kdtree()
assert ref is not None, "build oracle/_ref first (make -C oracle with /root/reference present)"
pts, segs = pkg.synth.make_map_points(1500, seed=21)
tree = np.zeros(len(pts), O.NODE_DTYPE)
ref.ref_kd_create(O.P(pts), len(pts), O.P(tree))
out["map_pts"], out["tree_ref_create"] = pts, tree.view(np.uint8).copy()
t2 = np.zeros(len(pts) + 40, O.NODE_DTYPE)
t2[:len(pts)] = tree
extra = pkg.synth.make_map_points(40, seed=22)[0]
extra[:, 3] = -100
for k in range(40):
    ref.ref_kd_insert_node(O.P(extra[k]), O.P(t2), len(pts) + k)
out["insert_pts"], out["tree_ref_inserted"] = extra, t2.view(np.uint8).copy()
t3 = t2.copy()
ref.ref_kd_balance(O.P(t3), len(t3))
out["tree_ref_balanced"] = t3.view(np.uint8).copy()

# ---- A2/A3 RNG + dispersion
fr, idx = np.meshgrid(np.arange(1, 9), np.array([0, 1, 2, 63, 64, 511, 512, 1023, 1024, 99999]), indexing="ij")
seeds = np.array([L.orc_engine_seed(int(a), int(b), 0) for a, b in zip(fr.ravel(), idx.ravel())], np.uint32)
out["rng_frames"], out["rng_idx"], out["rng_seeds"] = fr.ravel().astype(np.int32), idx.ravel().astype(np.int32), seeds
p = O.make_particles(256, 0.25, -0.5, 0.125)
out["noise_in"] = np.stack([p["x"], p["y"], p["theta"]], 1)
O.add_noise(p, 7, 0)
out["noise_frame"], out["noise_out"] = np.int32(7), np.stack([p["x"], p["y"], p["theta"]], 1)

# ---- A4/A5 traversal + score
scan = pkg.synth.make_scan(segs, (0.1, -0.2, 0.3), seed=23)
rng = np.random.RandomState(24)
q = np.zeros((4000, 3), np.float32)
q[:, :2] = rng.uniform(-20, 20, (4000, 2))
q[-1, :2] = (tree["x"][0], tree["y"][0])
best, visits = O.traverse_batch(tree, q)
out["scan"], out["trav_q"], out["trav_best"], out["trav_visits"] = scan, q, best, visits
parts = O.make_particles(64, 0.1, -0.2, 0.3)
O.add_noise(parts, 3, 0)
parts["w"] = rng.uniform(0.2, 1.0, 64).astype(np.float32)
out["score_particles"] = np.stack([parts["x"], parts["y"], parts["theta"], parts["w"]], 1)
fit = O.score_kd(tree, parts, scan)
out["score_fit"] = fit
# ---- A6
imin, imax = C.c_int(), C.c_int()
L.orc_minmax_first_f32(O.P(fit), 64, C.byref(imin), C.byref(imax))
pw = parts.copy()
rngv = np.float32(fit[imax.value]) - np.float32(fit[imin.value])
L.orc_update_weights_f32(O.P(pw), 64, O.P(fit), float(np.float32(1) / rngv), int(fit[imin.value]))
out["meas_best"], out["meas_w"] = np.int32(imax.value), pw["w"].copy()
# ---- A7-A9
robot, start = np.array([0.1, -0.2, 0.3], np.float32), np.array([0.12, -0.19, 0.31], np.float32)
pose, dbg = O.icp(tree, robot, start, scan)
out["icp_robot"], out["icp_start"], out["icp_pose"], out["icp_dbg"] = robot, start, pose, dbg[:28]
A = rng.uniform(-3, 3, (8, 9)).astype(np.float32)
usv = np.zeros((8, 27), np.float32)
for k in range(8):
    u, s, v = np.zeros(9, np.float32), np.zeros(9, np.float32), np.zeros(9, np.float32)
    L.orc_svd3(O.P(A[k]), O.P(u), O.P(s), O.P(v))
    usv[k] = np.concatenate([u, s, v])
out["svd_a"], out["svd_usv"] = A, usv
# ---- A10
fm, wm = O.get_walls(scan, 800, 800, np.float32(0.3))
out["walls_theta"], out["walls_wall_cells"], out["walls_free_cells"] = np.float32(0.3), np.flatnonzero(wm).astype(np.int32), np.flatnonzero(fm).astype(np.int32)
# ---- A11-A15 map update
cap = len(tree) + 600
t4 = np.zeros(cap, O.NODE_DTYPE)
t4[:len(tree)] = tree
size = OE.oracle_map_update(t4, len(tree), robot, scan, cap)
out["mapupd_tree"] = t4[:size].view(np.uint8).copy()
# ---- A16 resample
n = 3000
rp = O.make_particles(n)
rp["x"] = np.arange(n)
rp["w"] = rng.uniform(0, 1, n).astype(np.float32) ** 6
out["resample_w"] = rp["w"].copy()
src = np.zeros(n, np.int32)
neff = C.c_float()
assert L.orc_resample(O.P(rp), n, 11, C.byref(neff), O.P(src)) == 1
out["resample_frame"], out["resample_src"], out["resample_neff"] = np.int32(11), src, np.float32(neff.value)
# ---- A17/A18 grid
dim = 1600
grid = np.full((dim, dim), -100, np.int8)
gx = np.round(800 + pts[:, 0] / 0.025).astype(int)
gy = np.round(800 + pts[:, 1] / 0.025).astype(int)
grid[gx, gy] = rng.randint(-113, 114, len(pts))
out["grid_cells"], out["grid_vals"] = (gx * dim + gy).astype(np.int32), grid[gx, gy].copy()
patch = O.default_patch()
gfit = np.zeros(64, np.int32)
L.orc_score_grid(O.P(grid), dim, dim, C.byref(patch), O.P(parts), 64, O.P(scan), 1081, O.P(gfit))
out["grid_fit"] = gfit
g2 = grid.copy()
L.orc_update_map_grid(O.P(g2), dim, dim, C.byref(patch), O.P(robot), O.P(scan), 1081)
ch = np.flatnonzero(g2.ravel() != grid.ravel()).astype(np.int32)
out["grid_upd_cells"], out["grid_upd_vals"] = ch, g2.ravel()[ch].copy()
# ---- whole step replay
segs2, frames = pkg.synth.corridor_sequence(10, seed=5)
o = O.Slam(200, kd_capacity=1 << 16)
tr = []
for f, (pz, sc) in enumerate(frames, start=1):
    o.step(f, sc)
    t = o.trace()
    tr.append([t["best"], t["resampled"], t["n_wall"], t["n_free"], t["n_insert"], t["kd_size"]] + o.pose.view(np.int32).tolist())
out["replay_scans"] = np.stack([s for _, s in frames])
out["replay_trace"] = np.array(tr, np.int64)
fp = o.particles()
out["replay_particles"] = np.stack([fp["x"], fp["y"], fp["theta"], fp["w"]], 1)

path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
