"""debug: frame mode of every virtual rank, frame by frame"""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
pkg = importlib.import_module("gpu-icp-slam_amd")
import test_gpu_sharded as T
world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 30001 if world == 3 else 30000
tree, scans = T._bench_like(pkg, n, n_frames=16)
v = T._VirtualRanks(pkg, torch, n, world, kd_capacity=len(tree) + (1 << 18))
for e in v.engs:
    e.set_map(tree)
    for f in range(1, 6):
        e.motion_update(f)
for i, s in enumerate(scans):
    v.step(6 + i, s)
    print(i, [(e.frame_mode()["round5_frame"], e.frame_mode()["gates"], e.cell_stats()["flags"], e.cell_stats()["cells"]) for e in v.engs], flush=True)
