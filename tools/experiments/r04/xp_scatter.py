import importlib, sys, numpy as np
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module('gpu-icp-slam_amd')
pts, segs = pkg.synth.make_map_points(100000, seed=1)
tree = pkg.kd_create(pts)
h = pkg.PfSlam(1000, kd_capacity=100000 + (1 << 18))
h.set_map(tree)
scan = pkg.synth.make_scan(segs, (0.0, 0.0, 0.0), seed=2000)
h.step(6, scan)
for i in range(60):
    h.update_map_kd()
h.close()
