import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import importlib, time, numpy as np, sys
pkg=importlib.import_module('gpu-icp-slam_amd')
K=int(sys.argv[1]); N=int(sys.argv[2])
pts,segs=pkg.synth.make_map_points(K, seed=1)
tree=pkg.kd_create(pts)
h=pkg.PfSlam(N, kd_capacity=K+(1<<18)); h.set_map(tree)
for f in range(1,6): h.motion_update(f)
for k in range(104):
    f=6+k
    scan=pkg.synth.make_scan(segs,(0.002*k,0.001*k,0.0004*k),seed=2000+k)
    t0=time.perf_counter(); h.step(f,scan); dt=time.perf_counter()-t0
    if f in (104,105,106): print(f, 'host step ms %.2f'%(dt*1e3))
h.synchronize()
