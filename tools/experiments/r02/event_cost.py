import sys, time, importlib, numpy as np
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("gpu-icp-slam_amd")
pts, segs = pkg.synth.make_map_points(100000, seed=1)
tree = pkg.kd_create(pts)
N = 100000
frames = [pkg.synth.make_scan(segs, (0.004 * k, 0.002 * k, 0.001 * k), seed=500 + k) for k in range(160)]
sys.path.insert(0, "/root/repo/tests"); import oracle_lib as O
for timing in (0, 1, 0, 1):
    h = pkg.PfSlam(N, kd_capacity=len(tree) + (1 << 16))
    h.set_map(tree)
    p = O.make_particles(N, 0.0, 0.0, 0.0); h.set_particles(p)
    for f in range(1, 6): h.motion_update(f)
    f = 6
    for k in range(5): h.step(f, frames[k]); f += 1
    h.synchronize(); h.set_timing(timing)
    t0 = time.perf_counter()
    for k in range(5, 25): h.step(f, frames[k]); f += 1
    h.synchronize()
    dt = time.perf_counter() - t0
    print("timing", timing, "ms/step %.4f" % (dt / 20 * 1e3))
    h.close()
