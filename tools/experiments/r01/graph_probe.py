"""Direct launches vs a captured hipGraph for the launch-bound chain of the 2-D frame loop (run on the GPU box)."""
import ctypes as C, importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gpu-icp-slam_amd")
_, frames = pkg.synth.corridor_sequence(6, seed=5)
for n in (50, 1000, 10000, 100000):
    h = pkg.PfSlam(n)
    for f in range(1, 6):
        h.step_grid(f, frames[f - 1][1])
    out = (C.c_float * 2)()
    rc = h.L.pfslam_debug_graph_probe(h._h, 200, out)
    print(n, "rc", rc, "direct %.4f ms  graph %.4f ms" % (out[0], out[1]), flush=True)
    h.close()
