"""CPU prototype input for persist_sim.c: a run of consecutive bench frames.  For every frame f: the map as the frame's scoring pass
sees it, the frame's scan, and a 100 k-particle cloud with the oracle cloud's mean / covariance (the oracle SLAM runs with few
particles: the map evolution depends on the best pose only).
    python tools/experiments/r04/make_frames.py /tmp/pf 6 45 [particles]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O
pkg = importlib.import_module("gpu-icp-slam_amd")
out, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
big_n = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
pts, segs = pkg.synth.make_map_points(100000, seed=1)
tree = pkg.kd_create(pts)
n = 2000
o = O.Slam(n, kd_capacity=100000 + (1 << 18))
o.set_map(tree)
p = O.make_particles(n)
for f in range(1, 6):
    O.add_noise(p, f)
o.set_particles(p)
rs = np.random.RandomState(1)
for f in range(6, last + 1):
    k = f - 6
    scan = pkg.synth.make_scan(segs, (0.002 * k, 0.001 * k, 0.0004 * k), seed=2000 + k)
    if f >= first:
        t = o.tree()
        P = o.particles().copy()
        O.add_noise(P, f)  # the cloud as frame f scores it
        X = np.stack([P["x"], P["y"], P["theta"]], 1).astype(np.float64)
        mu, cov = X.mean(0), np.cov(X.T)
        big = rs.multivariate_normal(mu, cov, big_n).astype(np.float32)
        t.tofile("%s.%d.nodes" % (out, f)); big.tofile("%s.%d.particles" % (out, f)); scan.astype(np.float32).tofile("%s.%d.scan" % (out, f))
    o.step(f, scan)
    print(f, o.trace(), flush=True)
