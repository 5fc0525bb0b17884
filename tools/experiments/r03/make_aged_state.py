"""CPU prototype input for cell_rows_sim.c: the bench workload's map aged by the oracle SLAM (few particles: the map evolution
depends on the best pose only), a 100 k-particle cloud with the oracle cloud's mean / covariance, and the next scan.
    python tools/experiments/r03/make_aged_state.py /tmp/aged 30"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O
pkg = importlib.import_module("gpu-icp-slam_amd")
out, last = sys.argv[1], int(sys.argv[2])
pts, segs = pkg.synth.make_map_points(100000, seed=1)
tree = pkg.kd_create(pts)
n = 2000
o = O.Slam(n, kd_capacity=100000 + (1 << 18))
o.set_map(tree)
p = O.make_particles(n)
for f in range(1, 6):
    O.add_noise(p, f)
o.set_particles(p)
for k in range(last - 6 + 1):
    f = 6 + k
    o.step(f, pkg.synth.make_scan(segs, (0.002 * k, 0.001 * k, 0.0004 * k), seed=2000 + k))
    print(f, o.trace(), flush=True)
t = o.tree()
P = o.particles()
k = last - 6 + 1
scan = pkg.synth.make_scan(segs, (0.002 * k, 0.001 * k, 0.0004 * k), seed=2000 + k)
# the cloud as the next frame scores it: dispersed once more
O.add_noise(P, last + 1)
X = np.stack([P["x"], P["y"], P["theta"]], 1).astype(np.float64)
mu, cov = X.mean(0), np.cov(X.T)
big = np.random.RandomState(1).multivariate_normal(mu, cov, 100000).astype(np.float32)
t.tofile(out + ".nodes"); big.tofile(out + ".particles"); scan.astype(np.float32).tofile(out + ".scan")
print("nodes", len(t), "cloud sigma", np.sqrt(np.diag(cov)))
