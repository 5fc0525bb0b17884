"""How much of the score kernel's time is lane divergence?  Same map / scan, 100 k particles: (a) the dispersed cloud of the
bench, (b) all particles at one pose (every lane of every wave walks the same nodes), (c) the dispersed cloud with identity
lane order.  Run on the GPU box."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gpu-icp-slam_amd")


def make_particles(n, x, y, th):
    p = np.zeros(n, pkg.PARTICLE_DTYPE)
    p["x"], p["y"], p["theta"], p["w"] = x, y, th, 1.0
    return p

n = 100000
pts, segs = pkg.synth.make_map_points(100000, seed=1)
tree = pkg.kd_create(pts)
scan = pkg.synth.make_scan(segs, (0.0, 0.0, 0.0), seed=2000)
h = pkg.PfSlam(n, kd_capacity=1 << 18)
h.set_map(tree); h.set_scan(scan)
for sigma_steps, label in ((5, "dispersed (5 dispersion steps)"), (0, "all particles at one pose")):
    p = make_particles(n, 0.0, 0.0, 0.0)
    h.set_particles(p)
    for f in range(1, sigma_steps + 1):
        h.motion_update(f)
    for variant, vl in ((0, "Morton order"), (1, "identity order")):
        h.set_variant(variant)
        ms = h.time_score_kd(10)
        print("%-32s %-14s %.3f ms" % (label, vl, ms), flush=True)
h.close()

# ---- the same on the aged map of the bench (25 frames of inserts, resampled cloud)
h = pkg.PfSlam(n, kd_capacity=1 << 18)
h.set_map(tree)
for f in range(1, 6):
    h.motion_update(f)
frame = 6
for k in range(25):
    pose = (0.002 * k, 0.001 * k, 0.0004 * k)
    sc = pkg.synth.make_scan(segs, pose, seed=2000 + k)
    h.step(frame, sc); frame += 1
h.set_scan(sc)
for variant, vl in ((0, "Morton order"), (1, "identity order")):
    h.set_variant(variant)
    print("%-32s %-14s %.3f ms" % ("aged map, cloud after 25 frames", vl, h.time_score_kd(10)), flush=True)
p = h.particles().copy()
q = make_particles(n, float(h.pose[0]), float(h.pose[1]), float(h.pose[2]))
h.set_particles(q)
for variant, vl in ((0, "Morton order"), (1, "identity order")):
    h.set_variant(variant)
    print("%-32s %-14s %.3f ms" % ("aged map, all at the robot pose", vl, h.time_score_kd(10)), flush=True)
# spread statistics of the real cloud
print("cloud sigma x/y/theta:", float(np.std(p["x"])), float(np.std(p["y"])), float(np.std(p["theta"])))
h.close()
