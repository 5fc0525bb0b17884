"""tools/experiments/r06/gather_cost.py make|time STATE.npz -- what one gather per beam costs the scan-match kernel, on a FIXED state.
make: the default library steps the bench workload to frame 30 and saves particles, map and scan.
time: the library named by PFSLAM_LIB (a timing-only build: -DPF_X_NOWEIGHT / -DPF_X_NOSLOT0, results wrong by construction) loads that state
      and times 40 scoring passes (pfslam_time_score_kd: lane order + scan-match kernel + reduce) -- no feedback into the state."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("gpu-icp-slam_amd")
mode, path = sys.argv[1], sys.argv[2]
N, M = 100000, 100000
if mode == "make":
    pts, segs = pkg.synth.make_map_points(M, seed=1)
    tree = pkg.kd_create(pts)
    h = pkg.PfSlam(N, kd_capacity=M + (1 << 18))
    h.set_map(tree)
    for f in range(1, 6):
        h.motion_update(f)
    scan = None
    for f in range(6, 31):
        scan = pkg.synth.make_scan(segs, (0.002 * (f - 6), 0.001 * (f - 6), 0.0004 * (f - 6)), seed=2000 + f - 6)
        h.step(f, scan)
    h.synchronize()
    np.savez(path, particles=h.particles(), tree=h.map(), scan=scan)
    print("state saved:", len(h.map()), "nodes")
else:
    z = np.load(path)
    h = pkg.PfSlam(N, kd_capacity=len(z["tree"]) + (1 << 18))
    h.set_map(z["tree"]); h.set_particles(z["particles"]); h.set_scan(z["scan"])
    h.score_kd()                      # the rows are made here
    h.time_score_kd(5)
    t = [h.time_score_kd(20) for _ in range(3)]
    c = h.score_census()
    print(os.path.basename(os.environ.get("PFSLAM_LIB", "default")), "ms per scoring pass: %.4f %.4f %.4f" % tuple(t), " census trips %d uniform %d" % (c.get("trips", 0), c.get("uniform_trips", 0)))
