"""split_cmp.py: the bench workload for 30 frames with the map update as one chain (PFSLAM_MAP_SPLIT=0) and as two; prints a digest of
the final map (bytes), particles, pose and the cell-row statistics of each -- they must be equal."""
import hashlib, importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CHILD = r"""
import importlib, sys, hashlib, json, numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module('gpu-icp-slam_amd')
pts, segs = pkg.synth.make_map_points(100000, seed=1)
tree = pkg.kd_create(pts)
h = pkg.PfSlam(100000, kd_capacity=100000 + (1 << 18))
h.set_map(tree)
for f in range(1, 6): h.motion_update(f)
out = []
for i in range(30):
    h.step(6 + i, pkg.synth.make_scan(segs, (0.002 * i, 0.001 * i, 0.0004 * i), seed=2000 + i))
    if i %% 5 == 4:
        st = h.cell_stats()
        out.append((i, hashlib.md5(h.map().tobytes()).hexdigest()[:8], [int(v) for v in h.pose.view(np.int32)], int(st['extended']), int(st['cells']), int(st['rows']), h.trace()))
p = h.particles()
print(json.dumps({'frames': out, 'particles': hashlib.md5(b''.join(np.ascontiguousarray(p[k]).tobytes() for k in ('x', 'y', 'theta', 'w'))).hexdigest()}))
""" % ROOT
res = {}
for name, env in (("one_chain", {"PFSLAM_MAP_SPLIT": "0"}), ("two_chains", {"PFSLAM_MAP_SPLIT": "1"}), ("two_chains_again", {"PFSLAM_MAP_SPLIT": "1"})):
    o = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
    res[name] = json.loads(o.stdout.strip().splitlines()[-1]) if o.returncode == 0 else o.stderr[-2000:]
    print(name, json.dumps(res[name])[:1500])
print("EQUAL" if res["one_chain"] == res["two_chains"] == res["two_chains_again"] else "DIFFERENT")
