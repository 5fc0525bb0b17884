#!/usr/bin/env python3
"""tools/balance_bench.py [--points 500000] [--out profiles/rNN_balance.json]

KDTree::Balance (kdtree.cpp:25-67 restated in csrc/kd_host.cpp) on the host cores of this box: one build with every usable core
(what the sharded frame does: rank 0 builds, the device arrays are broadcast) beside eight builds at once (what eight ranks did
before: each its own), with the thread budget split (LOCAL_WORLD_SIZE=8) and not (every rank as many threads as there are cores),
and the library's plain std::sort for scale.  CPU only; writes one JSON object."""
import argparse, importlib, json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import importlib, sys, time, numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module('gpu-icp-slam_amd')
n, reps = int(sys.argv[1]), int(sys.argv[2])
pts, _ = pkg.synth.make_map_points(n, seed=1)
tree = pkg.kd_create(pts)
sys.stdout.write('READY\n'); sys.stdout.flush()
sys.stdin.readline()                     # all children start their builds together
ts = []
for _ in range(reps):
    nodes = tree.copy()
    t0 = time.perf_counter(); pkg.kd_balance(nodes, len(nodes)); ts.append((time.perf_counter() - t0) * 1e3)
print('MS', ' '.join('%%.3f' %% t for t in ts), 'THREADS', pkg.binding.load().pfslam_kd_sort_threads())
""" % ROOT


def run(n, reps, procs, env):
    ps = [subprocess.Popen([sys.executable, "-c", CHILD, str(n), str(reps)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                           env=dict(os.environ, **env)) for _ in range(procs)]
    for p in ps:
        assert p.stdout.readline().strip() == "READY"
    for p in ps:
        p.stdin.write("go\n"); p.stdin.flush()
    out = []
    for p in ps:
        line = [l for l in p.stdout.read().splitlines() if l.startswith("MS")][0].split()
        k = line.index("THREADS")
        out.append({"ms": [float(x) for x in line[1:k]], "threads": int(line[k + 1])})
        p.wait()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=500000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = {"points": a.points, "reps": a.reps, "cores_usable": len(os.sched_getaffinity(0)), "cases": {}}
    for name, procs, env in (("one_build_all_cores", 1, {}),
                             ("eight_builds_budget_split", 8, {"LOCAL_WORLD_SIZE": "8"}),
                             ("eight_builds_every_rank_all_cores", 8, {}),
                             ("one_build_one_thread", 1, {"PFSLAM_SORT_THREADS": "1"}),
                             ("one_build_plain_std_sort_one_thread", 1, {"PFSLAM_PLAIN_SORT": "1", "PFSLAM_SORT_THREADS": "1"})):
        r = run(a.points, a.reps, procs, env)
        res["cases"][name] = {"processes": procs, "threads_per_process": r[0]["threads"],
                              "best_ms_per_process": [min(x["ms"]) for x in r],
                              "median_ms_slowest_process": max(sorted(x["ms"])[len(x["ms"]) // 2] for x in r)}
    s = json.dumps(res)
    print(s)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
