#!/bin/bash
# tools/sharded_vs_single.sh [tag]  -- run ON THE GPU BOX from the repo root (via gpurun).
# The sharded frame (pfslam_shard_*: three all-gathers on a fixed schedule, no host wait) against the single-GPU frame
# (pfslam_step) on ONE GPU and the SAME workload (100 000 particles, 100 000-point map, frames 6..30, 20 timed):
#   a. python bench.py --gpus 1                                  pfslam_step
#   b. torchrun --nproc-per-node 1 bench.py --gpus 1             ShardedSlam over torch.distributed (RCCL, world 1)
#   c. host/pfslam_mgpu --gpus 1                                 the C++ driver on librccl
# -> gpurun_out/<tag>_sharded_vs_single.json (copy into profiles/)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY'
import importlib, numpy as np
pkg = importlib.import_module("gpu-icp-slam_amd")
pts, segs = pkg.synth.make_map_points(100000, seed=1)
pkg.kd_create(pts).tofile("/tmp/bench_map.nodes")
scans = np.stack([pkg.synth.make_scan(segs, (0.002 * f, 0.001 * f, 0.0004 * f), seed=2000 + f) for f in range(25)]).astype(np.float32)
scans.tofile("/tmp/bench_scans.f32")
PY
for rep in 1 2 3; do
  python bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | grep '^{' > /tmp/a$rep.json
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2955$rep bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | grep '^{' > /tmp/b$rep.json
  gpu-icp-slam_amd/host/pfslam_mgpu --gpus 1 /tmp/bench_map.nodes /tmp/bench_scans.f32 100000 --steps 20 --warmup 5 2>/dev/null | grep '^{' > /tmp/c$rep.json
done
python - $TAG <<'PY'
import json, sys
tag = sys.argv[1]
def best(prefix):
    ms = [json.load(open("/tmp/%s%d.json" % (prefix, r)))["ms_per_step"] for r in (1, 2, 3)]
    return min(ms), ms
a, al = best("a"); b, bl = best("b"); c, cl = best("c")
out = {"workload": "100000 particles x 1081 beams x 100000-point map, frames 6..30, 20 timed steps, one MI355X; best of 3 runs each",
       "pfslam_step_ms": a, "sharded_py_torchrun_rccl_world1_ms": b, "sharded_cpp_rccl_world1_ms": c,
       "sharded_py_over_single": b / a, "sharded_cpp_over_single": c / a, "runs_ms": {"single": al, "sharded_py": bl, "sharded_cpp": cl},
       "pose_cpp": json.load(open("/tmp/c1.json"))["config"]["pose"]}
json.dump(out, open("gpurun_out/%s_sharded_vs_single.json" % tag, "w"), indent=1)
print(json.dumps(out))
PY
