#!/bin/bash
# tools/experiments/r06/same_device.sh -- ON THE GPU BOX: the multi-process sharded frame with every rank on GPU 0 (gloo carries the collectives,
# issued with the frame's own stream as torch's current stream: RCCL refuses two ranks on one device).  2 and 8 processes x 4 streams on one
# GPU: does the gates' self-test hold, which mode do the frames report, do the ranks agree with the single handle?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 2 8; do
  per=$((100000 / n))
  timeout 600 python bench.py --gpus $n --same-device --backend gloo --particles $per --no-cpu-baseline --steps 20 --warmup 5 2>gpurun_out/same_device_$n.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'n_procs_one_gpu': d['n_gpus'], 'particles_per_rank': $per, 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'frame_mode': d['config'].get('frame_mode'), 'per_rank_ms_per_step': d.get('per_rank_ms_per_step'), 'steady_state': d.get('steady_state'), 'collectives': d.get('collectives')}))"
  tail -3 gpurun_out/same_device_$n.err
done | tee gpurun_out/r06_same_device.txt
