#!/bin/bash
# A/B: build variants on the GPU box (hipcc exists there too) and bench each
for v in "$@"; do
  PFSLAM_EXTRA_FLAGS="$v" python gpu-icp-slam_amd/build.py > /dev/null 2>&1
  touch gpu-icp-slam_amd/csrc/kd_device.h
  echo "== $v"
  python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step ms %.3f  value %.3e  kernel ms %.3f kernel evals/s %.3e'%(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel_evals_per_s']))"
done
