#!/bin/bash
# the three pieces of wave_level_tests.patch on their own (-DPF_WV=1: range test on the bit pattern; 2: +-20 m test decided per wave; 4: window test decided per wave)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 0 1 2 4; do
  PFSLAM_LIB=$GRAFT_REPO_ROOT/tools/experiments/r05/libs/libpfslam_wv$v.so python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('PF_WV=$v step %.4f ms kernel %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done | tee gpurun_out/ab_wave_parts.txt
