#!/bin/bash
# LDS window of inserted nodes in the score kernel: default vs variant 4 (no window), and the beam-chunk count of the
# 16-wave workgroups (PFSLAM_TARGET_WAVES)
for v in 0 4; do
  echo "variant $v"; python bench.py --no-cpu-baseline --variant $v "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'kernel', round(r['kernel_ms'],3), r['gathers']['wave_gathers_16B_per_launch'], r['gathers']['wave_gathers_4B_per_launch'])"
done
for t in 65536 32768 16384; do
  echo "target waves $t"; PFSLAM_TARGET_WAVES=$t python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'kernel', round(r['kernel_ms'],3))"
done
