#!/bin/bash
# heading weight of the lane order (cells equal in x, y and reach x theta x weight) on the plan-driven build
for w in 0.5 1 1.5 2 3; do
  echo "theta weight $w"
  PFSLAM_THETA_WEIGHT=$w python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'kernel', round(r['kernel_ms'],3), 'plan', round(r['plan']['kernel_ms'],3), 'cand', round(r['plan']['candidates'],2), 'path', round(r['plan']['path_len'],1), 'box', r['plan']['box_dx'], r['plan']['box_dtheta'])"
done
