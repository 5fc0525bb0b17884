#!/bin/bash
# 1 M particles in one handle: the 2^21-cell counting sort of the lane order (round 2 measured it equal to 30-bit Hilbert keys + a
# library radix sort -- 109.67 vs 109.99 M evals/s -- and the radix-sort path was deleted)
python bench.py --no-cpu-baseline --particles 1000000 --steps 10 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'kernel', round(r['kernel_ms'],3), 'plan', round(r['plan']['kernel_ms'],3), r['plan']['candidates'], r['plan']['box_dx'], r['plan']['box_dtheta'])"
