# lane order at 1 M particles in one handle: the 2^21-cell counting sort (default) vs 30-bit Hilbert keys + radix sort (variant 6)
for v in 0 6; do
  echo "variant $v"
  python bench.py --no-cpu-baseline --particles 1000000 --steps 10 --warmup 5 --variant $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'kernel', round(r['kernel_ms'],3), 'plan', round(r['plan']['kernel_ms'],3), r['plan']['candidates'], r['plan']['box_dx'], r['plan']['box_dtheta'])"
done
