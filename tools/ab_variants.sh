#!/bin/bash
# scoring-pass variants: 0 = default, 1 = identity lane order, 2 = no shared-prefix plan, 3 = plan at any size
for v in "$@"; do
  python bench.py --no-cpu-baseline --variant $v 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v: step ms %.3f  value %.3e  kernel ms %.3f kernel evals/s %.3e'%(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel_evals_per_s']))"
done
