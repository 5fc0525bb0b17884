#!/bin/bash
# tools/ab_env.sh "VAR=a VAR2=b" "VAR=c" ...  -- ON THE GPU BOX: bench.py --no-cpu-baseline once per environment setting ("-" = none),
# each twice; prints frame ms, evals/s, scan-match kernel ms, planning ms.  Extra bench.py args after "--".
ARGS=""
SETS=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; ARGS="$*"; break; fi
  SETS+=("$1"); shift
done
for s in "${SETS[@]}"; do
  for rep in 1 2; do
    if [ "$s" == "-" ]; then e=""; else e="$s"; fi
    env $e python bench.py --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=r.get('cells') or {}
print('%-40s step %.4f ms  %.4e evals/s  kernel %.4f ms  plan %.4f ms  cells %s' % ('$s', d['ms_per_step'], d['value'], r['kernel_ms'], (c.get('kernel_ms') or 0), {k: c.get(k) for k in ('cells','walked_from_root','extended','updates','wipes')}))"
  done
done
