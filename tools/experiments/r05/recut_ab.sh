#!/bin/bash
# tools/experiments/r05/recut_ab.sh -- ON THE GPU BOX: how often the publishing pass re-cuts every row (pool compaction), 64 timed frames each
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for e in 8 16 32 64 0; do
  PFSLAM_CELLS_RECUT_EVERY=$e python bench.py --no-cpu-baseline --steps 64 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=r.get('cells') or {}
print('recut_every=$e step %.4f ms kernel %.4f ms pool_slots %s rows %s' % (d['ms_per_step'], r['kernel_ms'], c.get('pool_slots'), c.get('rows')))"
done; done 2>&1 | tee gpurun_out/recut_ab.txt
