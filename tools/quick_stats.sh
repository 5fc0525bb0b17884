#!/bin/bash
# tools/quick_stats.sh [bench args]  -- GPU box: one rocprofv3 --kernel-trace --stats pass of bench.py --no-cpu-baseline, top kernels
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/quick; mkdir -p gpurun_out/quick
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/quick -o kt -- python bench.py --no-cpu-baseline $* > gpurun_out/quick/bench.log 2>&1
grep '^{' gpurun_out/quick/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.1f M/s  %.4f ms/step  kernel %s %.4f ms  plan %.4f ms' % (d['value']/1e6, d['ms_per_step'], r['kernel'], r['kernel_ms'], (r.get('cells') or r.get('plan'))['kernel_ms']))
print('cells', r.get('cells')); print('census last', r['census']['last'])"
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/quick/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print("%-60s calls %5s avg %9.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
