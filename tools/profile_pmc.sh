#!/bin/bash
# usage: profile_pmc.sh "<counters>" tag   (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rocprofv3 --pmc $1 -d gpurun_out/prof/$2 -o r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline ${3:-} > gpurun_out/prof/$2.log 2>&1
python - <<PY
import sqlite3,glob
for f in glob.glob('gpurun_out/prof/$2/*.db'):
    cur=sqlite3.connect(f).cursor()
    for r in cur.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_score_kd%' group by counter_name"): print(r)
PY
