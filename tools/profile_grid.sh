#!/bin/bash
# tools/profile_grid.sh <tag>  -- ON THE GPU BOX: the 2-D grid path (A17 / A18) under rocprofv3: kernel stats + FETCH_SIZE / WRITE_SIZE of
# k_score_grid and k_update_grid -> gpurun_out/prof_<tag>_grid/summary
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_${TAG}_grid; mkdir -p $OUT/summary
python tools/grid_bench.py --out $OUT/summary/grid_bench.json > $OUT/grid_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python tools/grid_bench.py > $OUT/kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python tools/grid_bench.py --counts 1000000 > $OUT/pmc_$N.log 2>&1
done
python - <<PY
import csv, glob, json, collections
out = "$OUT"
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f))]
    with open(out + "/summary/kernel_stats_grid.csv", "w") as g:
        w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
pm = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_score_grid") or k.startswith("k_update_grid"):
            per[(k, r["Counter_Name"])][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c), d in per.items():
        v = list(d.values())
        pm[k][c] = sum(v) / len(v)
res = {k: dict(v) for k, v in pm.items()}
for k, v in res.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
json.dump({"workload": "tools/grid_bench.py --counts 1000000 (1 M particles, 1600 x 1600 grid)", "per_launch": res}, open(out + "/summary/pmc_grid.json", "w"), indent=1)
print(json.dumps(res)[:600])
PY
cat $OUT/grid_bench.log | tail -3
