#!/bin/bash
# tools/pmc_memory_path.sh <variant> <tag>: TA / TCP / TD / SQ counters of the score kernel, one rocprofv3 --pmc pass per group
# (run ON THE GPU BOX from the repo root).  Output: gpurun_out/pmcmem_<tag>.json = average per k_score_kd launch.
V=${1:-0}; TAG=${2:-v$V}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcmem_$TAG
mkdir -p $OUT
CMD="python bench.py --no-cpu-baseline --steps 4 --warmup 2 --variant $V"
i=0
for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
         "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" "TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $C"
done
python - <<PY
import csv, glob, json, collections
pm = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_score_kd" in r["Kernel_Name"] and not __import__("re").search(r"(<|,\s*)true\s*>", r["Kernel_Name"]):
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in sorted(pm.items())}
json.dump({"variant": $V, "kernel": "k_score_kd", "avg_per_launch": avg, "launches": {k: len(v) for k, v in pm.items()}}, open("gpurun_out/pmcmem_$TAG.json", "w"), indent=1)
for k, v in avg.items(): print("%-44s %.4g" % (k, v))
PY
