#!/bin/bash
# tools/pmc_kernel.sh <kernel-name-substring>: SQ / TA counters of one kernel of the default bench (GPU box)
K=${1:-k_plan}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmck_$K; mkdir -p $OUT
CMD="python bench.py --no-cpu-baseline --steps 6 --warmup 20 $2"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import csv, glob, collections
pm = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$K" in r["Kernel_Name"] and "true>" not in r["Kernel_Name"].replace(" ", ""):
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(pm.items()): print("%-40s %.4g  (n=%d)" % (k, sum(v[-6:]) / len(v[-6:]), len(v)))
PY
