#!/bin/bash
# tools/profile_round.sh <tag> [bench.py workload args...]  -- run ON THE GPU BOX from the repo root (via gpurun).
#   tools/profile_round.sh r02                                          default workload (100 k particles, 100 k-point map)
#   tools/profile_round.sh r02_cfg3 --particles 125000 --map-points 500000   BASELINE configs[3]'s per-GPU share
# 1. rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline <args>`  -> gpurun_out/prof_<tag>/kt
# 2. separate PMC passes (FETCH_SIZE, WRITE_SIZE; SQ / TA sets) of the same command   -> gpurun_out/prof_<tag>/pmc_*
# 3. summaries -> gpurun_out/prof_<tag>/summary/*.csv|json   (copy the ones to be judged into profiles/)
TAG=${1:-r02}; shift
ARGS="$*"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT/summary
CMD="python bench.py --no-cpu-baseline $ARGS"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/bench_kt.log 2>&1
grep '^{' $OUT/bench_kt.log > $OUT/summary/bench_under_kernel_trace.json
# round 4: measured busy cycles of the vector ALUs (SQ_ACTIVE_INST_VALU is in quad-cycles: rocprofiler's own VALUBusy = 4 x it / SIMDs /
# cycles), the SQ's busy cycles, and the fp64 / fp32 / integer split of the VALU instructions; a pass whose counter does not exist on
# this part fails on its own and leaves the others alone
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" \
         "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
         "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32" \
         "SQ_LEVEL_WAVES" "SQ_INST_CYCLES_VALU" "SQ_IFETCH SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- $CMD > $OUT/bench_pmc_$N.log 2>&1
done
python - "$ARGS" <<PY
import csv, glob, json, os, collections, sys, re
out = "$OUT"
args = sys.argv[1]
def arg(name, default):
    m = re.search(name + r"\s+(\d+)", args)
    return int(m.group(1)) if m else default
workload = {"particles": arg("--particles", 100000), "map_points": arg("--map-points", 100000),
            "steps": arg("--steps", 20), "warmup": arg("--warmup", 5)}
# kernel stats
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(out + "/summary/kernel_stats.csv", "w") as g:
        w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
# per-dispatch durations of the score kernel; the census launches (k_score_kd<..., true>) are a different instantiation
timed_ms = all_ms = None
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_score_kd" in r["Kernel_Name"]]
    plain = [r for r in rows if not re.search(r"(<|,\s*)true\s*>", r["Kernel_Name"])]   # the counting instantiations end in <..., true>
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in plain)
    durs = [x[1] / 1e3 for x in d]
    timed = durs[workload["warmup"]:workload["warmup"] + workload["steps"]]
    timed_ms = sum(timed) / max(len(timed), 1) / 1e3
    all_ms = sum(durs) / max(len(durs), 1) / 1e3
    json.dump({"kernel": "k_score_kd", "workload": workload, "launch_us": durs, "mean_us_all": sum(durs) / len(durs),
               "mean_us_timed": timed_ms * 1e3,
               "note": "first %d launches are bench.py's untimed warm-up steps; mean_us_timed = the %d timed ones; the launches behind "
                       "them are bench.py's census replay of the same %d frames (same states, so the same durations) and, without "
                       "--no-cpu-baseline, the long-run leg" % (workload["warmup"], workload["steps"], workload["warmup"] + workload["steps"])},
              open(out + "/summary/score_kd_launches.json", "w"), indent=1)
# PMC per launch of the scan-match kernel: the timed instantiation only, and of its launches only bench.py's TIMED ones --
# dispatches [warmup, warmup + steps) of the process (the warm-up launches before them score a younger map; the launches behind
# them belong to bench.py's census replay of the same frames) -- so that the averages describe exactly the launches the JSON
# line's roofline block is about
pm, pm_all, dur = collections.defaultdict(list), collections.defaultdict(list), {}
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_score_kd" in k and not re.search(r"(<|,\s*)true\s*>", k):
            per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for name, rows in per.items():
        rows.sort()
        # one row per (dispatch, counter instance): sum the instances of a dispatch
        by = collections.OrderedDict()
        for d, v, t in rows:
            by.setdefault(d, [0.0, t])[0] += v
        vals = list(by.values())
        timed = vals[workload["warmup"]:workload["warmup"] + workload["steps"]]
        pm[name] = [v for v, _ in timed]
        pm_all[name] = [v for v, _ in vals]
        dur[name] = sum(t for _, t in timed) / max(len(timed), 1) / 1e6   # ms, in THAT pass (counters perturb durations)
avg = {k: sum(v) / len(v) for k, v in pm.items() if v}
res = {"kernel": "k_score_kd", "workload": workload, "kernel_ms": timed_ms, "kernel_ms_all": all_ms,
       "launches_profiled": {k: len(v) for k, v in pm.items()}, "avg_per_launch": avg,
       "avg_per_launch_all_launches": {k: sum(v) / len(v) for k, v in pm_all.items() if v},
       "pass_kernel_ms": dur}
if "GRBM_GUI_ACTIVE" in avg and dur.get("GRBM_GUI_ACTIVE"):
    res["measured_clock_ghz"] = avg["GRBM_GUI_ACTIVE"] / 8.0 / (dur["GRBM_GUI_ACTIVE"] * 1e-3) / 1e9   # the counter sums the 8 XCDs
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 rocprofv3 reports half the bytes of wide reads (MI355X_MICROARCH.md, HBM)
    res["hbm_bytes_per_launch"] = (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
    res["hbm_bytes_note"] = "(2 x FETCH_SIZE + WRITE_SIZE) KB; the x2 read correction is the guide's and is uncalibrated for 16-B gathers, so this is an upper bound"
res["kernel_ms_note"] = ("kernel_ms = mean duration of the TIMED launches (dispatches [warmup, warmup + steps) of the timed instantiation) in the "
                         "--kernel-trace pass of the same command; avg_per_launch = the same launches in the PMC passes; pass_kernel_ms = "
                         "their mean duration inside each PMC pass (counters perturb durations: use it only with counters of that pass, "
                         "e.g. measured_clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / pass duration)")
json.dump(res, open(out + "/summary/pmc_score_kd.json", "w"), indent=1)
print(json.dumps(res)[:800])
PY
ls -la $OUT/summary
