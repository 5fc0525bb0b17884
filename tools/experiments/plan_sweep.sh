python tools/plan_probe.py 2>&1 | grep frame
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_plan2 -o kt -- python bench.py --no-cpu-baseline > gpurun_out/prof_plan2.log 2>&1
python - <<PY
import csv,glob,json
for f in glob.glob("gpurun_out/prof_plan2/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
d=json.loads([l for l in open("gpurun_out/prof_plan2.log") if l.startswith("{")][0]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
