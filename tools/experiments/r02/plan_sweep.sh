python tools/plan_probe.py 2>&1 | grep frame | cut -c1-420
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_plan7 -o kt -- python bench.py --no-cpu-baseline > gpurun_out/prof_plan7.log 2>&1
python - <<PY
import csv,glob,json
for f in glob.glob("gpurun_out/prof_plan7/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
d=json.loads([l for l in open("gpurun_out/prof_plan7.log") if l.startswith("{")][0]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
(PFSLAM_PLAN_MIN_N=1 timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bench_contract.py > gpurun_out/pytest_fork.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_fork.log); tail -4 gpurun_out/pytest_fork.log
