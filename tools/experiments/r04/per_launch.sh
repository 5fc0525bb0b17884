# per-launch durations of the scan-match kernel over the timed window, under the kernel trace, for an environment setting
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for e in "X=1" "PFSLAM_CELLS_MODE=1"; do
rm -rf gpurun_out/pl; env $e timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pl -o kt -- python bench.py --no-cpu-baseline > /dev/null 2>&1
python - "$e" <<'PY'
import csv, glob, sys
rows=[]
for f in glob.glob("gpurun_out/pl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_score_kd_cells<false" in r["Kernel_Name"]: rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
rows.sort(); d=[v for _,v in rows][5:25]
print(sys.argv[1], "mean %.1f" % (sum(d)/len(d)), " ".join("%.0f"%v for v in d))
PY
done
