R="python tools/experiments/r05/replay_case.py tools/experiments/r05/fuzz_case_n2049.npz 2"
export PFSLAM_PLAN_MIN_N=1
echo "== main"; $R 2>&1 | tail -1
echo "== main SCAN_GATE=0"; env PFSLAM_SCAN_GATE=0 $R 2>&1 | tail -1
