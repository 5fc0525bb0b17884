cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 step %.4f ms kernel %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
run base; run base
for v in 1 2; do
  PFSLAM_EXTRA_FLAGS="-DPF_SETPRIO=$v" python gpu-icp-slam_amd/build.py > /dev/null 2>&1
  PFSLAM_EXTRA_FLAGS="-DPF_SETPRIO=$v" run setprio$v; PFSLAM_EXTRA_FLAGS="-DPF_SETPRIO=$v" run setprio$v
done
python gpu-icp-slam_amd/build.py > /dev/null 2>&1
run base; run base
