# frame time with the cell rows from any particle count (PFSLAM_PLAN_MIN_N=1) against the default threshold (4608)
for n in 500 1000 2000 3000 5000; do for e in "X=1" "PFSLAM_PLAN_MIN_N=1"; do
env $e python bench.py --no-cpu-baseline --particles $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('n $n $e: step %.4f ms  kernel %.4f ms  %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel']))"
done; done
