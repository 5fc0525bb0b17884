cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/tools/experiments/r05/libs/libpfslam_base.so
timeout 600 python -m pytest tests/test_gpu_score.py tests/test_gpu_frame.py tests/test_gpu_edges.py -m gpu -x -q -k "not long_differential" 2>&1 | tail -3
for rep in 1 2 3; do for v in base new; do
  if [ $v == base ]; then export PFSLAM_LIB=$L; else unset PFSLAM_LIB; fi
  for n in 100000; do python bench.py --no-cpu-baseline --particles $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v n=$n step %.4f ms kernel %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done
done; done | tee gpurun_out/ab_wave.txt
for v in base new; do
  if [ $v == base ]; then export PFSLAM_LIB=$L; else unset PFSLAM_LIB; fi
  for n in 1000 10000; do python bench.py --no-cpu-baseline --particles $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v n=$n step %.4f ms kernel %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done
done | tee -a gpurun_out/ab_wave.txt
