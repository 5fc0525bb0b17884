cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/tools/experiments/r05/libs/libpfslam_base.so
for rep in 1 2 3; do
  for v in base new; do
    if [ $v == base ]; then export PFSLAM_LIB=$L; else unset PFSLAM_LIB; fi
    python tools/frame_probe.py 2>/dev/null | python -c "
import sys,re
t=sys.stdin.read()
print('$v', re.search(r'chain .*', t).group(0)[:70], re.search(r'frame  .*', t).group(0)[:75], re.search(r'violations.: \d+', t).group(0))"
  done
done | tee gpurun_out/ab_part.txt
for rep in 1 2; do for v in base new; do
  if [ $v == base ]; then export PFSLAM_LIB=$L; else unset PFSLAM_LIB; fi
  for n in 100000 1000; do python bench.py --no-cpu-baseline --particles $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['roofline'].get('cells') or {}
print('$v n=$n step %.4f ms' % d['ms_per_step'], {k: c.get(k) for k in ('cells','rows','candidates','cells_without_row','pool_slots','extended','reused')})"; done
done; done | tee -a gpurun_out/ab_part.txt
unset PFSLAM_LIB
timeout 600 python -m pytest tests/test_gpu_frame.py tests/test_gpu_score.py tests/test_gpu_edges.py -m gpu -x -q -k "not long_differential and not fuzz" 2>&1 | tail -3
