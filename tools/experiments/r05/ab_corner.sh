cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/tools/experiments/r05/libs/libpfslam_base.so
for rep in 1 2 3; do
  for v in base new; do
    if [ $v == base ]; then export PFSLAM_LIB=$L; else unset PFSLAM_LIB; fi
    python tools/frame_probe.py 2>/dev/null | python -c "
import sys,re
t=sys.stdin.read()
def g(name):
    m=re.search(re.escape(name)+r'\s+\+?(-?[\d.]+)', t); return float(m.group(1))
print('$v', 'cells update %.1f us' % (g('C cells update: last wg ends')-g('C cells update ')), re.search(r'chain .*', t).group(0)[:70], re.search(r'frame  .*', t).group(0)[:75])"
  done
done | tee gpurun_out/ab_corner.txt
for v in base new; do
  if [ $v == base ]; then export PFSLAM_LIB=$L; else unset PFSLAM_LIB; fi
  for n in 100000 1000; do python bench.py --no-cpu-baseline --particles $n --steps 64 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['roofline'].get('cells') or {}
print('$v n=$n step %.4f ms' % d['ms_per_step'], {k: c.get(k) for k in ('cells','rows','candidates','redescent_candidates','pool_slots','extended')})"; done
done | tee -a gpurun_out/ab_corner.txt
unset PFSLAM_LIB
timeout 300 python -m pytest tests/test_gpu_frame.py -m gpu -x -q -k "not long_differential" 2>&1 | tail -3
