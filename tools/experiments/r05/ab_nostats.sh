cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in base nostats; do
    export PFSLAM_LIB=$GRAFT_REPO_ROOT/tools/experiments/r05/libs/libpfslam_$v.so
    python tools/frame_probe.py 2>/dev/null | python -c "
import sys,re
t=sys.stdin.read()
def g(name):
    m=re.search(re.escape(name)+r'\s+\+?(-?[\d.]+)', t); return float(m.group(1))
print('$v', 'cells update %.1f us' % (g('C cells update: last wg ends')-g('C cells update ')), re.search(r'chain .*', t).group(0)[:70], re.search(r'frame  .*', t).group(0)[:75])"
  done
done | tee gpurun_out/ab_nostats.txt
