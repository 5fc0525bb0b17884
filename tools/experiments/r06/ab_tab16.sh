#!/bin/bash
# the 16-byte table entries (round 6) against the build before them (libs/libpfslam_base.so), fixed state + bench, alternating
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=tools/experiments/r05/libs
python tools/experiments/r06/gather_cost.py make /tmp/state30.npz 2>/dev/null
for rep in 1 2; do
  PFSLAM_LIB=$PWD/$L/libpfslam_base.so python tools/experiments/r06/gather_cost.py time /tmp/state30.npz 2>/dev/null
  python tools/experiments/r06/gather_cost.py time /tmp/state30.npz 2>/dev/null
done
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export PFSLAM_LIB=$PWD/$L/libpfslam_base.so; else unset PFSLAM_LIB; fi
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v run $rep: kernel_ms %.4f  frame_ms %.4f  steady %.4f  gathers/launch %.3e replay_identical %s' % (r['kernel_ms'], d['ms_per_step'], d['steady_state']['ms_per_step'], r['census']['wave_gathers_per_launch'], r['census']['replay_identical']))"
  done
done
