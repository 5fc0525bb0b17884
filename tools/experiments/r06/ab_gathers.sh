#!/bin/bash
# What ONE gather per beam costs the scan-match kernel: timing-only builds (results are WRONG by construction) with the weight gather
# (-DPF_X_NOWEIGHT) or the row's first-slot gather (-DPF_X_NOSLOT0) replaced by arithmetic, against the default build, on ONE saved state
# (tools/experiments/r06/gather_cost.py: no feedback of the wrong scores into the particles).  Is the kernel bound by the gather path (then
# a 16-byte table entry holding the first candidate -- one scattered gather fewer for the ~75 % of the queries whose row is a single
# candidate -- would pay) or by VALU issue?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=tools/experiments/r05/libs
python tools/experiments/r06/gather_cost.py make /tmp/state30.npz 2>/dev/null
for rep in 1 2; do
  for v in base fp32ep; do
    PFSLAM_LIB=$PWD/$L/libpfslam_$v.so python tools/experiments/r06/gather_cost.py time /tmp/state30.npz 2>/dev/null
  done
done
