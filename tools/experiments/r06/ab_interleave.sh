#!/bin/bash
# publishing pass: a wave's 64 records taken W apart (-DPF_CU_INTERLEAVE) instead of side by side -- do the records an insert touches
# cluster in a few waves (neighbours in claim order), and is the pass's slowest wave one of those?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=tools/experiments/r05/libs
for rep in 1 2 3; do
  for v in base interleave; do
    for n in 100000 1000; do
      echo "== $v particles $n (run $rep)"
      PFSLAM_LIB=$PWD/$L/libpfslam_$v.so python tools/frame_probe.py --particles $n 2>/dev/null | grep -E "^C cells update|^frame|^chain|^scan-match kernel|violations"
    done
  done
done
