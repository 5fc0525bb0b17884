#!/bin/bash
# publishing pass without its corner tests (-DPF_CU_NOCORNER=1: rows list more candidates, results unchanged) against the default build:
# an UPPER BOUND of what spreading the quadratic corner tests over a wave's idle lanes could take off k_cells_update<true>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=tools/experiments/r05/libs
for rep in 1 2; do
  for v in base nocorner; do
    for n in 100000 1000; do
      echo "== $v particles $n (run $rep)"
      PFSLAM_LIB=$PWD/$L/libpfslam_$v.so python tools/frame_probe.py --particles $n 2>/dev/null | grep -E "^C cells update|^C scan-match|^frame|^chain|^scan-match kernel|candidates"
    done
  done
done
