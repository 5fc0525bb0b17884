#!/bin/bash
# differential fuzz of the final tree, new seeds, three modes (see profiles/r05_fuzz_final2.txt)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_fuzz_final2.txt
echo "# Differential fuzz of the final round-5 tree, tests/fuzz_step.py on one MI355X, new seeds" > $OUT
echo "## cells_forced_seed61" >> $OUT
PFSLAM_PLAN_MIN_N=1 PFSLAM_VARIANT=3 timeout 500 python tests/fuzz_step.py 300 61 2>&1 | grep -v amdgpu.ids | tail -2 >> $OUT
echo "## cells_overflow_seed62" >> $OUT
PFSLAM_PLAN_MIN_N=1 PFSLAM_VARIANT=3 PFSLAM_CELL_LIST_CAP=500 PFSLAM_CELL_POOL_CAP=4000 timeout 400 python tests/fuzz_step.py 200 62 2>&1 | grep -v amdgpu.ids | tail -2 >> $OUT
echo "## default_seed63" >> $OUT
timeout 500 python tests/fuzz_step.py 300 63 2>&1 | grep -v amdgpu.ids | tail -2 >> $OUT
echo "modes: cells_forced = PFSLAM_PLAN_MIN_N=1 PFSLAM_VARIANT=3 (300 s); cells_overflow = the same + PFSLAM_CELL_LIST_CAP=500 PFSLAM_CELL_POOL_CAP=4000 (200 s); default (300 s)" >> $OUT
cat $OUT
