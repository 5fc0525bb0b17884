#!/bin/bash
# differential fuzz of the final tree, new seeds, three modes (see profiles/r06_fuzz_final2.txt)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_fuzz_final2.txt
echo "# Differential fuzz of the final round-6 tree, tests/fuzz_step.py on one MI355X, new seeds" > $OUT
echo "## cells_forced_seed81" >> $OUT
PFSLAM_PLAN_MIN_N=1 PFSLAM_VARIANT=3 timeout 400 python tests/fuzz_step.py 240 81 2>&1 | grep -v amdgpu.ids | tail -2 >> $OUT
echo "## cells_overflow_seed82" >> $OUT
PFSLAM_PLAN_MIN_N=1 PFSLAM_VARIANT=3 PFSLAM_CELL_LIST_CAP=500 PFSLAM_CELL_POOL_CAP=4000 timeout 300 python tests/fuzz_step.py 150 82 2>&1 | grep -v amdgpu.ids | tail -2 >> $OUT
echo "## default_seed83" >> $OUT
timeout 400 python tests/fuzz_step.py 240 83 2>&1 | grep -v amdgpu.ids | tail -2 >> $OUT
echo "modes: cells_forced = PFSLAM_PLAN_MIN_N=1 PFSLAM_VARIANT=3 (240 s); cells_overflow = the same + PFSLAM_CELL_LIST_CAP=500 PFSLAM_CELL_POOL_CAP=4000 (150 s); default (240 s)" >> $OUT
cat $OUT
