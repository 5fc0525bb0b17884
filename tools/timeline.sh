#!/bin/bash
# tools/timeline.sh [bench.py args]  -- ON THE GPU BOX: where does a frame's wall time go?  One rocprofv3 --kernel-trace pass of
# bench.py, then per frame (k_motion .. next k_motion) of the timed window: busy time per kernel, idle gaps and what follows them.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/timeline; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-300
python tools/timeline.py $OUT/kt | tee $OUT/timeline.txt
