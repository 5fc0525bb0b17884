# same box, alternating: this build (asynchronous / synchronous cell passes) against the round-3 build in .r03cmp (git worktree add .r03cmp 447e291)
for rep in 1 2 3; do
for cfg in ".:-" ".:PFSLAM_CELLS_MODE=1" ".r03cmp:-"; do
d=${cfg%%:*}; e=${cfg##*:}; [ "$e" == "-" ] && e="X=1"
(cd $d; env $e python bench.py --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg: step %.4f ms  %.4e evals/s  kernel %.4f ms' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))")
done; done
