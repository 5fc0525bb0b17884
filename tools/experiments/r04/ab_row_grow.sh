for rep in 1 2 3; do for g in 0 1 4; do
PFSLAM_EXTRA_FLAGS="-DPF_ROW_GROW=$g" PFSLAM_CELLS_MODE=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['roofline']['cells']; print('sync, PF_ROW_GROW=$g: step %.4f ms kernel %.4f ms pool %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], c['pool_slots']))"
done; done
