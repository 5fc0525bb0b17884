for e in PFSLAM_CELLS_MODE=1 PFSLAM_CELLS_MODE=0; do
env $e python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['roofline']['census']
print('$e', d['ms_per_step'], d['roofline']['kernel_ms'], 'gathers', c['wave_gathers_per_launch'], 'first', c['first'], 'last', c['last'])"
done
