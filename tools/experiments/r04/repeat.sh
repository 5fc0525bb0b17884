# repeat.sh N "ENV=..." : N consecutive bench runs (same box): kernel ms / frame ms / cells / pool slots / extensions
N=$1; shift; e=${1:-X=1}
for i in $(seq 1 $N); do env $e python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['roofline']['cells']; g=d['roofline']['census']
print('%.4f/%.4f cells %d pool %d ext %d walked %d gathers %.0f' % (d['roofline']['kernel_ms'], d['ms_per_step'], c['cells'], c['pool_slots'], c['extended'], c['walked_from_root'], g['wave_gathers_per_launch']))"; done; echo " <- $e"
