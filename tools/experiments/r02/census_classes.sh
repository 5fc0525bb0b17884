for args in "" "--particles 125000 --map-points 500000"; do python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['roofline']['gathers']
for k in ('census_before','census_after'):
    c=g[k]; v=c['visits']; print(k, 'visits/eval %.2f trips %d tests %d  redescent-visits %.3f  first-descent: 1-child %.3f  2-child %.3f  leaf/z %.3f'%(v/(d['config']['particles_global']*1081), c['trips'], c['tests'], c['uniform_trips']/v, c['prefix_trips']/v, c['redescents_noop']/v, 1-(c['uniform_trips']+c['prefix_trips']+c['redescents_noop'])/v))
"; done
