cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r06f > gpurun_out/profile_round.log 2>&1
ls gpurun_out/prof_r06f/summary/
bash tools/timeline.sh > gpurun_out/timeline_r06f.txt 2>&1
python tools/frame_probe.py 2>/dev/null | grep -v "^cell stats" > gpurun_out/frame_probe_r06f.txt
python tools/frame_probe.py --particles 1000 2>/dev/null | grep -v "^cell stats" > gpurun_out/frame_probe_1k_r06f.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_r06f.json
python bench.py --particles 1000 2>/dev/null | tail -1 > gpurun_out/bench_1k_r06f.json
python bench.py --particles 10000 2>/dev/null | tail -1 > gpurun_out/bench_10k_r06f.json
python bench.py --particles 125000 --map-points 500000 2>/dev/null | tail -1 > gpurun_out/bench_cfg3_r06f.json
for f in gpurun_out/bench_r06f.json gpurun_out/bench_1k_r06f.json gpurun_out/bench_10k_r06f.json gpurun_out/bench_cfg3_r06f.json; do python -c "
import json,sys
d=json.loads(open('$f').read()); r=d['roofline']
print('$f', 'value %.4e' % d['value'], 'ms %.4f' % d['ms_per_step'], 'long', d.get('value_long_run'), 'kernel_ms %.4f' % r['kernel_ms'], 'frac %.3f' % r['frac'], 'chain', (d.get('frame') or {}).get('chain_us_mean'))"; done
