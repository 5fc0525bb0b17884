bash tools/profile_round.sh r04 2>&1 | tail -2
python bench.py --dry-collectives > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
bash tools/timeline.sh > gpurun_out/r04_timeline.txt 2>&1
bash tools/profile_round.sh r04_cfg3 --particles 125000 --map-points 500000 2>&1 | tail -2
python bench.py --particles 125000 --map-points 500000 > gpurun_out/r04_bench_cfg3.json 2>/dev/null
bash tools/profile_grid.sh r04 2>&1 | tail -4
for n in 1000 10000 50000 250000 1000000; do python bench.py --no-cpu-baseline --particles $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('particles $n: step %.4f ms  %.4e evals/s  kernel %.4f ms' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))"; done | tee gpurun_out/r04_particle_counts.txt
head -c 300 gpurun_out/r04_bench.json
