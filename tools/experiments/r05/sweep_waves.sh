cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PFSLAM_TPARTS=0
for n in 1000 2000 4000 10000 20000 40000 100000; do
  for t in 8192 16384 24576 32768 49152 65536 98304; do
    PFSLAM_TARGET_WAVES=$t python bench.py --no-cpu-baseline --particles $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('n=$n target=$t step %.4f ms kernel %.4f ms' % (d['ms_per_step'], r['kernel_ms']))"
  done
done 2>&1 | tee gpurun_out/sweep_waves.txt
