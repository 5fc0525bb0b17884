cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for w in 1000000 128; do
  PFSLAM_REDUCE4_WGS=$w python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2957$rep bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sharded world-1 wgs=$w step %.4f ms  %.4e evals/s' % (d['ms_per_step'], d['value']))"
done; done | tee gpurun_out/ab_r4reduce.txt
for rep in 1 2; do for w in 1000000 128; do
  PFSLAM_FRAME_V2=0 PFSLAM_REDUCE4_WGS=$w python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('round-4 frame wgs=$w step %.4f ms' % d['ms_per_step'])"
done; done | tee -a gpurun_out/ab_r4reduce.txt
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_stages.py tests/test_gpu_cfg3.py -m gpu -x -q 2>&1 | tail -3
