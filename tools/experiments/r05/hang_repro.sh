cd $GRAFT_REPO_ROOT
export PFSLAM_BENCH_WATCHDOG=80
for i in 1 2 3; do
  timeout 200 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q -k "contract_fields or torchrun_rccl" > /tmp/p.log 2>&1; rc=$?
  echo "pytest $i rc=$rc"; tail -3 /tmp/p.log | cut -c1-200
  if [ $rc -ne 0 ]; then grep -n "File \"/\|Thread\|most recent" /tmp/p.log | grep -v "dist-packages\|/usr/lib" | tail -40 | cut -c1-220; fi
done
