python -m pytest tests/test_gpu_edges.py -m gpu -q 2>&1 | tail -3
PFSLAM_PLAN_MIN_N=1 PFSLAM_VARIANT=3 timeout 700 python tests/fuzz_step.py 600 21 2>&1 | tail -2
PFSLAM_PLAN_MIN_N=1 PFSLAM_VARIANT=3 PFSLAM_CELL_LIST_CAP=500 PFSLAM_CELL_POOL_CAP=4000 timeout 400 python tests/fuzz_step.py 300 22 2>&1 | tail -2
timeout 400 python tests/fuzz_step.py 300 23 2>&1 | tail -2
