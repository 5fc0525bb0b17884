cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do rm -rf gpurun_out/pl; env ${OVENV:-X=1} timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pl -o kt -- python bench.py --no-cpu-baseline > /dev/null 2>&1; python tools/experiments/r04/overlaps.py gpurun_out/pl | cut -c1-${OVCUT:-400}; echo ----; done
