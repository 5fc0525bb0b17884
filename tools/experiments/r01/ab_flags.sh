#!/bin/bash
# build with extra compile flags on the GPU box and bench (experiments): tools/ab_flags.sh "" "-DFOO"
# (the flags are part of build.py's staleness digest, so they must stay exported while the bench loads the library)
for v in "$@"; do
  export PFSLAM_EXTRA_FLAGS="$v"
  python gpu-icp-slam_amd/build.py > /dev/null 2>&1
  echo "== flags: $v"
  ./tools/ab_variants.sh 0
done
unset PFSLAM_EXTRA_FLAGS
python gpu-icp-slam_amd/build.py > /dev/null 2>&1
