#!/bin/bash
# build with extra compile flags on the GPU box and bench (experiments): tools/ab_flags.sh "" "-DFOO"
for v in "$@"; do
  touch gpu-icp-slam_amd/csrc/kd_device.h
  PFSLAM_EXTRA_FLAGS="$v" python gpu-icp-slam_amd/build.py > /dev/null 2>&1
  echo "== flags: $v"
  ./tools/ab_variants.sh 0
done
touch gpu-icp-slam_amd/csrc/kd_device.h
