#!/bin/bash
# loop-trip census of the traversal: rebuild with -DPF_EXP_COUNT, run tools/census.py, restore the default build
export PFSLAM_EXTRA_FLAGS="-DPF_EXP_COUNT"
python gpu-icp-slam_amd/build.py > /dev/null 2>&1
python tools/census.py
unset PFSLAM_EXTRA_FLAGS
python gpu-icp-slam_amd/build.py > /dev/null 2>&1
