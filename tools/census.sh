#!/bin/bash
touch gpu-icp-slam_amd/csrc/kd_device.h
PFSLAM_EXTRA_FLAGS="-DPF_EXP_COUNT" python gpu-icp-slam_amd/build.py > /dev/null 2>&1
python tools/census.py
touch gpu-icp-slam_amd/csrc/kd_device.h
