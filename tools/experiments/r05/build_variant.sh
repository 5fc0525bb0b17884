#!/bin/bash
# tools/experiments/r05/build_variant.sh NAME "-DPF_X=0 ..."  -> tools/experiments/r05/libs/libpfslam_NAME.so (use with PFSLAM_LIB=...)
cd "$(dirname "$0")/../../.."
mkdir -p tools/experiments/r05/libs /tmp/pfv_$1
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Wno-unused-result $2"
/opt/rocm/bin/hipcc $F -c gpu-icp-slam_amd/csrc/pfslam_hip.hip -o /tmp/pfv_$1/a.o && /opt/rocm/bin/hipcc $F -c gpu-icp-slam_amd/csrc/kd_host.cpp -o /tmp/pfv_$1/b.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/pfv_$1/a.o /tmp/pfv_$1/b.o -lpthread -o tools/experiments/r05/libs/libpfslam_$1.so && echo built $1
