"""gpu-icp-slam_amd -- MI355X-native particle-filter SLAM inner loop (the hot path of
michaelwillett/GPU-ICP-SLAM's src/kernel.cu), as hand-written gfx950 HIP kernels behind a C-ABI.

This Python package is only the test / bench harness around libpfslam_hip.so (ctypes binding,
synthetic-workload generator, build recipe).  The product is the shared library and the C++
`kernel.h`-compatible host layer in host/.  There is no CPU fallback: `load()` raises if the
library cannot be built or loaded.

The directory name contains a hyphen; import it with
    importlib.import_module("gpu-icp-slam_amd")
"""
from .binding import (NODE_DTYPE, PARTICLE_DTYPE, Config, PfSlam, PfSlamError, kd_balance, kd_create,  # noqa: F401
                      kd_insert_node, load, device_count, MgpuRank, mgpu_make_id, load_mgpu)
from . import synth  # noqa: F401
from .build import build  # noqa: F401
