#!/usr/bin/env python3
"""Convert the reference's train_lidar*.mat (MATLAB v5: cell array `lidar`, each cell a struct with a single[1081]
field `scan` -- src/lidar.cpp:17-49) into the flat little-endian float32 file `frames x 1081` that
gpu-icp-slam_amd/host/lidar.h reads.   usage: mat2bin.py train_lidar0.mat out.f32"""
import sys

import numpy as np


def main():
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    from scipy.io import loadmat
    m = loadmat(sys.argv[1], squeeze_me=True, struct_as_record=False)
    cells = np.atleast_1d(m["lidar"])
    scans = []
    for c in cells:
        s = getattr(c, "scan", None)
        if s is not None:
            scans.append(np.asarray(s, dtype=np.float32).ravel())
    out = np.stack(scans).astype("<f4")
    if out.shape[1] != 1081:
        print("warning: %d beams per scan, expected 1081" % out.shape[1], file=sys.stderr)
    out.tofile(sys.argv[2])
    print("wrote %d scans x %d beams" % out.shape)


if __name__ == "__main__":
    main()
