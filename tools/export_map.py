#!/usr/bin/env python3
"""Dump a map for a visual check / end-to-end comparison: the KD nodes with w > -100 as CSV
(x, y, w -- the filter the reference's viewer applies, main.cpp:269-284) and a PGM raster of them.
usage: export_map.py nodes.npy out_prefix        (nodes.npy = array with the KDTree::Node dtype, e.g. PfSlam.map())"""
import sys

import numpy as np


def export(nodes, prefix, res=0.025, extent=20.0):
    keep = nodes["w"] > -100
    pts = np.stack([nodes["x"][keep], nodes["y"][keep], nodes["w"][keep]], 1)
    np.savetxt(prefix + ".csv", pts, fmt="%.4f", header="x y w", comments="")
    dim = int(round(2 * extent / res))
    img = np.full((dim, dim), 127, np.uint8)
    gx = np.clip(np.round((pts[:, 0] + extent) / res).astype(int), 0, dim - 1)
    gy = np.clip(np.round((pts[:, 1] + extent) / res).astype(int), 0, dim - 1)
    img[gx, gy] = np.clip(127 - pts[:, 2], 0, 255).astype(np.uint8)
    with open(prefix + ".pgm", "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (dim, dim))
        f.write(img.tobytes())
    return len(pts)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    print("exported", export(np.load(sys.argv[1]), sys.argv[2]), "points")
