#!/usr/bin/env python3
"""Map export for a visual check / end-to-end comparison, same files as host/pfslamExportMap:
  PREFIX.kd.bin   float32 (x, y, z, w) of every KD node with w > -100, in node order -- the filter the reference's viewer
                  applies (main.cpp:269-284);  PREFIX.kd.csv the same as text;  PREFIX.kd.pgm a raster of them
  PREFIX.grid.i8  the 2-D occupancy grid, dim.x * dim.y signed bytes, cell (x, y) at x * dim.x + y (kernel.cu:120, 539)
  PREFIX.grid.pgm the same as an image (value + 128, one row per x)
usage: export_map.py nodes.npy [grid.npy] out_prefix    (nodes.npy: KDTree::Node dtype, e.g. PfSlam.map(); grid.npy: int8 2-D)
`export(nodes, grid, prefix)` returns the number of exported points; compare(a_prefix, b_prefix) checks two exports
cell for cell."""
import os
import sys

import numpy as np


def kept_points(nodes):
    """(k, 4) float32 x, y, z, w of the nodes the reference's viewer draws (w > -100), in node order."""
    for fld in ("x", "y", "z", "w"):
        if fld not in (nodes.dtype.names or ()):
            raise ValueError("nodes must have the KDTree::Node fields (axis, left, right, parent, x, y, z, w)")
    keep = nodes["w"] > -100
    return np.stack([nodes[f][keep] for f in ("x", "y", "z", "w")], 1).astype(np.float32)


def export(nodes, grid, prefix, res=0.025, extent=20.0):
    pts = kept_points(nodes)
    pts.tofile(prefix + ".kd.bin")
    with open(prefix + ".kd.csv", "w") as f:
        f.write("x y z w\n")
        for p in pts:
            f.write("%.9g %.9g %.9g %.9g\n" % tuple(float(v) for v in p))
    dim = int(round(2 * extent / res))
    img = np.full((dim, dim), 127, np.uint8)
    gx = np.clip(np.round((pts[:, 0] + extent) / res).astype(int), 0, dim - 1)
    gy = np.clip(np.round((pts[:, 1] + extent) / res).astype(int), 0, dim - 1)
    img[gx, gy] = np.clip(127 - pts[:, 3], 0, 255).astype(np.uint8)
    with open(prefix + ".kd.pgm", "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (dim, dim))
        f.write(img.tobytes())
    if grid is not None:
        grid = np.asarray(grid)
        if grid.dtype != np.int8 or grid.ndim != 2:
            raise ValueError("grid must be a 2-D int8 array (MAP_TYPE = char)")
        grid.tofile(prefix + ".grid.i8")
        with open(prefix + ".grid.pgm", "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (grid.shape[1], grid.shape[0]))
            f.write((grid.astype(np.int16) + 128).astype(np.uint8).tobytes())
    return len(pts)


def compare(a_prefix, b_prefix):
    """Cell-for-cell comparison of two exports: {suffix: (equal, detail)}."""
    out = {}
    for sfx, dt in ((".kd.bin", np.float32), (".grid.i8", np.int8)):
        fa, fb = a_prefix + sfx, b_prefix + sfx
        if not (os.path.exists(fa) and os.path.exists(fb)):
            out[sfx] = (False, "missing file")
            continue
        a, b = np.fromfile(fa, dtype=dt), np.fromfile(fb, dtype=dt)
        if a.shape != b.shape:
            out[sfx] = (False, "sizes %d vs %d" % (a.size, b.size))
        else:
            diff = int((a.view(np.int32 if dt == np.float32 else np.int8) != b.view(np.int32 if dt == np.float32 else np.int8)).sum())
            out[sfx] = (diff == 0, "%d differing values of %d" % (diff, a.size))
    return out


if __name__ == "__main__":
    if len(sys.argv) not in (3, 4):
        raise SystemExit(__doc__)
    nodes = np.load(sys.argv[1])
    grid = np.load(sys.argv[2]) if len(sys.argv) == 4 else None
    print("exported", export(nodes, grid, sys.argv[-1]), "points")
