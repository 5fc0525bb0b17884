"""golden_v3.npz: the closed-loop end-to-end run (tests/loop_scenario.py; BASELINE configs[4]) through the CPU oracle, KD frame
loop and 2-D frame loop, with UpdateTopology + CheckLoopClosure at the end of every frame (kernel.cu:1750-1751): per-frame pose
bits / map size / resample flag / loop-closure pairs, the topology graph, and the final map EXPORTS (the point cloud as the
reference's viewer filters it, main.cpp:269-284, and the occupancy grid).  Run here (CPU):  python tests/golden/make_golden_v3.py
The map points are stored as grid-cell integers (every map point is k * 0.025f, kernel.cu:52) -- checked on the way in."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import export_map as EM
import loop_scenario as LS
import oracle_lib as O

pkg = importlib.import_module("gpu-icp-slam_amd")


def pack_records(rec):
    """records -> flat arrays: per frame (pose bits x3, kd size, resampled, first pair index, pair count) + all pairs"""
    frames, pairs = [], []
    for pose, kd, did, pr in rec:
        frames.append(list(pose) + [kd, did, len(pairs), len(pr)])
        pairs.extend(pr)
    return np.asarray(frames, np.int32), np.asarray(pairs, np.int32).reshape(-1, 2)


def run_oracle(grid_path):
    o = O.Slam(LS.N_PARTICLES, kd_capacity=1 << 18)
    mk = lambda p: O.make_particles(LS.N_PARTICLES, float(p[0]), float(p[1]), float(p[2]))
    rec = LS.run(o, mk, LS.scans(pkg), grid_path=grid_path)
    nodes, idx = o.topology()
    pts = EM.kept_points(o.tree()) if o.kd_size else np.zeros((0, 4), np.float32)
    return o, rec, nodes, idx, pts


if __name__ == "__main__":
    out = {}
    for name, grid_path in (("kd", False), ("grid", True)):
        o, rec, nodes, idx, pts = run_oracle(grid_path)
        frames, pairs = pack_records(rec)
        out[name + "_frames"], out[name + "_pairs"] = frames, pairs
        out[name + "_topo"], out[name + "_topo_idx"] = nodes, np.int32(idx)
        if not grid_path:
            cells = np.round(pts[:, :2] / np.float32(0.025)).astype(np.int16)
            assert (cells.astype(np.float32) * np.float32(0.025) == pts[:, :2]).all() and (pts[:, 2] == 0).all()
            assert (pts[:, 3] == np.round(pts[:, 3])).all() and np.abs(pts[:, 3]).max() <= 113
            out["kd_export_cells"], out["kd_export_w"] = cells, pts[:, 3].astype(np.int8)
            out["kd_export_negzero"] = np.packbits(np.signbit(pts[:, :2]) & (pts[:, :2] == 0))  # roundf(-0.3) * res = -0.0f
        else:
            out["grid_export"] = o.grid  # 1600 x 1600 int8; long runs of -100, deflates to ~60 KB
        print(name, "frames", len(rec), "closure frames", int((frames[:, 6] > 0).sum()), "pairs", len(pairs), "nodes", len(nodes),
              "exported", len(pts), "kd", o.kd_size)
        o.close()
    path = os.path.join(ROOT, "tests", "golden", "golden_v3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
