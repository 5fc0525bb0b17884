"""End-to-end loop-closure run and map export (BASELINE configs[4]; SURVEY 8f #2 topology / loop closure, #4 map export):
260 frames of a closed synthetic loop (tests/loop_scenario.py) through the frame loop with UpdateTopology + CheckLoopClosure
at the end of every frame, where the reference has them commented out (kernel.cu:1750-1751).
  CPU:  the oracle reproduces the committed golden fixture (tests/golden/golden_v3.npz, made by make_golden_v3.py) -- per-frame
        pose bits, map size, resample flags, loop-closure pairs, the topology graph and the final exported maps.
  GPU:  the product reproduces the same fixture through the C-ABI (pfslam_set_topology / pfslam_get_closures / pfslam_step /
        pfslam_step_grid), and its exported maps equal the oracle's cell for cell (tools/export_map.py, the filter of the
        reference's viewer main.cpp:269-284); host/pfslam_replay's `loop export=` mode writes the same files."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import loop_scenario as LS
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import export_map as EM
import make_golden_v3 as G3

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "golden_v3.npz"))
HOST = os.path.join(ROOT, "gpu-icp-slam_amd", "host")


def golden_points():
    cells, w = GOLD["kd_export_cells"], GOLD["kd_export_w"]
    pts = np.zeros((len(cells), 4), np.float32)
    pts[:, :2] = cells.astype(np.float32) * np.float32(0.025)
    negzero = np.unpackbits(GOLD["kd_export_negzero"])[:2 * len(cells)].reshape(-1, 2).astype(bool)
    pts[:, :2][negzero] = np.float32(-0.0)
    pts[:, 3] = w.astype(np.float32)
    return pts


def check_against_golden(name, rec, nodes, idx):
    frames, pairs = G3.pack_records(rec)
    want = GOLD[name + "_frames"]
    bad = np.flatnonzero((frames != want).any(1))
    assert len(bad) == 0, "%s: frame %d differs: got %s want %s" % (name, bad[0] + 1, frames[bad[0]], want[bad[0]])
    assert (pairs == GOLD[name + "_pairs"]).all()
    assert idx == int(GOLD[name + "_topo_idx"]) and (np.asarray(nodes, np.float32).view(np.int32) == GOLD[name + "_topo"].view(np.int32)).all()


def test_golden_fixture_is_a_real_loop_closure_run():
    for name in ("kd", "grid"):
        fr = GOLD[name + "_frames"]
        assert len(fr) == LS.N_FRAMES >= 200
        assert (fr[:, 6] > 0).sum() > 50 and len(GOLD[name + "_pairs"]) > 500     # many frames propose closures ...
        assert (fr[:150, 6] == 0).all()                                            # ... but only once some node is > 20 m away along the graph
        assert len(GOLD[name + "_topo"]) == 9 and fr[:, 4].sum() > 20              # 8 nodes around the 24 m square + the origin node
    assert len(GOLD["kd_export_cells"]) > 20000 and (GOLD["grid_export"] != -100).sum() > 400000


@pytest.mark.parametrize("name", ["kd", "grid"])
def test_oracle_reproduces_the_golden_loop_run(pkg, name):
    o, rec, nodes, idx, pts = G3.run_oracle(name == "grid")
    check_against_golden(name, rec, nodes, idx)
    if name == "kd":
        assert (pts.view(np.int32) == golden_points().view(np.int32)).all()
    else:
        assert (o.grid == GOLD["grid_export"]).all()
    o.close()


def test_export_failure_paths(tmp_path):
    nodes = np.zeros(4, O.NODE_DTYPE)
    nodes["w"] = [-100, -99, 5, -100.5]
    nodes["x"] = [1, 2, 3, 4]
    assert EM.export(nodes, np.full((4, 4), -100, np.int8), str(tmp_path / "a")) == 2          # w > -100 only
    assert np.fromfile(str(tmp_path / "a.kd.bin"), np.float32).reshape(-1, 4)[:, 0].tolist() == [2.0, 3.0]
    with pytest.raises(ValueError):
        EM.export(np.zeros(3, np.float32), None, str(tmp_path / "b"))                             # not a node array
    with pytest.raises(ValueError):
        EM.export(nodes, np.zeros((4, 4), np.float32), str(tmp_path / "b"))                       # grid must be int8
    EM.export(nodes, np.full((4, 4), -100, np.int8), str(tmp_path / "c"))
    assert all(ok for ok, _ in EM.compare(str(tmp_path / "a"), str(tmp_path / "c")).values())
    g = np.full((4, 4), -100, np.int8); g[1, 2] = 7
    nodes["w"][2] = 6
    EM.export(nodes, g, str(tmp_path / "d"))
    res = EM.compare(str(tmp_path / "a"), str(tmp_path / "d"))
    assert res[".kd.bin"] == (False, "1 differing values of 8") and res[".grid.i8"] == (False, "1 differing values of 16")
    assert EM.compare(str(tmp_path / "a"), str(tmp_path / "missing"))[".kd.bin"] == (False, "missing file")
    EM.export(nodes[:3], None, str(tmp_path / "e"))
    assert EM.compare(str(tmp_path / "a"), str(tmp_path / "e"))[".kd.bin"][0] is False            # different point count


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kd", "grid"])
def test_product_loop_run_matches_golden_and_oracle_export(pkg, tmp_path, name):
    assert pkg.device_count() > 0
    h = pkg.PfSlam(LS.N_PARTICLES, kd_capacity=1 << 18)
    mk = lambda p: O.make_particles(LS.N_PARTICLES, float(p[0]), float(p[1]), float(p[2]))
    rec = LS.run(h, mk, LS.scans(pkg), grid_path=(name == "grid"))
    nodes, idx = h.topology()
    check_against_golden(name, rec, nodes, idx)
    # exported maps: product vs golden (= oracle), cell for cell
    n = EM.export(h.map() if name == "kd" else np.zeros(0, pkg.NODE_DTYPE), h.grid(), str(tmp_path / "prod"))
    gold_nodes = np.zeros(len(GOLD["kd_export_cells"]) if name == "kd" else 0, O.NODE_DTYPE)
    if name == "kd":
        gp = golden_points()
        gold_nodes["x"], gold_nodes["y"], gold_nodes["w"] = gp[:, 0], gp[:, 1], gp[:, 3]
    gold_grid = GOLD["grid_export"] if name == "grid" else np.full((1600, 1600), -100, np.int8)
    EM.export(gold_nodes, gold_grid, str(tmp_path / "gold"))
    res = EM.compare(str(tmp_path / "prod"), str(tmp_path / "gold"))
    assert all(ok for ok, _ in res.values()), res
    assert n == len(gold_nodes)
    h.close()


@pytest.mark.gpu
def test_replay_driver_loop_and_export_mode(pkg, tmp_path):
    """host/pfslam_replay ... loop export=PREFIX (no re-centring of the cloud: the plain frame loop) prints the frame's
    loop-closure proposals and writes the export files; both equal a Python-driven handle with pfslam_set_topology."""
    assert pkg.device_count() > 0
    pkg.load()
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    scans = np.stack(LS.scans(pkg, n_frames=24)).astype(np.float32)
    lidar = tmp_path / "lidar.f32"
    np.concatenate([np.zeros((1, 1081), np.float32), scans]).tofile(str(lidar))      # scans[0] is never used (frames start at 1)
    scene = tmp_path / "scene.txt"
    scene.write_text("CAMERA\nRES 800 800\nFOVY 45\nFILE map0\nEYE 0 0 25\nLOOKAT 0 0 0\nUP 0 1 0\n\nMAP\nSIZE 40 40\nRES .025\n")
    env = dict(os.environ, PFSLAM_PARTICLES="200", PFSLAM_KD_CAPACITY=str(1 << 17))
    for mode in ((), ("grid",)):
        prefix = str(tmp_path / ("replay_" + "_".join(mode or ("kd",))))
        out = subprocess.check_output([os.path.join(HOST, "pfslam_replay"), str(scene), str(lidar), "0", "loop", "export=" + prefix] + list(mode),
                                      env=env).decode()
        lines = [l for l in out.splitlines() if l.startswith("frame ")]
        assert len(lines) == 24 and all(" closures " in l for l in lines)
        h = pkg.PfSlam(200, kd_capacity=1 << 17)
        h.set_topology(True)
        for f in range(1, 25):
            (h.step_grid if mode else h.step)(f, scans[f - 1])
            tok = lines[f - 1].split()
            assert [int(tok[k], 16) for k in (7, 8, 9)] == h.pose.view(np.uint32).tolist(), lines[f - 1]
            assert int(tok[tok.index("closures") + 1]) == len(h.closures())
        n = EM.export(h.map() if h.kd_size else np.zeros(0, pkg.NODE_DTYPE), h.grid(), prefix + "_py")
        assert ("exported %d map points" % n) in out
        res = EM.compare(prefix, prefix + "_py")
        assert all(ok for ok, _ in res.values()), res
        assert open(prefix + ".kd.csv").read() == open(prefix + "_py.kd.csv").read()
        assert open(prefix + ".grid.pgm", "rb").read() == open(prefix + "_py.grid.pgm", "rb").read()
        h.close()
