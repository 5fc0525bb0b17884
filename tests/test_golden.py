"""Committed golden vectors (tests/golden/golden_v1.npz and golden_v2.npz, made by tests/golden/make_golden*.py).
CPU part: the oracle still reproduces them and the product's host tree code reproduces the REFERENCE's kdtree.cpp
output stored in them (no /root/reference needed at test time).  GPU part: the HIP path reproduces them."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz"))


def nodes(key):
    return np.frombuffer(G[key].tobytes(), dtype=O.NODE_DTYPE).copy()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def particles_from(arr):
    p = O.make_particles(len(arr))
    p["x"], p["y"], p["theta"] = arr[:, 0], arr[:, 1], arr[:, 2]
    if arr.shape[1] > 3:
        p["w"] = arr[:, 3]
    return p


# ------------------------------------------------------------------ CPU
def test_host_tree_code_reproduces_reference_kdtree_output(pkg):
    want = nodes("tree_ref_create")
    assert pkg.kd_create(G["map_pts"]).tobytes() == want.tobytes()
    assert O.kd_create(G["map_pts"]).tobytes() == want.tobytes()
    n = len(want)
    t = np.zeros(n + 40, pkg.NODE_DTYPE)
    t[:n] = want
    for k in range(40):
        pkg.kd_insert_node(t, n + k, G["insert_pts"][k])
    assert t.tobytes() == nodes("tree_ref_inserted").tobytes()
    pkg.kd_balance(t, len(t))
    assert t.tobytes() == nodes("tree_ref_balanced").tobytes()


def test_oracle_reproduces_golden(oracle):
    L = O.lib()
    tree = nodes("tree_ref_create")
    seeds = [L.orc_engine_seed(int(a), int(b), 0) for a, b in zip(G["rng_frames"], G["rng_idx"])]
    assert seeds == G["rng_seeds"].tolist()
    p = particles_from(G["noise_in"])
    O.add_noise(p, int(G["noise_frame"]))
    assert (bits(np.stack([p["x"], p["y"], p["theta"]], 1)) == bits(G["noise_out"])).all()
    best, visits = O.traverse_batch(tree, G["trav_q"])
    assert (best == G["trav_best"]).all() and (visits == G["trav_visits"]).all()
    parts = particles_from(G["score_particles"])
    assert (bits(O.score_kd(tree, parts, G["scan"])) == bits(G["score_fit"])).all()
    pose, dbg = O.icp(tree, G["icp_robot"], G["icp_start"], G["scan"])
    assert (bits(pose) == bits(G["icp_pose"])).all() and (bits(dbg[:28]) == bits(G["icp_dbg"])).all()
    fm, wm = O.get_walls(G["scan"], 800, 800, G["walls_theta"])
    assert (np.flatnonzero(wm) == G["walls_wall_cells"]).all() and (np.flatnonzero(fm) == G["walls_free_cells"]).all()


def test_svd_golden_is_a_valid_decomposition():
    for a, usv in zip(G["svd_a"], G["svd_usv"]):
        A, U, S, V = a.reshape(3, 3), usv[:9].reshape(3, 3), usv[9:18].reshape(3, 3), usv[18:].reshape(3, 3)
        assert np.abs(U @ S @ V.T - A).max() < 2e-4
        assert np.abs(U.T @ U - np.eye(3)).max() < 2e-4 and np.abs(V.T @ V - np.eye(3)).max() < 2e-4
        assert np.abs(np.abs(np.diag(S)) - np.linalg.svd(A.astype(np.float64), compute_uv=False)).max() < 2e-3


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_path_reproduces_golden(pkg):
    assert pkg.device_count() > 0
    tree = nodes("tree_ref_create")
    scan = G["scan"]
    h = pkg.PfSlam(64, kd_capacity=len(tree) + 600)
    h.set_map(tree)
    h.set_scan(scan)
    assert (h.traverse(G["trav_q"]) == G["trav_best"]).all()
    parts = particles_from(G["score_particles"])
    h.set_particles(parts)
    assert (bits(h.score_kd()) == bits(G["score_fit"])).all()
    best, _, _ = h.measurement_update()
    assert best == int(G["meas_best"]) and (bits(h.particles()["w"]) == bits(G["meas_w"])).all()
    h.set_pose(G["icp_robot"])
    pose, dbg = h.icp(G["icp_start"])
    assert (bits(pose) == bits(G["icp_pose"])).all() and (bits(dbg[:28]) == bits(G["icp_dbg"])).all()
    assert np.abs(pose - G["icp_pose"]).max() <= 1e-4
    h.set_pose(G["icp_robot"])
    h.update_map_kd()
    assert (h.cells(0) == G["walls_wall_cells"]).all() and (h.cells(1) == G["walls_free_cells"]).all()
    assert h.map().tobytes() == nodes("mapupd_tree").tobytes()
    # grid path
    grid = np.full(1600 * 1600, -100, np.int8)
    grid[G["grid_cells"]] = G["grid_vals"]
    h.set_grid(grid.reshape(1600, 1600))
    h.set_particles(parts)
    assert (h.score_grid() == G["grid_fit"]).all()
    h.set_pose(G["icp_robot"])
    h.update_map_grid()
    want = grid.copy()
    want[G["grid_upd_cells"]] = G["grid_upd_vals"]
    assert (h.grid().ravel() == want).all()
    h.close()
    # dispersion
    h2 = pkg.PfSlam(256)
    h2.set_particles(particles_from(G["noise_in"]))
    h2.motion_update(int(G["noise_frame"]))
    g = h2.particles()
    assert (bits(np.stack([g["x"], g["y"], g["theta"]], 1)) == bits(G["noise_out"])).all()
    h2.close()
    # resample
    n = len(G["resample_w"])
    rp = O.make_particles(n)
    rp["x"] = np.arange(n)
    rp["w"] = G["resample_w"]
    h3 = pkg.PfSlam(n)
    h3.set_particles(rp)
    did, neff = h3.resample(int(G["resample_frame"]))
    assert did == 1 and np.float32(neff) == G["resample_neff"]
    assert (h3.particles()["x"].astype(np.int32) == G["resample_src"]).all()
    h3.close()


@pytest.mark.gpu
def test_hip_step_replay_reproduces_golden(pkg):
    h = pkg.PfSlam(200, kd_capacity=1 << 16)
    for f, scan in enumerate(G["replay_scans"], start=1):
        h.step(f, scan)
        t = h.trace()
        row = [t["best"], t["resampled"], t["n_wall"], t["n_free"], t["n_insert"], t["kd_size"]] + h.pose.view(np.int32).tolist()
        assert row == G["replay_trace"][f - 1].tolist(), f
    p = h.particles()
    assert (bits(np.stack([p["x"], p["y"], p["theta"], p["w"]], 1)) == bits(G["replay_particles"])).all()
    h.close()


# ---- golden_v2: the 2-D frame loop and the topology graph (rows added after golden_v1) ----------------------
def _v2():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v2.npz"))


def _check_grid_replay(step, trace, pose, grid, particles, G):
    import zlib
    scans = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz"))["replay_scans"]
    for f, scan in enumerate(scans, start=1):
        step(f, scan)
        t = trace()
        row = [t["best"], t["resampled"], int(np.float32(t["neff"]).view(np.int32))] + [int(v) for v in pose().view(np.int32)]
        assert row == G["grid_replay_trace"][f - 1].tolist(), f
    g = grid()
    assert zlib.crc32(g.tobytes()) == int(G["grid_replay_crc"]) and int((g < -100).sum()) == int(G["grid_replay_free_count"])
    cells = np.flatnonzero(g.ravel() > -100)
    assert (cells == G["grid_replay_wall_cells"]).all() and (g.ravel()[cells] == G["grid_replay_wall_vals"]).all()
    p = particles()
    got = np.stack([p["x"], p["y"], p["theta"], p["w"]], 1)
    assert (got.view(np.int32) == G["grid_replay_particles"].view(np.int32)).all()


def test_oracle_reproduces_golden_v2(oracle):
    G = _v2()
    o = O.Slam(200)
    _check_grid_replay(o.step_grid, o.trace, lambda: o.pose, lambda: o.grid, o.particles, G)
    o.close()
    t = O.Topology()
    created = [t.update(r) for r in G["topo_path"]]
    assert created == G["topo_created"].tolist() and (t.nodes().view(np.int32) == G["topo_nodes"].view(np.int32)).all()
    grid = np.full(1600 * 1600, -100, np.int8)
    grid[G["topo_grid_cells"]] = 113
    grid = grid.reshape(1600, 1600)
    assert (t.loop_closure(grid, G["topo_path"][-1]) == G["topo_pairs"]).all() and len(G["topo_pairs"]) > 0
    assert O.find_walls(grid, G["topo_walls_a"], G["topo_walls_b"]) == int(G["topo_walls_n"]) == 2


@pytest.mark.gpu
def test_hip_path_reproduces_golden_v2(pkg):
    assert pkg.device_count() > 0
    G = _v2()
    h = pkg.PfSlam(200)
    _check_grid_replay(h.step_grid, h.trace, lambda: h.pose, h.grid, h.particles, G)
    grid = np.full(1600 * 1600, -100, np.int8)
    grid[G["topo_grid_cells"]] = 113
    h.set_grid(grid.reshape(1600, 1600))
    created, n_prev = [], 1          # the graph starts with the origin node (particleFilterInit)
    for r in G["topo_path"]:
        h.set_pose(r)
        n = h.topology_update()
        created.append(n - n_prev)
        n_prev = n
    assert created == G["topo_created"].tolist()
    nodes, node_idx = h.topology()
    assert (np.asarray(nodes, np.float32).view(np.int32) == G["topo_nodes"].view(np.int32)).all()
    assert (h.check_loop_closure() == G["topo_pairs"]).all()
    assert h.find_walls(G["topo_walls_a"], G["topo_walls_b"]) == int(G["topo_walls_n"])
    h.close()
