"""BASELINE configs[4] for real: the closed loop FREE-RUNNING at 100 000 particles (tests/loop_scenario.py run_free).  The cloud
is never re-centred: the commanded motion of every frame reaches the filter as an odometry shift of every pose it holds
(pfslam_shift_particles -- the reference's filter has no motion model besides its 1.5 cm diffusion), everything else is the frame
loop with UpdateTopology + CheckLoopClosure inside (kernel.cu:1750-1751).
  CPU:  the committed fixture tests/golden/golden_v4.npz (oracle, made by make_golden_v4.py) is a real run -- the filter tracks the
        24 m drive, the loop closes -- and the oracle reproduces its first frames.
  GPU:  the product reproduces the fixture frame by frame (pose bits, map size, resample flags, closure pairs), the topology graph,
        the final particle arrays (CRC) and the exported maps cell for cell; with the topology calls booked one frame late
        (pfslam_set_topology 2) the graph and the final frame's proposals are the same."""
import importlib
import os
import sys

import numpy as np
import pytest

import loop_scenario as LS
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import export_map as EM
import make_golden_v4 as G4
from make_golden_v3 import pack_records

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "golden_v4.npz"))


def test_golden_v4_is_a_tracked_closed_loop():
    for name in ("kd", "grid"):
        fr = GOLD[name + "_frames"]
        assert len(fr) == LS.N_FRAMES
        err = GOLD[name + "_track_err"]
        # free-running: the estimate stays with the drive all the way round the 24 m square (no re-centring anywhere).  The 2-D loop
        # tracks to 7 cm; the KD loop -- pose = best particle + ICP increment, the map grown at that pose -- drifts to 1.6 m
        assert err.max() < (2.0 if name == "kd" else 0.2) and err[:60].max() < 0.8, (name, float(err.max()), float(err[-1]))
        assert fr[:, 4].sum() > 20                                    # resamples: the cloud lives on from frame to frame
        assert (fr[:, 6] > 0).sum() > 20 and len(GOLD[name + "_pairs"]) > 100 and (fr[:150, 6] == 0).all()   # the loop closes, late
        assert len(GOLD[name + "_topo"]) >= 8
    assert len(GOLD["kd_export_cells"]) > 20000 and (GOLD["grid_export"] != -100).sum() > 400000


def test_oracle_reproduces_the_first_frames_of_golden_v4(pkg, monkeypatch):
    """The whole fixture is a quarter of an hour of oracle time; its first 6 frames at the full 100 000 particles pin it here."""
    monkeypatch.setenv("ORC_THREADS", str(min(16, len(os.sched_getaffinity(0)))))
    n_frames = 6
    o, rec, nodes, idx, pts = G4.run_oracle(False, n_frames=n_frames)
    frames, _ = pack_records(rec)
    assert (frames[:, :5] == GOLD["kd_frames"][:n_frames, :5]).all()
    o.close()


def _check(name, rec, nodes, idx):
    frames, pairs = pack_records(rec)
    want = GOLD[name + "_frames"]
    bad = np.flatnonzero((frames != want).any(1))
    assert len(bad) == 0, "%s: frame %d differs: got %s want %s" % (name, bad[0] + 1, frames[bad[0]], want[bad[0]])
    assert (pairs == GOLD[name + "_pairs"]).all()
    assert idx == int(GOLD[name + "_topo_idx"]) and (np.asarray(nodes, np.float32).view(np.int32) == GOLD[name + "_topo"].view(np.int32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kd", "grid"])
def test_product_free_running_loop_matches_golden_v4(pkg, tmp_path, name):
    assert pkg.device_count() > 0
    n = G4.N_GRID if name == "grid" else LS.N_PARTICLES_FREE
    h = pkg.PfSlam(n, kd_capacity=1 << 18)
    rec = LS.run_free(h, LS.scans(pkg), grid_path=(name == "grid"))
    nodes, idx = h.topology()
    _check(name, rec, nodes, idx)
    assert G4.particle_crc(h.particles()) == GOLD[name + "_particle_crc"].tolist()
    # exported maps: product vs golden (= oracle), cell for cell
    EM.export(h.map() if name == "kd" else np.zeros(0, pkg.NODE_DTYPE), h.grid(), str(tmp_path / "prod"))
    if name == "kd":
        cells, w = GOLD["kd_export_cells"], GOLD["kd_export_w"]
        pts = np.zeros((len(cells), 4), np.float32)
        pts[:, :2] = cells.astype(np.float32) * np.float32(0.025)
        negzero = np.unpackbits(GOLD["kd_export_negzero"])[:2 * len(cells)].reshape(-1, 2).astype(bool)
        pts[:, :2][negzero] = np.float32(-0.0)
        pts[:, 3] = w.astype(np.float32)
        got = EM.kept_points(h.map())
        assert got.shape == pts.shape and (got.view(np.int32) == pts.view(np.int32)).all()
    else:
        assert (h.grid() == GOLD["grid_export"]).all()
    h.close()


@pytest.mark.gpu
def test_topology_booked_one_frame_late_gives_the_same_graph(pkg):
    """pfslam_set_topology(h, 2): the KD frames stay in flight, UpdateTopology / CheckLoopClosure run when a frame is booked.
    Looked at every 20th frame only; poses, map sizes and closure pairs at those frames, the graph at the end and the final particles
    equal the fixture's (= the synchronous order of the reference)."""
    assert pkg.device_count() > 0
    h = pkg.PfSlam(LS.N_PARTICLES_FREE, kd_capacity=1 << 18)
    rec = LS.run_free(h, LS.scans(pkg), look_every=20, topology_mode=2)
    frames, pairs = pack_records(rec)
    want = GOLD["kd_frames"]
    looked = [f - 1 for f in range(1, LS.N_FRAMES + 1) if f % 20 == 0 or f == LS.N_FRAMES]
    assert (frames[:, :5] == want[looked, :5]).all()
    want_pairs = np.concatenate([GOLD["kd_pairs"][want[k, 5]:want[k, 5] + want[k, 6]] for k in looked]) if len(GOLD["kd_pairs"]) else pairs
    assert (frames[:, 6] == want[looked, 6]).all() and (pairs == want_pairs).all()
    nodes, idx = h.topology()
    assert idx == int(GOLD["kd_topo_idx"]) and (np.asarray(nodes, np.float32).view(np.int32) == GOLD["kd_topo"].view(np.int32)).all()
    assert G4.particle_crc(h.particles()) == GOLD["kd_particle_crc"].tolist()
    h.close()
