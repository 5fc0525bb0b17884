"""Round-5 frame loop (csrc/pfslam_frame.hip.inc): four in-order chains of launches on four streams, ordered by a fixed set of edges.

What an oracle comparison cannot see -- rows that are right today and that nobody watches, records published from list slots that
were counted but not yet written -- is checked here in two other ways:
  * the same frames stepped with every launch on ONE stream (pfslam_set_serial) give the same results AND the same bookkeeping
    (claimed cells, records, rows, pool slots, extensions, publishing passes), to the last slot;
  * pfslam_debug_check_cells: every published record owns its table words, every row lists candidates of its record in order, no
    watched link has gained a node without the cell being extended, the table holds what the counters say.
Both round-4 races, re-introduced by a switch (PFSLAM_FAULT), are caught by these checks."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOOK = ("cells", "rows", "pool_slots", "walked_from_root", "extended", "reused", "claimed", "updates", "wipes", "flags", "cells_without_row")
CHECK = ("records", "unwalked", "fresh", "published", "row_words", "pending_words", "fallback_words", "dead")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def run_frames(pkg, tree, scans, n, serial, variant=0, look_every=0, first_frame=6, env=None):
    """Step `scans` through a fresh handle; returns (per-frame rows of trace + pose bits, particles, map bytes, bookkeeping, check)."""
    old = {}
    env = dict(env or {}, PFSLAM_STABLE_ORDER="1")  # (a canonical lane order: the counting sort's is arrival order inside a Hilbert cell)
    for k, v in env.items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        h = pkg.PfSlam(n, kd_capacity=len(tree) + (1 << 18))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    h.set_map(tree)
    if variant:
        h.set_variant(variant)
    if serial:
        h.set_serial(1)
    for f in range(1, 6):
        h.motion_update(f)
    rows = []
    for i, s in enumerate(scans):
        h.step(first_frame + i, s)
        if look_every and (i + 1) % look_every == 0:
            t = h.trace()
            rows.append([t["best"], t["resampled"], t["n_wall"], t["n_free"], t["n_insert"], t["kd_size"]] + bits(h.pose).tolist())
    h.synchronize()
    t = h.trace()
    rows.append([t["best"], t["resampled"], t["n_wall"], t["n_free"], t["n_insert"], t["kd_size"]] + bits(h.pose).tolist())
    st = h.cell_stats()
    chk = h.check_cells()
    p = h.particles().copy()
    m = h.map().tobytes()
    h.close()
    return rows, p, m, {k: st[k] for k in BOOK}, chk


@pytest.fixture(scope="module")
def world(pkg):
    pts, segs = pkg.synth.make_map_points(100000, seed=1)
    tree = pkg.kd_create(pts)
    scans = [pkg.synth.make_scan(segs, (0.002 * i, 0.001 * i, 0.0004 * i), seed=2000 + i) for i in range(30)]
    return tree, scans


@pytest.mark.parametrize("n,variant,frames,look", [(100000, 0, 24, 0), (100000, 0, 12, 3), (20000, 0, 30, 1), (3000, 3, 30, 0), (300, 3, 20, 2)])
def test_serial_and_concurrent_frames_agree_in_results_and_bookkeeping(pkg, world, n, variant, frames, look):
    """Four streams or one: same poses, traces, particles, map -- and the same cell-row bookkeeping, with the invariants intact."""
    tree, scans = world
    a = run_frames(pkg, tree, scans[:frames], n, serial=False, variant=variant, look_every=look)
    b = run_frames(pkg, tree, scans[:frames], n, serial=True, variant=variant, look_every=look)
    assert a[0] == b[0]
    for fld in ("x", "y", "theta", "w"):
        assert (bits(a[1][fld]) == bits(b[1][fld])).all(), fld
    assert a[2] == b[2]
    assert a[3] == b[3], (a[3], b[3])
    assert a[3]["cells"] > 1000 and a[3]["flags"] == 0
    for chk in (a[4], b[4]):
        assert chk["violations"] == 0, chk
    assert {k: a[4][k] for k in CHECK} == {k: b[4][k] for k in CHECK}, (a[4], b[4])


def test_events_instead_of_gates_give_the_same_frames(pkg, world):
    """PFSLAM_GATES=0: the frame's cross-stream edges as events (what a handle falls back to when its streams share a hardware queue)."""
    tree, scans = world
    a = run_frames(pkg, tree, scans[:12], 50000, serial=False)
    b = run_frames(pkg, tree, scans[:12], 50000, serial=False, env={"PFSLAM_GATES": "0"})
    c = run_frames(pkg, tree, scans[:12], 50000, serial=False, env={"PFSLAM_MARK_EARLY": "0", "PFSLAM_PUBLISH_LAG": "3"})
    assert a[0] == b[0] == c[0] and a[2] == b[2] == c[2]
    assert a[3] == b[3], (a[3], b[3])
    assert b[4]["violations"] == 0 and c[4]["violations"] == 0, (b[4], c[4])


def test_injected_faults_are_caught(pkg, world):
    """The two round-4 races, switched back on (PFSLAM_FAULT, see CellPass): neither changes a score on these frames -- the rows are
    valid when they are published --, both are caught: the bookkeeping differs from the one-stream run's, or a published record has a
    watched link that has gained a node."""
    tree, scans = world
    good = run_frames(pkg, tree, scans[:16], 100000, serial=True)
    assert good[4]["violations"] == 0
    # fault 2: a walked record's links are not validated when it is published (rows made on an older tree)
    f2 = run_frames(pkg, tree, scans[:16], 100000, serial=False, env={"PFSLAM_FAULT": "2"})
    assert f2[4]["v_watched_link_has_child"] > 0, f2[4]
    # fault 1: a publishing pass takes every record up to the list counter (not only those a finished walk pass has written)
    caught = False
    for rep in range(3):
        f1 = run_frames(pkg, tree, scans[:16], 100000, serial=False, env={"PFSLAM_FAULT": "1"})
        caught = caught or f1[3] != good[3] or f1[4]["violations"] > 0 or {k: f1[4][k] for k in CHECK} != {k: good[4][k] for k in CHECK}
    assert caught


def test_frame_chain_is_short():
    """The launches between two scan-match kernels, read off the frame's own kernels (pfslam_set_probe, tools/frame_probe.py): reduce -> walls +
    insert -> cell rows -> next scan-match within 160 us at 100 000 particles (round 4: 180-227).  In a process of its own: timing in a
    process that has created and destroyed dozens of handles and torch streams says little (343 us there, once)."""
    import re
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "frame_probe.py"), "--frames", "20"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"chain .*?mean ([0-9.]+) us", out.stdout)
    assert m and "frames with stamps: 20 of 20" in out.stdout, out.stdout[-2000:]
    gates = "'gates': True" in out.stdout
    assert float(m.group(1)) < (160.0 if gates else 200.0), out.stdout[-2500:]
    assert "'violations': 0" in out.stdout


def test_long_differential_fuzz_with_cell_rows_at_every_count():
    """Four minutes of tests/fuzz_step.py with the lattice-cell rows (hence the round-5 frame) forced on at every particle count, one
    more with every launch on one stream, one with the edges as events: no divergence from the oracle."""
    for secs, seed, extra in ((240, "105", {}), (45, "106", {"PFSLAM_SERIAL": "1"}), (45, "107", {"PFSLAM_GATES": "0"})):
        env = dict(os.environ, PFSLAM_PLAN_MIN_N="1", PFSLAM_VARIANT="3", **extra)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_step.py"), str(secs), seed], capture_output=True, text=True,
                             timeout=secs + 240, env=env)
        assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_a_cloud_that_widens_behind_the_gates_back_costs_a_bounded_frame(pkg, world):
    """The cell rows' marking pass costs the AREA of the waves' beam-end boxes, and the choice of the organisation is the host's, from
    an estimate of the cloud's spread: pfslam_set_particles takes it from the first 1024 particles.  A cloud whose first 1024 particles
    are tight and whose other 99 000 are spread over metres (a kidnapped-robot re-seed between two frames) is therefore scored with the
    cell rows once -- round 4: 61 ms for that pass; now a group whose box is wider than PF_CELL_MARK_MAX cells is simply not marked, its
    lanes take the generic traversal, and the frame costs what the plain traversal does.  The header of that frame carries the real
    spread, and the frames behind it are organised by it.  Results stay the oracle's throughout (checked on the pose against a handle
    that scores with the plain traversal)."""
    import time
    tree, scans = world
    n = 100000
    h = pkg.PfSlam(n, kd_capacity=len(tree) + (1 << 18))
    ref = pkg.PfSlam(n, kd_capacity=len(tree) + (1 << 18))
    ref.set_variant(2)                       # plain traversal: no rows, no marking
    for e in (h, ref):
        e.set_map(tree)
        for f in range(1, 6):
            e.motion_update(f)
    for i in range(8):
        h.step(6 + i, scans[i]); ref.step(6 + i, scans[i])
    p = h.particles().copy()
    rng = np.random.RandomState(3)
    mx, my, mt = [float(np.mean(p[k])) for k in ("x", "y", "theta")]
    p["x"] = (mx + rng.normal(0, 1.0, n)).astype(np.float32); p["y"] = (my + rng.normal(0, 1.0, n)).astype(np.float32)
    p["theta"] = (mt + rng.normal(0, 0.12, n)).astype(np.float32)
    p["x"][:1024] = np.float32(mx); p["y"][:1024] = np.float32(my); p["theta"][:1024] = np.float32(mt)   # what the host's estimate sees
    for e in (h, ref):
        e.set_particles(p)
    ref.synchronize(); h.synchronize()
    times = []
    for i in range(8, 14):
        t0 = time.perf_counter()
        h.step(6 + i, scans[i]); h.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        ref.step(6 + i, scans[i]); ref.synchronize()
        assert (bits(h.pose) == bits(ref.pose)).all(), i
    t0 = time.perf_counter()
    for i in range(14, 20):
        ref.step(6 + i, scans[i])
    ref.synchronize()
    plain_ms = (time.perf_counter() - t0) / 6 * 1e3
    assert max(times) < 3.0 * max(plain_ms, 2.0), (times, plain_ms)
    h.close(); ref.close()
