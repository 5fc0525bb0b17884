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


def test_frame_chain_is_short(pkg, world):
    """The launches between two scan-match kernels, read off the frame's own kernels (pfslam_set_probe): reduce -> walls + insert -> cell rows
    -> next scan-match within 160 us at 100 000 particles (round 4: ~190), and every launch of the chain present in every frame."""
    tree, scans = world
    h = pkg.PfSlam(100000, kd_capacity=len(tree) + (1 << 18))
    h.set_map(tree)
    for f in range(1, 6):
        h.motion_update(f)
    h.set_probe(64)
    for i, s in enumerate(scans[:25]):
        h.step(6 + i, s)
    h.synchronize()
    names, t, last = h.probe(25)
    sc, rd, up = names.index("C scan-match"), names.index("C reduce"), names.index("C cells update")
    t = t[-15:]
    assert (t[:, [sc, rd, up]] > 0).all()
    chain = t[1:, sc] - t[:-1, rd]
    assert np.median(chain) < 160.0, chain
    assert h.check_cells()["violations"] == 0
    h.close()


def test_long_differential_fuzz_with_cell_rows_at_every_count():
    """Four minutes of tests/fuzz_step.py with the lattice-cell rows (hence the round-5 frame) forced on at every particle count, one
    more with every launch on one stream, one with the edges as events: no divergence from the oracle."""
    for secs, seed, extra in ((240, "105", {}), (45, "106", {"PFSLAM_SERIAL": "1"}), (45, "107", {"PFSLAM_GATES": "0"})):
        env = dict(os.environ, PFSLAM_PLAN_MIN_N="1", PFSLAM_VARIANT="3", **extra)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_step.py"), str(secs), seed], capture_output=True, text=True,
                             timeout=secs + 240, env=env)
        assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
