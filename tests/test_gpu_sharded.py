"""Sharded handles on ONE GPU: 'virtual ranks' are stepped through the sharded frame with
the collectives done by hand on the zero-copy torch views; the result must equal the unsharded oracle and the
unsharded GPU step bit for bit.  (The real multi-process path is covered under gloo in test_sharded_gloo.py.)"""
import importlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


@pytest.mark.parametrize("n_global,world", [(1000, 2), (1001, 3)])
def test_virtual_ranks_on_one_gpu_match_oracle(pkg, n_global, world):
    """The sharded frame of include/pfslam.h (fixed schedule: pose blocks, records, weights) with the all-gathers done by hand
    between handles that share one GPU: equal shards (2 x 500) and a ragged job (1001 = 334 + 334 + 333, exchange buffers padded
    to the stride)."""
    torch = pytest.importorskip("torch")
    assert pkg.device_count() > 0 and torch.cuda.is_available()
    sharded = importlib.import_module("gpu-icp-slam_amd.sharded")
    n_frames = 12
    lay = [sharded.shard_layout(n_global, world, r) for r in range(world)]
    engs = [pkg.PfSlam(cnt, kd_capacity=1 << 16, global_offset=off, global_n=n_global, shard_stride=stride) for stride, off, cnt in lay]
    bufs = [sharded.GpuBuffers(e, torch, 0) for e in engs]
    o = O.Slam(n_global, kd_capacity=1 << 16)
    segs, frames = pkg.synth.corridor_sequence(n_frames, seed=5)
    n_resampled = 0

    def sync():
        for e in engs:
            e.synchronize()
        torch.cuda.synchronize()

    for e in engs:
        e.set_shard_balance(True)   # ONE KDTree::Balance per job (include/pfslam.h, pfslam_shard_balance_*): rank 0 builds ...
    n_adopted = 0
    for f, (pose, scan) in enumerate(frames, start=1):
        o.step(f, scan)
        due = [e.shard_balance_due(f) for e in engs]
        assert len(set(due)) == 1
        if due[0][0]:
            engs[0].shard_balance_build(f)
            sync()
            src = bufs[0].tree_buffers(due[0][1])
            for b in bufs[1:]:                                                  # ... the others get its device arrays ("broadcast")
                for dst, s_ in zip(b.tree_buffers(due[0][1]), src):
                    dst.copy_(s_)
            sync()
            for e in engs[1:]:
                e.shard_balance_adopt()
                n_adopted += 1
        seeded = [e.shard_disperse(f, scan) for e in engs]
        assert len(set(seeded)) == 1
        if seeded[0]:
            continue
        sync()
        blocks = [b.pose_blocks() for b in bufs]                                # the local block alternates: ask every frame
        g = torch.cat([loc for loc, _ in blocks])                               # all-gather of the [x | y | theta] blocks
        for _, glob in blocks:
            glob.copy_(g)
        sync()
        for e in engs:
            e.shard_score()
        sync()
        packs = torch.cat([b.pack for b in bufs])                               # all-gather of the 16-byte records (packed keys)
        for b in bufs:
            b.packs.copy_(packs)
        sync()
        for e in engs:
            e.shard_weights()
        sync()
        gw = torch.cat([b.w for b in bufs])                                     # all-gather of the (padded) weights
        for b in bufs:
            b.gw.copy_(gw)
        sync()
        for e in engs:
            e.shard_finish()
        t = o.trace()
        n_resampled += t["resampled"]
        for e in engs:                                                          # trace() books the frame in flight
            te = e.trace()
            assert te["best"] == t["best"] and te["resampled"] == t["resampled"]
            assert (bits(e.pose) == bits(o.pose)).all()
            assert e.kd_size == o.kd_size
    assert n_resampled > 0 and n_adopted == world - 1      # frame 5 re-balances (12 frames)
    want = o.particles()
    got = [e.particles() for e in engs]
    for fld in ("x", "y", "theta", "w"):
        assert (bits(np.concatenate([g[fld] for g in got])) == bits(want[fld])).all(), fld
    for e in engs:
        assert e.map().tobytes() == o.tree().tobytes()
        e.close()
    o.close()


def test_sharded_wrapper_world1_equals_plain_step(pkg):
    torch = pytest.importorskip("torch")
    sharded = importlib.import_module("gpu-icp-slam_amd.sharded")
    n, n_frames = 500, 10
    a = pkg.PfSlam(n, kd_capacity=1 << 16)
    s = sharded.ShardedSlam(pkg, n, 0, 1, device=0, torch=torch, kd_capacity=1 << 16)
    segs, frames = pkg.synth.corridor_sequence(n_frames, seed=7)
    for f, (pose, scan) in enumerate(frames, start=1):
        a.step(f, scan)
        s.step(f, scan)
        assert (bits(a.pose) == bits(s.pose)).all()
    s.synchronize()
    pa, ps = a.particles(), s.eng.particles()
    for fld in ("x", "y", "theta", "w"):
        assert (bits(pa[fld]) == bits(ps[fld])).all()
    assert a.map().tobytes() == s.eng.map().tobytes()
    a.close(); s.eng.close()


def _mp_worker(rank, world, port, n_global, n_frames, out_dir):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("gpu-icp-slam_amd")
    sharded = importlib.import_module("gpu-icp-slam_amd.sharded")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    s = sharded.ShardedSlam(pkg, n_global, rank, world, device=0, dist=dist, torch=torch, kd_capacity=1 << 16)
    segs, frames = pkg.synth.corridor_sequence(n_frames, seed=5)
    poses = []
    for f, (pose, scan) in enumerate(frames, start=1):
        s.step(f, scan)
        poses.append(s.pose.view(np.int32).tolist())
    s.synchronize()
    p = s.eng.particles()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=p["x"], y=p["y"], th=p["theta"], w=p["w"], poses=np.array(poses),
             tree=s.eng.map().view(np.uint8))
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_one_gpu_gloo_match_single_handle(tmp_path, pkg):
    """The real multi-process path (ShardedSlam over torch.distributed) with both ranks on GPU 0 and gloo carrying the
    collectives: bit-identical to the single-handle step.  (RCCL itself needs one GPU per rank.)"""
    import os
    import socket
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    n_global, n_frames, world = 800, 10, 2
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    mp.spawn(_mp_worker, args=(world, port, n_global, n_frames, str(tmp_path)), nprocs=world, join=True)
    a = pkg.PfSlam(n_global, kd_capacity=1 << 16)
    segs, frames = pkg.synth.corridor_sequence(n_frames, seed=5)
    poses = []
    for f, (pose, scan) in enumerate(frames, start=1):
        a.step(f, scan)
        poses.append(a.pose.view(np.int32).tolist())
    want = a.particles()
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(world)]
    for k in range(world):
        assert r[k]["poses"].tolist() == poses
        assert r[k]["tree"].tobytes() == a.map().tobytes()
    for fld, key in (("x", "x"), ("y", "y"), ("theta", "th"), ("w", "w")):
        assert (bits(np.concatenate([r[k][key] for k in range(world)])) == bits(want[fld])).all(), fld
    a.close()


class _VirtualRanks:
    """`world` sharded handles on one GPU stepped as ONE engine: the three all-gathers of the frame done by hand on the zero-copy
    views (the protocol of ShardedSlam.step), every rank's pose / trace / closures / graph checked against rank 0's."""

    def __init__(self, pkg, torch, n_global, world, **kw):
        sharded = importlib.import_module("gpu-icp-slam_amd.sharded")
        self.torch, self.n_global = torch, n_global
        self.lay = [sharded.shard_layout(n_global, world, r) for r in range(world)]
        self.engs = [pkg.PfSlam(cnt, global_offset=off, global_n=n_global, shard_stride=stride, **kw) for stride, off, cnt in self.lay]
        self.bufs = [sharded.GpuBuffers(e, torch, 0) for e in self.engs]

    def _sync(self):
        for e in self.engs:
            e.synchronize()
        self.torch.cuda.synchronize()

    def set_topology(self, mode):
        for e in self.engs:
            e.set_topology(mode)

    def shift_particles(self, d):
        for e in self.engs:
            e.shift_particles(d)

    def step(self, f, scan):
        torch, engs, bufs = self.torch, self.engs, self.bufs
        seeded = [e.shard_disperse(f, scan) for e in engs]
        assert len(set(seeded)) == 1
        if seeded[0]:
            return
        self._sync()
        blocks = [b.pose_blocks() for b in bufs]
        g = torch.cat([loc for loc, _ in blocks])
        for _, glob in blocks:
            glob.copy_(g)
        self._sync()
        for e in engs:
            e.shard_score()
        self._sync()
        packs = torch.cat([b.pack for b in bufs])
        for b in bufs:
            b.packs.copy_(packs)
        self._sync()
        for e in engs:
            e.shard_weights()
        self._sync()
        gw = torch.cat([b.w for b in bufs])
        for b in bufs:
            b.gw.copy_(gw)
        self._sync()
        for e in engs:
            e.shard_finish()

    def _same(self, fn):
        vals = [fn(e) for e in self.engs]
        for v in vals[1:]:
            assert repr(v) == repr(vals[0])
        return vals[0]

    def trace(self): return self._same(lambda e: e.trace())

    @property
    def pose(self):
        self._same(lambda e: np.asarray(e.pose, np.float32).view(np.int32).tolist())
        return self.engs[0].pose

    def closures(self):
        self._same(lambda e: e.closures().tolist())
        return self.engs[0].closures()
    def topology(self): return self.engs[0].topology()
    def close(self):
        for e in self.engs:
            e.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_sharded_frames_with_topology_match_the_single_handle(pkg, mode):
    """BASELINE configs[4] on the product: the free-running closed loop (tests/loop_scenario.py: odometry shifts, topology graph and
    loop-closure proposals inside the frame) on three virtual ranks of one GPU against ONE handle stepping the same 20 000 particles:
    every frame's pose, map size, resample flag and proposals, and the final graph; every rank keeps the same graph."""
    torch = pytest.importorskip("torch")
    import loop_scenario as LS
    n, world, n_frames = 20000, 3, LS.N_FRAMES
    scans = LS.scans(pkg, n_frames=n_frames)
    one = pkg.PfSlam(n, kd_capacity=1 << 18)
    want = LS.run_free(one, scans, n_frames=n_frames, look_every=1 if mode == 1 else 4, topology_mode=mode)
    want_graph = one.topology()
    v = _VirtualRanks(pkg, torch, n, world, kd_capacity=1 << 18)
    got = LS.run_free(v, scans, n_frames=n_frames, look_every=1 if mode == 1 else 4, topology_mode=mode)
    assert got == want
    for e in v.engs:
        nodes, idx = e.topology()
        assert idx == want_graph[1] and (np.asarray(nodes, np.float32).view(np.int32) == np.asarray(want_graph[0], np.float32).view(np.int32)).all()
    assert sum(len(r[3]) for r in want) > 100   # the loop closes: proposals were made
    one.close(); v.close()


def _bench_like(pkg, n, n_map=20000, n_frames=30, seed=1):
    pts, segs = pkg.synth.make_map_points(n_map, seed=seed)
    tree = pkg.kd_create(pts)
    scans = [pkg.synth.make_scan(segs, (0.002 * f, 0.001 * f, 0.0004 * f), seed=2000 + f) for f in range(n_frames)]
    return tree, scans


def _run(e, tree, scans, step, first=6):
    e.set_map(tree)
    for f in range(1, 6):
        e.motion_update(f)
    poses = []
    for i, s in enumerate(scans):
        step(first + i, s)
        poses.append(bits(e.pose).tolist())
    e.synchronize()
    return poses, e.particles().copy(), e.map().tobytes(), e.trace()


def test_sharded_frame_is_the_round5_frame(pkg):
    """The sharded calls enqueue the four parts of the very frame pfslam_step enqueues (pfslam_frame.hip.inc): a one-rank job stepped
    through pfslam_shard_* runs round-5 frames (pfslam_frame_mode) and gives pose, particles, map and trace of pfslam_step bit for bit --
    at a particle count whose scan-match pass is organised by lattice-cell rows."""
    n = 20000
    tree, scans = _bench_like(pkg, n)
    a = pkg.PfSlam(n, kd_capacity=len(tree) + (1 << 18))
    want = _run(a, tree, scans, a.step)
    assert a.frame_mode()["round5_frame"]
    a.close()
    b = pkg.PfSlam(n, kd_capacity=len(tree) + (1 << 18))

    def shard_step(f, s):
        if b.shard_disperse(f, s):
            return
        b.shard_score(); b.shard_weights(); b.shard_finish()
    got = _run(b, tree, scans, shard_step)
    assert b.frame_mode()["round5_frame"], "the sharded frame fell back to the staged chain"
    assert got[0] == want[0]
    for fld in ("x", "y", "theta", "w"):
        assert (bits(got[1][fld]) == bits(want[1][fld])).all(), fld
    assert got[2] == want[2]
    assert {k: got[3][k] for k in ("best", "resampled", "kd_size")} == {k: want[3][k] for k in ("best", "resampled", "kd_size")}
    b.close()


def test_native_rccl_rank_world1_equals_plain_step(pkg):
    """libpfslam_mgpu.so (include/pfslam_mgpu.h) steps a one-rank job: the same frames as pfslam_step, bit for bit; its barrier and
    statistics work without a communicator.  (More than one rank needs one GPU per rank: RCCL refuses two ranks on one device.)"""
    n = 20000
    tree, scans = _bench_like(pkg, n, n_frames=12)
    a = pkg.PfSlam(n, kd_capacity=len(tree) + (1 << 18))
    want = _run(a, tree, scans, a.step)
    a.close()
    b = pkg.PfSlam(n, kd_capacity=len(tree) + (1 << 18))
    m = pkg.MgpuRank(b, 1, 0)
    got = _run(b, tree, scans, m.step)
    assert got[0] == want[0] and got[2] == want[2]
    for fld in ("x", "y", "theta", "w"):
        assert (bits(got[1][fld]) == bits(want[1][fld])).all(), fld
    assert m.barrier_max(3.5) == 3.5
    st = m.stats()
    assert st["world"] == 1 and st["collectives"] == 3 * len(scans)
    assert m.time_collectives(2) == {"pose_blocks": 0.0, "records": 0.0, "weights": 0.0}
    m.close(); b.close()


@pytest.mark.parametrize("world", [2, 3])
def test_virtual_ranks_run_round5_frames_at_cell_row_sizes(pkg, world):
    """Virtual ranks at a size where every shard's scan-match pass uses lattice-cell rows: the sharded frames are round-5 frames (not the
    staged fallback), the job-wide |heading| maximum reaches every rank's header, and the job equals ONE handle stepping all particles."""
    torch = pytest.importorskip("torch")
    n = 30001 if world == 3 else 30000
    tree, scans = _bench_like(pkg, n, n_frames=16)
    one = pkg.PfSlam(n, kd_capacity=len(tree) + (1 << 18))
    want = _run(one, tree, scans, one.step)
    one.close()
    v = _VirtualRanks(pkg, torch, n, world, kd_capacity=len(tree) + (1 << 18))
    for e in v.engs:
        e.set_map(tree)
        for f in range(1, 6):
            e.motion_update(f)
    poses, v2_frames = [], 0
    for i, s in enumerate(scans):
        v.step(6 + i, s)
        poses.append(bits(v.pose).tolist())
        fms = [e.frame_mode() for e in v.engs]
        v2_frames += all(fm["round5_frame"] and not fm["gates"] for fm in fms)   # (several handles in one process: the edges are events)
    # (a shard whose cloud has grown too wide for its size goes back to the staged chain for a frame -- the organisation follows the cloud's
    # spread, per rank -- and the job still equals the single handle; here nearly every frame is a round-5 frame on every rank)
    assert v2_frames >= len(scans) - 3
    for e in v.engs:
        e.synchronize()
    assert poses == want[0]
    got = [e.particles() for e in v.engs]
    for fld in ("x", "y", "theta", "w"):
        assert (bits(np.concatenate([g[fld] for g in got])) == bits(want[1][fld])).all(), fld
    for e in v.engs:
        assert e.map().tobytes() == want[2]
    v.close()


def test_two_handles_interleaved_do_not_wait_on_each_other(pkg):
    """Two handles of one process with frames in flight at the same time: their eight streams may share hardware queues, so neither uses
    stream gates (a gate spins inside a kernel); both run to the results of a handle stepped alone, and a handle that is alone again
    goes back to gates."""
    import time
    n = 20000
    tree, scans = _bench_like(pkg, n, n_frames=14)
    solo = pkg.PfSlam(n, kd_capacity=len(tree) + (1 << 18))
    want = _run(solo, tree, scans, solo.step)
    gates_alone = solo.frame_mode()["gates"]
    solo.close()
    a = pkg.PfSlam(n, kd_capacity=len(tree) + (1 << 18))
    b = pkg.PfSlam(n, kd_capacity=len(tree) + (1 << 18))
    for e in (a, b):
        e.set_map(tree)
        for f in range(1, 6):
            e.motion_update(f)
    t0 = time.time()
    for i, s in enumerate(scans):
        a.step(6 + i, s)
        b.step(6 + i, s)
    assert not a.frame_mode()["gates"] and not b.frame_mode()["gates"]
    a.synchronize(); b.synchronize()
    assert time.time() - t0 < 20.0          # (a gate that gives up takes a second per frame)
    for e in (a, b):
        assert bits(e.pose).tolist() == want[0][-1]
        assert e.map().tobytes() == want[2]
    b.close()
    a.step(6 + len(scans), scans[0])        # alone again
    a.synchronize()
    assert a.frame_mode()["gates"] == gates_alone
    a.close()
