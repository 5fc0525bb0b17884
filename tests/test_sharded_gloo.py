"""The multi-GPU orchestration under gloo on CPU, world_size 2: the sharded step must reproduce the unsharded
oracle bit for bit (partitioning by global index, packed min/max/argmax keys, best-pose merge, global-array
resample).  The per-rank compute is the oracle-backed engine; the GPU engine has the same stage interface."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_global, n_frames, strict, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_engine as OE
    pkg = importlib.import_module("gpu-icp-slam_amd")
    sharded = importlib.import_module("gpu-icp-slam_amd.sharded")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    stride, goff, n = sharded.shard_layout(n_global, world, rank)
    eng = OE.OracleShardEngine(n, goff, n_global, strict_host_mirror=strict, stride=stride)
    s = sharded.ShardedSlam(pkg, n_global, rank, world, dist=dist, torch=torch, engine=eng, buffers=OE.OracleBuffers(eng))
    segs, frames = pkg.synth.corridor_sequence(n_frames, seed=5)
    log = []
    for f, (pose, scan) in enumerate(frames, start=1):
        s.step(f, scan)
        t = s.trace()
        log.append((t.get("best", -1), t.get("resampled", 0), t.get("kd_size", 0)) + tuple(eng.robot.view(np.int32).tolist()))
    # the fixed schedule: three all-gathers in every frame (pose blocks, records, weights), whether it resamples or not
    n_stepped = sum(1 for row in log if row[0] >= 0)
    assert s.collectives == 3 * n_stepped, (s.collectives, n_stepped)
    # KDTree::Balance at frame 5: ONE host build in the whole job (rank 0), everybody else adopts the broadcast map
    builds = torch.tensor([eng.builds], dtype=torch.int64)
    dist.all_reduce(builds)
    assert int(builds.item()) == 1 and s.balance_broadcasts == 1, (int(builds.item()), s.balance_broadcasts)
    assert eng.builds == (1 if rank == 0 else 0) and eng.adopted == (0 if rank == 0 else 1) and s.balance_builds == eng.builds
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=eng.x, y=eng.y, th=eng.th, w=eng.w, log=np.array(log, np.int64),
             tree=eng.tree[:eng.size])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("strict,world,n_global", [(1, 2, 600), (0, 2, 600), (1, 4, 600), (1, 4, 601), (1, 3, 500)])
def test_ranks_match_unsharded_oracle(tmp_path, pkg, oracle, strict, world, n_global):
    """Also ragged jobs: 601 particles over 4 ranks = 151 + 151 + 151 + 148, 500 over 3 = 167 + 167 + 166."""
    n_frames = 9
    mp.spawn(_worker, args=(world, _free_port(), n_global, n_frames, strict, str(tmp_path)), nprocs=world, join=True)
    o = oracle.Slam(n_global, kd_capacity=1 << 16, strict_host_mirror=strict)
    segs, frames = pkg.synth.corridor_sequence(n_frames, seed=5)
    want_log = []
    for f, (pose, scan) in enumerate(frames, start=1):
        o.step(f, scan)
        t = o.trace()
        want_log.append((t["best"], t["resampled"], t["kd_size"]) + tuple(o.pose.view(np.int32).tolist()))
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(world)]
    for k in range(world):
        assert r[k]["log"].tolist() == [list(map(int, row)) for row in want_log], "rank %d trace differs" % k
        assert r[k]["tree"].tobytes() == o.tree().tobytes()
    want = o.particles()
    for fld, key in (("x", "x"), ("y", "y"), ("theta", "th"), ("w", "w")):
        got = np.concatenate([r[k][key] for k in range(world)])
        assert (got.view(np.int32) == want[fld].view(np.int32)).all(), fld
    assert any(row[1] for row in want_log), "the replay must include a resample"
    o.close()


def _loop_worker(rank, world, port, out_dir, n_frames):
    """BASELINE configs[4] as written: the loop-closure run SHARDED, with the topology graph / loop-closure proposals inside the frame."""
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import loop_scenario as LS
    import oracle_engine as OE
    import oracle_lib as O
    pkg = importlib.import_module("gpu-icp-slam_amd")
    sharded = importlib.import_module("gpu-icp-slam_amd.sharded")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    n_global = LS.N_PARTICLES
    stride, goff, n = sharded.shard_layout(n_global, world, rank)
    eng = OE.OracleShardEngine(n, goff, n_global, kd_capacity=1 << 18, stride=stride)
    s = sharded.ShardedSlam(pkg, n_global, rank, world, dist=dist, torch=torch, engine=eng, buffers=OE.OracleBuffers(eng))
    mk = lambda p: O.make_particles(n_global, float(p[0]), float(p[1]), float(p[2]))
    rec = LS.run(s, mk, LS.scans(pkg, n_frames=n_frames), n_frames=n_frames)
    nodes, idx = s.topology()
    from make_golden_v3 import pack_records
    frames, pairs = pack_records(rec)
    np.savez(os.path.join(out_dir, "loop%d.npz" % rank), frames=frames, pairs=pairs, nodes=np.asarray(nodes, np.float32), idx=idx, tree=s.map())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_loop_closure_run_matches_golden_v3(tmp_path, pkg, world):
    """configs[4]: the 260-frame closed loop with UpdateTopology / CheckLoopClosure in the frame (kernel.cu:1750-1751), particles sharded
    over 2 and 4 ranks: every rank reproduces the committed fixture of the UNSHARDED oracle -- per-frame pose bits, map size, resample
    flags, loop-closure proposals, the topology graph -- and all ranks end with the same map."""
    import loop_scenario as LS
    gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_v3.npz"))
    n_frames = LS.N_FRAMES
    mp.spawn(_loop_worker, args=(world, _free_port(), str(tmp_path), n_frames), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "loop%d.npz" % k)) for k in range(world)]
    for k in range(world):
        bad = np.flatnonzero((r[k]["frames"] != gold["kd_frames"]).any(1))
        assert len(bad) == 0, "rank %d: frame %d differs: got %s want %s" % (k, bad[0] + 1, r[k]["frames"][bad[0]], gold["kd_frames"][bad[0]])
        assert (r[k]["pairs"] == gold["kd_pairs"]).all()
        assert int(r[k]["idx"]) == int(gold["kd_topo_idx"]) and (r[k]["nodes"].view(np.int32) == gold["kd_topo"].view(np.int32)).all()
        assert r[k]["tree"].tobytes() == r[0]["tree"].tobytes()
    assert (gold["kd_frames"][:, 6] > 0).sum() > 50   # the run does propose closures


def test_key_packing_orders_like_minmax_element():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_engine as OE
    rng = np.random.RandomState(0)
    fit = np.concatenate([rng.randint(-5, 5, 200).astype(np.float32), [-0.5, 0.5, -113 * 1081, 113 * 1081]]).astype(np.float32)
    rng.shuffle(fit)
    gi = np.arange(len(fit)).astype(np.int64)
    kmax = np.max((OE.f32_to_ordered(fit) << 32) | (0xFFFFFFFF - gi))
    best = int(0xFFFFFFFF - (int(kmax) & 0xFFFFFFFF))
    assert best == int(np.argmax(fit)) and OE.ordered_to_f32(kmax >> 32) == fit.max()
    kmin = np.max(OE.f32_to_ordered(-fit) << 32)
    assert -OE.ordered_to_f32(kmin >> 32) == fit.min()
