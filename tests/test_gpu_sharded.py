"""Sharded handles on ONE GPU: two 'virtual ranks' (global offsets 0 and n/2) are stepped stage by stage with
the collectives done by hand on the zero-copy torch views; the result must equal the unsharded oracle and the
unsharded GPU step bit for bit.  (The real multi-process path is covered under gloo in test_sharded_gloo.py.)"""
import importlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def test_two_virtual_ranks_on_one_gpu_match_oracle(pkg):
    torch = pytest.importorskip("torch")
    assert pkg.device_count() > 0 and torch.cuda.is_available()
    sharded = importlib.import_module("gpu-icp-slam_amd.sharded")
    n_global, world, n_frames = 1000, 2, 12
    n = n_global // world
    engs = [pkg.PfSlam(n, kd_capacity=1 << 16, global_offset=r * n, global_n=n_global) for r in range(world)]
    bufs = [sharded.GpuBuffers(e, torch, 0) for e in engs]
    o = O.Slam(n_global, kd_capacity=1 << 16)
    segs, frames = pkg.synth.corridor_sequence(n_frames, seed=5)
    n_resampled = 0
    for f, (pose, scan) in enumerate(frames, start=1):
        o.step(f, scan)
        for e in engs:
            e.set_scan(scan); e.maybe_balance(f)
        if engs[0].kd_size == 0:
            for e in engs:
                e.set_pose(np.zeros(3, np.float32)); e.update_map_kd()
            continue
        for e in engs:
            e.motion_update(f); e.score_kd(fetch=False); e.measurement_local(); e.synchronize()
        merged = torch.maximum(bufs[0].stats[:2], bufs[1].stats[:2])          # all-reduce MAX
        for b in bufs:
            b.stats[:2].copy_(merged)
        torch.cuda.synchronize()
        res = [e.measurement_apply() for e in engs]
        assert res[0] == res[1]
        start = bufs[0].start + bufs[1].start                                   # all-reduce SUM
        for b in bufs:
            b.start.copy_(start)
        torch.cuda.synchronize()
        for e in engs:
            e.icp(None); e.synchronize()
        gw = torch.cat([bufs[0].w, bufs[1].w])                                   # all-gather
        for b in bufs:
            b.gw.copy_(gw)
        torch.cuda.synchronize()
        for e in engs:
            e.update_map_kd()
        plans = [e.resample_plan(f) for e in engs]
        assert plans[0] == plans[1]
        if plans[0][0]:
            n_resampled += 1
            views = [b.pose_views() for b in bufs]
            for k in range(3):
                g = torch.cat([views[0][0][k], views[1][0][k]])
                for v in views:
                    v[1][k].copy_(g)
            torch.cuda.synchronize()
            for e in engs:
                e.resample_gather()
        t = o.trace()
        assert res[0][0] == t["best"] and plans[0][0] == t["resampled"]
        for e in engs:
            assert (bits(e.pose) == bits(o.pose)).all()
            assert e.kd_size == o.kd_size
    assert n_resampled > 0
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
