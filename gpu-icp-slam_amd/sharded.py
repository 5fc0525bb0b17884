"""Multi-GPU stepping: particles shard over the ranks of one node (one process per GPU), the map, the scan,
the ICP solve and the map update are replicated, and the merges are a handful of small collectives
(torch.distributed: backend "nccl" is RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Per frame:
    all-reduce MAX  of 2 x int64   packed (fit, -index) keys -> global min / max / first-occurrence argmax
    all-reduce SUM  of 4 x f32     best particle's pose (zero on the ranks that do not own it)
    all-gather      of n x f32     weights -> global array (Neff, cdf and sampling run on it, replicated)
    all-gather      of 3 n x f32   poses, only in frames that resample
Everything that touches randomness is keyed by GLOBAL particle indices, and every sum runs on the global
array in the canonical order, so the result is bit-identical for any number of ranks.

`engine` is anything with the PfSlam stage interface (the GPU handle in production; the CPU tests plug the
oracle in to exercise this orchestration under gloo without a GPU)."""
import numpy as np


class _DevView:
    """Zero-copy torch view of a device buffer of the C-ABI handle (through __cuda_array_interface__)."""

    def __init__(self, ptr, nbytes, typestr, itemsize):
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr,
                                         "data": (ptr, False), "version": 2}


class GpuBuffers:
    """torch tensors aliasing the handle's device buffers (ids of pfslam_device_ptr)."""

    def __init__(self, eng, torch, device):
        self.torch = torch

        cache = {}

        def view(which, typestr, itemsize, dtype):
            # wrapping a pointer costs ~20 us; the pose buffers only alternate between two addresses, so keep the wrappers
            ptr, nbytes = eng.device_ptr(which)
            key = (ptr, nbytes, typestr)
            t = cache.get(key)
            if t is None:
                t = cache[key] = torch.as_tensor(_DevView(ptr, nbytes, typestr, itemsize), device=torch.device("cuda", device)).view(dtype)
            return t

        self.stats = view(0, "<i8", 8, torch.int64)
        self.start = view(8, "<f4", 4, torch.float32)
        self.w = view(5, "<f4", 4, torch.float32)
        self.gw = view(10, "<f4", 4, torch.float32)
        self.eng, self._view = eng, view

    def pose_views(self):
        # x/y/theta swap buffers on every resample, so re-query the pointers
        t = self.torch
        return ([self._view(k, "<f4", 4, t.float32) for k in (2, 3, 4)],
                [self._view(k, "<f4", 4, t.float32) for k in (11, 12, 13)])


class ShardedSlam:
    def __init__(self, pkg, n_global, rank, world, device=0, dist=None, torch=None, engine=None, buffers=None, **kw):
        if n_global % world:
            raise ValueError("n_global must be a multiple of the world size (equal shards for all-gather)")
        self.n_global, self.rank, self.world = n_global, rank, world
        self.n = n_global // world
        self.dist, self.torch = dist, torch
        if engine is None:
            engine = pkg.PfSlam(self.n, device=device, global_offset=rank * self.n, global_n=n_global, **kw)
            if torch is not None:
                # run the kernels on torch's current stream so RCCL collectives and kernels are stream-ordered; a stream of
                # our own rather than the legacy null stream, which synchronises implicitly with every blocking stream
                if torch.cuda.current_stream().cuda_stream == 0:
                    self._stream = torch.cuda.Stream(device=device)
                    torch.cuda.set_stream(self._stream)
                engine.set_stream(torch.cuda.current_stream().cuda_stream)
            buffers = GpuBuffers(engine, torch, device)
        self.eng, self.buf = engine, buffers
        self._last = {}
        self.want_best = True  # trace()['best'] costs one tiny read-back per step; bench.py turns it off

    # -- pass-throughs
    def set_map(self, tree): self.eng.set_map(tree)
    def set_variant(self, v): self.eng.set_variant(v)
    def set_timing(self, e): self.eng.set_timing(e)
    def timers(self): return self.eng.timers()
    def motion_update(self, frame): self.eng.motion_update(frame)
    def synchronize(self): self.eng.synchronize()
    def trace(self): return dict(self._last)
    @property
    def pose(self): return self.eng.pose

    def _all_gather(self, dst, src, async_op=False):
        """Returns a work handle when async_op (wait() it before the result is used), else None."""
        if self.world == 1:
            dst.copy_(src)
            return None
        return self.dist.all_gather_into_tensor(dst, src, async_op=async_op) if async_op else \
            self.dist.all_gather_into_tensor(dst, src)

    def step(self, frame, scan):
        """One frame; the host synchronises once (in shard_finish), everything else is enqueued."""
        e, d, b = self.eng, self.dist, self.buf
        if e.shard_begin(frame, scan):          # first scan seeds the map (kernel.cu:1714-1717); replicated
            self._last = {"best": -1, "resampled": 0, "kd_size": e.kd_size}
            return
        if self.world > 1:
            d.all_reduce(b.stats[:2], op=d.ReduceOp.MAX)
        e.measurement_apply(fetch=False)        # weights + this rank's share of the best pose
        if self.world > 1:
            d.all_reduce(b.start, op=d.ReduceOp.SUM)
        # the weights are final after measurement_apply: gather them while the (replicated) map update runs
        pending = self._all_gather(b.gw, b.w, async_op=True)
        e.icp(None, fetch=False)                # adds the increment of the solve that ran under the score kernel
        e.shard_map()                           # replicated map update (device part): the all-gather runs under it
        if pending is not None:
            pending.wait()                      # stream-level: orders the compute stream behind the collective
        did, neff = e.shard_finish(frame)       # Neff on the global weights, header, the one host sync, resample plan
        if did:
            local, glob = b.pose_views()
            for dst, src in zip(glob, local):
                self._all_gather(dst, src)
            e.resample_gather()
        best = int(0xFFFFFFFF - (int(b.stats[0].item()) & 0xFFFFFFFF)) if self.want_best else -1
        self._last = {"best": best, "resampled": did, "neff": neff, "kd_size": e.kd_size}
