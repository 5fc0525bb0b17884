"""Multi-GPU stepping: particles shard over the ranks of one node (one process per GPU), the map, the scan,
the ICP solve and the map update are replicated, and the merges are three all-gathers per frame on a FIXED schedule.

Production (bench.py --gpus N, host/pfslam_mgpu.cpp): `native` = pkg.MgpuRank -- libpfslam_mgpu.so steps the rank, its RCCL
all-gathers launched straight into the frame's own streams (include/pfslam_mgpu.h).  Without it (the CPU tests under gloo, the
one-GPU gloo test) the same protocol runs through torch.distributed, each collective issued with the stream pfslam_shard_stream
names as torch's current stream.

Per frame (include/pfslam.h "multi-GPU"):
    all-gather  of 3 stride x f32       [x | y | theta] in one piece, right after the dispersion: on the particle stream, which has
                                        nothing else to do until the reduce -- it runs UNDER the scan-match kernel
    all-gather  of 16 B per rank        a shard's packed {max key, negated-min key}, straight from its reduce: every rank derives the
                                        global min / max / first-occurrence argmax, and reads the best particle's pose out of the
                                        gathered pose blocks
    all-gather  of stride x f32         weights -> global array (Neff, cdf and sampling run on it, replicated)
No call waits for the device and no collective depends on data: the resample is decided ON the device (Neff of the gathered
weights), its sources are read from the gathered pose blocks, and the frame is booked one step later from its pinned header --
the sharded frame IS the single-GPU frame of pfslam_step, in four parts with three collectives between them.
Rank r owns the global particles [r * stride, min((r + 1) * stride, N)), stride = ceil(N / world): the last shard may be
shorter, the exchange buffers are padded to `stride`.  Everything that touches randomness is keyed by GLOBAL particle
indices, and every sum runs on the global array in the canonical order, so the result is bit-identical for any number of
ranks.

`engine` is anything with the PfSlam shard interface (the GPU handle in production; the CPU tests plug the
oracle in to exercise this orchestration under gloo without a GPU)."""
import numpy as np


def shard_layout(n_global, world, rank):
    """(stride, offset, count) of rank's shard."""
    stride = (n_global + world - 1) // world
    off = rank * stride
    cnt = min(stride, n_global - off)
    if cnt <= 0:
        raise ValueError("%d particles over %d ranks leaves rank %d empty (stride %d)" % (n_global, world, rank, stride))
    return stride, off, cnt


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
            # wrapping a pointer costs ~20 us; the pose block only alternates between two addresses, so keep the wrappers
            ptr, nbytes = eng.device_ptr(which)
            key = (ptr, nbytes, typestr)
            t = cache.get(key)
            if t is None:
                t = cache[key] = torch.as_tensor(_DevView(ptr, nbytes, typestr, itemsize), device=torch.device("cuda", device)).view(dtype)
            return t

        self.stats = view(0, "<i8", 8, torch.int64)
        self.packs = view(15, "<i8", 8, torch.int64)    # world x 16 bytes: every rank's packed keys
        self.w = view(5, "<f4", 4, torch.float32)       # stride floats (padded)
        self.gw = view(10, "<f4", 4, torch.float32)     # world x stride
        self.eng, self._view = eng, view

    @property
    def pack(self):
        # this rank's 16-byte record (its packed max / negated-min keys) alternates between two addresses: ask after shard_score
        return self._view(14, "<i8", 8, self.torch.int64)

    def pose_blocks(self):
        # the local block [x | y | theta] swaps buffers on every resample, so re-query the pointer
        t = self.torch
        return self._view(16, "<f4", 4, t.float32), self._view(17, "<f4", 4, t.float32)

    def tree_buffers(self, n_nodes):
        """The device's map arrays, first n_nodes entries (hot records 16 B, parents, z / z-level links, weights: 4 B each) + the
        16-byte state: what the rank that re-balanced broadcasts to the others."""
        t = self.torch
        return [self._view(20, "<i4", 4, t.int32)[:4 * n_nodes], self._view(21, "<i4", 4, t.int32)[:n_nodes],
                self._view(22, "<i4", 4, t.int32)[:n_nodes], self._view(23, "<i4", 4, t.int32)[:n_nodes],
                self._view(24, "<i4", 4, t.int32)]


class ShardedSlam:
    def __init__(self, pkg, n_global, rank, world, device=0, dist=None, torch=None, engine=None, buffers=None, native_id=None, native=False, **kw):
        self.n_global, self.rank, self.world = n_global, rank, world
        self.stride, self.goff, self.n = shard_layout(n_global, world, rank)
        shard_layout(n_global, world, world - 1)  # the last rank must not be empty
        self.dist, self.torch = dist, torch
        self.native = None
        self._ext = {}
        if engine is None:
            engine = pkg.PfSlam(self.n, device=device, global_offset=self.goff, global_n=n_global, shard_stride=self.stride, **kw)
            if native or native_id is not None:
                # libpfslam_mgpu.so steps the rank: RCCL all-gathers launched straight into the frame's own streams
                self.native = pkg.MgpuRank(engine, world, rank, native_id)
            if torch is not None:
                buffers = GpuBuffers(engine, torch, device)
        self.eng, self.buf = engine, buffers
        self.device = device
        self.collectives = 0   # all-gathers issued: 3 in every frame (fixed schedule)
        # KDTree::Balance (frame % 100 == 5) ONCE per node: rank 0 builds, the others receive the device arrays (28 B per node)
        self._balance_builds = 0      # host builds this rank ran
        self._balance_broadcasts = 0  # re-balances this rank took part in
        if world > 1 and self.native is None:
            engine.set_shard_balance(True)

    @property
    def balance_builds(self):
        return self.native.stats()["balance_builds"] if self.native is not None else self._balance_builds

    @property
    def balance_broadcasts(self):
        return self.native.stats()["balance_broadcasts"] if self.native is not None else self._balance_broadcasts

    # -- pass-throughs
    def set_map(self, tree): self.eng.set_map(tree)
    def set_variant(self, v): self.eng.set_variant(v)
    def set_timing(self, e): self.eng.set_timing(e)
    def timers(self): return self.eng.timers()
    def motion_update(self, frame): self.eng.motion_update(frame)
    def synchronize(self): self.eng.synchronize()
    def trace(self): return self.eng.trace()   # books the frames in flight first (pfslam_get_trace)
    @property
    def pose(self): return self.eng.pose
    @property
    def kd_size(self): return self.eng.kd_size

    # -- BASELINE configs[4]: the topology graph and the loop-closure proposals (UpdateTopology / CheckLoopClosure, kernel.cu:623-795,
    # call sites 1750-1751) inside the SHARDED frame.  Both read only replicated state -- the frame's pose and the 2-D grid --, so every
    # rank keeps the same graph and proposes the same pairs; nothing is exchanged for them.
    def set_topology(self, mode=1):
        """1 = at the end of every frame (the frame is booked at once), 2 = with the booking of the frame, one step late."""
        self.eng.set_topology(mode)
    def closures(self): return self.eng.closures()
    def topology(self): return self.eng.topology()
    def map(self): return self.eng.map()
    def shift_particles(self, delta):
        """Odometry increment: every pose of every shard and the (replicated) robot pose move by the same (dx, dy, dtheta)."""
        self.eng.shift_particles(delta)
    def set_particles(self, p_global):
        """The GLOBAL particle array (every rank passes the same one); this rank keeps its shard."""
        self.eng.set_particles(np.ascontiguousarray(p_global[self.goff:self.goff + self.n]))
    def particles(self): return self.eng.particles()   # this rank's shard

    def _all_gather(self, dst, src, which):
        """Collective `which` (0 pose blocks, 1 keys, 2 weights) of the frame being enqueued, in the stream the engine names for it
        (GPU engines: pfslam_shard_stream -- stream order is all the ordering there is; gloo completes it before it returns)."""
        self.collectives += 1
        if self.world == 1:
            return
        st = self.eng.shard_stream(which) if hasattr(self.eng, "shard_stream") else None
        if st is None or self.torch is None or not src.is_cuda:
            self.dist.all_gather_into_tensor(dst, src)
            return
        ext = self._ext.get(st)
        if ext is None:
            ext = self._ext[st] = self.torch.cuda.ExternalStream(st, device=self.torch.device("cuda", self.device))
        with self.torch.cuda.stream(ext):
            self.dist.all_gather_into_tensor(dst, src)

    def step(self, frame, scan):
        """One frame, enqueued: nothing here waits for the device (see include/pfslam.h, pfslam_shard_*)."""
        if self.native is not None:
            self.native.step(frame, scan)
            self.collectives += 3
            return
        e, b = self.eng, self.buf
        if self.world > 1:
            due, n_nodes = e.shard_balance_due(frame)   # the same answer on every rank: frame number and (replicated) map size
            if due:
                if self.rank == 0:
                    e.shard_balance_build(frame)
                    self._balance_builds += 1
                for t in b.tree_buffers(n_nodes):       # rank 0's re-built map, straight from / into the device arrays
                    self.dist.broadcast(t, src=0)
                if self.rank != 0:
                    e.shard_balance_adopt()
                self._balance_broadcasts += 1
        if e.shard_disperse(frame, scan):       # first scan seeds the map (kernel.cu:1714-1717); replicated
            return
        if self.world == 1:                     # buffers 10 / 17 alias 5 / 16, nothing reads buffer 15: no collective at all
            self.collectives += 3
            e.shard_score()
            e.shard_weights()
            e.shard_finish()
            return
        local, glob = b.pose_blocks()           # the local block alternates between two allocations: ask every frame
        self._all_gather(glob, local, 0)        # the poses are final: gathered on the particle stream, under the scan-match kernel
        e.shard_score()                         # scan-match, reduce -> this shard's packed keys
        self._all_gather(b.packs, b.pack, 1)    # 16 bytes per rank, on the chain stream between the reduce and the walls
        e.shard_weights()                       # global min / max / argmax, pose = best + ICP increment, walls + insert; this shard's weights
        self._all_gather(b.gw, b.w, 2)          # the weights are final
        e.shard_finish()                        # Neff on the global weights, header, gated resample; the free cells' chain; booking

    def close(self):
        if self.native is not None:
            self.native.close()
            self.native = None
