"""ctypes binding of include/pfslam.h (libpfslam_hip.so).  No fallback of any kind."""
import ctypes as C
import os

import numpy as np

from . import build as _build

NODE_DTYPE = np.dtype(
    [("axis", "<i4"), ("left", "<i4"), ("right", "<i4"), ("parent", "<i4"),
     ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4")])
PARTICLE_DTYPE = np.dtype(
    {"names": ["x", "y", "theta", "w", "cluster", "map"],
     "formats": ["<f4", "<f4", "<f4", "<f4", "u1", "<u8"],
     "offsets": [0, 4, 8, 12, 16, 24], "itemsize": 32})

# every symbol include/pfslam.h declares (checked by tests/test_cabi_symbols.py)
SYMBOLS = [
    "pfslam_default_config", "pfslam_create", "pfslam_destroy", "pfslam_last_error", "pfslam_device_count",
    "pfslam_set_stream", "pfslam_synchronize", "pfslam_step", "pfslam_step_grid", "pfslam_shard_disperse", "pfslam_shard_score", "pfslam_shard_weights", "pfslam_shard_finish", "pfslam_shard_stream", "pfslam_get_pose", "pfslam_get_particles",
    "pfslam_get_map", "pfslam_get_grid", "pfslam_get_trace", "pfslam_get_cells", "pfslam_set_map",
    "pfslam_set_particles", "pfslam_shift_particles", "pfslam_set_scan", "pfslam_set_pose", "pfslam_set_grid", "pfslam_motion_update",
    "pfslam_score_kd", "pfslam_measurement_update", "pfslam_icp", "pfslam_update_map_kd", "pfslam_resample",
    "pfslam_score_grid", "pfslam_update_map_grid", "pfslam_traverse", "pfslam_measurement_local",
    "pfslam_measurement_apply", "pfslam_device_ptr", "pfslam_time_score_kd", "pfslam_set_variant", "pfslam_set_lag",
    "pfslam_kd_create", "pfslam_kd_insert_list", "pfslam_kd_insert_node", "pfslam_kd_balance", "pfslam_set_timing", "pfslam_get_timers", "pfslam_resample_plan", "pfslam_resample_gather", "pfslam_maybe_balance", "pfslam_kd_size", "pfslam_topology_update", "pfslam_find_walls",
    "pfslam_check_loop_closure", "pfslam_get_topology", "pfslam_set_topology", "pfslam_get_closures", "pfslam_score_census", "pfslam_set_census", "pfslam_get_census_log", "pfslam_ubench_gather", "pfslam_plan_stats", "pfslam_cell_stats", "pfslam_kd_parallel_sort", "pfslam_kd_sort_threads", "pfslam_kd_whole_node",
    "pfslam_set_serial", "pfslam_set_trig", "pfslam_debug_check_cells", "pfslam_set_probe", "pfslam_get_probe", "pfslam_probe_name", "pfslam_frame_mode",
    "pfslam_time_score_grid", "pfslam_set_shard_balance", "pfslam_shard_balance_due", "pfslam_shard_balance_build", "pfslam_shard_balance_adopt",
]


class Config(C.Structure):
    _fields_ = [("n_particles", C.c_int32), ("n_beams", C.c_int32),
                ("map_scale_x", C.c_float), ("map_scale_y", C.c_float),
                ("map_res_x", C.c_float), ("map_res_y", C.c_float),
                ("kd_capacity", C.c_int32), ("device", C.c_int32),
                ("strict_host_mirror", C.c_int32), ("free_upload_bug", C.c_int32),
                ("balance_period", C.c_int32), ("global_offset", C.c_int32), ("global_n", C.c_int32),
                ("shard_stride", C.c_int32), ("reserved_", C.c_int32 * 2)]


class PfSlamError(RuntimeError):
    pass


_lib = None


def _one_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  If this library initialises /opt/rocm's copy first,
    a later `import torch` in the same process sees "No HIP GPUs" (two runtimes).  When torch is installed but not
    yet imported, pre-load ITS runtime globally so that both bind to one copy.  No torch -> nothing to do."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("PFSLAM_NO_TORCH_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Build (if stale) and load libpfslam_hip.so.  Raises if that is not possible."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PFSLAM_LIB") or _build.LIB  # (PFSLAM_LIB: an A/B build of the same sources, tools/experiments)
    try:
        if path == _build.LIB and _build.stale():
            _build.build()
    except Exception as e:  # a prebuilt .so that travelled with the snapshot is still usable
        if not os.path.exists(path):
            raise PfSlamError("libpfslam_hip.so is missing and could not be built: %s" % e)
    _one_hip_runtime()
    L = C.CDLL(path)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    L.pfslam_last_error.restype = C.c_char_p
    L.pfslam_default_config.restype = None
    L.pfslam_default_config.argtypes = [vp]
    L.pfslam_create.argtypes = [vp, vp]
    L.pfslam_destroy.argtypes = [vp]
    L.pfslam_set_stream.argtypes = [vp, vp]
    L.pfslam_synchronize.argtypes = [vp]
    L.pfslam_step.argtypes = [vp, i32, vp]
    L.pfslam_step_grid.argtypes = [vp, i32, vp]
    L.pfslam_shard_disperse.argtypes = [vp, i32, vp, vp]
    L.pfslam_shard_score.argtypes = [vp]
    L.pfslam_shard_weights.argtypes = [vp]
    L.pfslam_shard_finish.argtypes = [vp]
    L.pfslam_shard_stream.argtypes = [vp, i32, vp]
    L.pfslam_time_score_grid.argtypes = [vp, i32, vp, vp]
    L.pfslam_set_shard_balance.argtypes = [vp, i32]
    L.pfslam_shard_balance_due.argtypes = [vp, i32, vp, vp]
    L.pfslam_shard_balance_build.argtypes = [vp, i32]
    L.pfslam_shard_balance_adopt.argtypes = [vp]
    L.pfslam_get_pose.argtypes = [vp, vp]
    L.pfslam_get_particles.argtypes = [vp, vp, vp]
    L.pfslam_get_map.argtypes = [vp, vp, vp]
    L.pfslam_get_grid.argtypes = [vp, vp, vp, vp]
    L.pfslam_get_trace.argtypes = [vp, vp]
    L.pfslam_get_cells.argtypes = [vp, i32, vp, i32, vp]
    L.pfslam_set_map.argtypes = [vp, vp, i32]
    L.pfslam_set_particles.argtypes = [vp, vp, i32]
    L.pfslam_set_scan.argtypes = [vp, vp, i32]
    L.pfslam_shift_particles.argtypes = [vp, vp]
    L.pfslam_set_pose.argtypes = [vp, vp]
    L.pfslam_set_grid.argtypes = [vp, vp, i32, i32]
    L.pfslam_motion_update.argtypes = [vp, i32]
    L.pfslam_score_kd.argtypes = [vp, vp]
    L.pfslam_measurement_update.argtypes = [vp, vp, vp, vp]
    L.pfslam_icp.argtypes = [vp, vp, vp, vp]
    L.pfslam_update_map_kd.argtypes = [vp]
    L.pfslam_resample.argtypes = [vp, i32, vp, vp]
    L.pfslam_resample_plan.argtypes = [vp, i32, vp, vp]
    L.pfslam_resample_gather.argtypes = [vp]
    L.pfslam_maybe_balance.argtypes = [vp, i32]
    L.pfslam_kd_size.argtypes = [vp]
    L.pfslam_topology_update.argtypes = [vp, vp]
    L.pfslam_find_walls.argtypes = [vp, vp, vp, vp]
    L.pfslam_check_loop_closure.argtypes = [vp, vp, i32, vp]
    L.pfslam_get_topology.argtypes = [vp, vp, i32, vp, vp]
    L.pfslam_set_topology.argtypes = [vp, i32]
    L.pfslam_get_closures.argtypes = [vp, vp, i32, vp]
    L.pfslam_score_census.argtypes = [vp, vp]
    L.pfslam_set_census.argtypes = [vp, i32]
    L.pfslam_get_census_log.argtypes = [vp, vp, i32, vp]
    L.pfslam_ubench_gather.argtypes = [vp, vp]
    L.pfslam_plan_stats.argtypes = [vp, vp]
    L.pfslam_cell_stats.argtypes = [vp, vp]
    L.pfslam_set_serial.argtypes = [vp, i32]
    L.pfslam_set_trig.argtypes = [vp, i32]
    L.pfslam_debug_check_cells.argtypes = [vp, vp]
    L.pfslam_set_probe.argtypes = [vp, i32]
    L.pfslam_get_probe.argtypes = [vp, vp, i32, vp, vp]
    L.pfslam_frame_mode.argtypes = [vp, vp]
    L.pfslam_probe_name.argtypes = [i32]
    L.pfslam_probe_name.restype = C.c_char_p
    L.pfslam_score_grid.argtypes = [vp, vp]
    L.pfslam_update_map_grid.argtypes = [vp]
    L.pfslam_traverse.argtypes = [vp, vp, i32, vp]
    L.pfslam_measurement_local.argtypes = [vp]
    L.pfslam_measurement_apply.argtypes = [vp, vp, vp, vp]
    L.pfslam_device_ptr.argtypes = [vp, i32, vp, vp]
    L.pfslam_time_score_kd.argtypes = [vp, i32, vp]
    L.pfslam_set_variant.argtypes = [vp, i32]
    L.pfslam_set_lag.argtypes = [vp, i32]
    L.pfslam_kd_create.argtypes = [vp, i32, vp]
    L.pfslam_kd_insert_node.argtypes = [vp, vp, i32]
    L.pfslam_kd_insert_list.argtypes = [vp, i32, vp, i32, i32]
    L.pfslam_kd_balance.argtypes = [vp, i32]
    L.pfslam_debug_math.argtypes = [vp, i32, vp, i32, vp]
    L.pfslam_set_timing.argtypes = [vp, i32]
    L.pfslam_get_timers.argtypes = [vp, vp]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _chk(rc, what=""):
    if rc != 0:
        raise PfSlamError("%s failed: %s" % (what, load().pfslam_last_error().decode()))


def device_count():
    return load().pfslam_device_count()


# ---- host-side map structure (no GPU needed) -------------------------------------------------
def kd_create(points_xyzw):
    pts = np.ascontiguousarray(points_xyzw, dtype=np.float32).reshape(-1, 4)
    out = np.zeros(len(pts), dtype=NODE_DTYPE)
    _chk(load().pfslam_kd_create(_p(pts), len(pts), _p(out)), "pfslam_kd_create")
    return out


def kd_insert_node(nodes, size, p4):
    p = np.ascontiguousarray(p4, dtype=np.float32)
    _chk(load().pfslam_kd_insert_node(_p(p), _p(nodes), size), "pfslam_kd_insert_node")


def kd_balance(nodes, size):
    _chk(load().pfslam_kd_balance(_p(nodes), size), "pfslam_kd_balance")


# ---- libpfslam_mgpu.so: the sharded frame with its all-gathers on librccl, launched into the frame's own streams (include/pfslam_mgpu.h)
MGPU_SYMBOLS = ["pfslam_mgpu_make_id", "pfslam_mgpu_create", "pfslam_mgpu_destroy", "pfslam_mgpu_step", "pfslam_mgpu_barrier_max",
                "pfslam_mgpu_stats", "pfslam_mgpu_time_collectives", "pfslam_mgpu_last_error"]
MGPU_ID_BYTES = 256
_mgpu = None


def load_mgpu():
    """ctypes handle of host/libpfslam_mgpu.so (built by `make -C gpu-icp-slam_amd/host`, which __graft_entry__.build() runs)."""
    global _mgpu
    if _mgpu is not None:
        return _mgpu
    load()  # (libpfslam_hip.so first: the multi-GPU library links it)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "libpfslam_mgpu.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.dirname(path), "libpfslam_mgpu.so"], stdout=subprocess.DEVNULL)
    M = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, i32 = C.c_void_p, C.c_int32
    M.pfslam_mgpu_last_error.restype = C.c_char_p
    M.pfslam_mgpu_make_id.argtypes = [vp]
    M.pfslam_mgpu_create.argtypes = [vp, i32, i32, vp, vp]
    M.pfslam_mgpu_destroy.argtypes = [vp]
    M.pfslam_mgpu_step.argtypes = [vp, i32, vp]
    M.pfslam_mgpu_barrier_max.argtypes = [vp, vp]
    M.pfslam_mgpu_stats.argtypes = [vp, vp]
    M.pfslam_mgpu_time_collectives.argtypes = [vp, i32, vp]
    _mgpu = M
    return M


def mgpu_make_id():
    """The job id rank 0 makes (two ncclUniqueId); the caller carries the bytes to the other ranks."""
    M = load_mgpu()
    buf = (C.c_ubyte * MGPU_ID_BYTES)()
    if M.pfslam_mgpu_make_id(buf):
        raise PfSlamError("pfslam_mgpu_make_id failed: %s" % M.pfslam_mgpu_last_error().decode())
    return bytes(buf)


class MgpuRank:
    """One rank of a sharded job stepped natively: pfslam_mgpu_step = the four pfslam_shard_* calls with the three RCCL all-gathers
    launched straight into the frame's own streams."""

    def __init__(self, eng, world, rank, job_id=None):
        self.M, self.eng = load_mgpu(), eng
        self._m = C.c_void_p(0)
        idbuf = (C.c_ubyte * MGPU_ID_BYTES).from_buffer_copy(job_id) if job_id is not None else None
        self._chk(self.M.pfslam_mgpu_create(idbuf, world, rank, eng._h, C.byref(self._m)), "pfslam_mgpu_create")

    def _chk(self, rc, what):
        if rc != 0:
            raise PfSlamError("%s failed: %s" % (what, self.M.pfslam_mgpu_last_error().decode()))

    def step(self, frame, scan):
        scan = np.ascontiguousarray(scan, dtype=np.float32)
        self._chk(self.M.pfslam_mgpu_step(self._m, frame, _p(scan)), "pfslam_mgpu_step")

    def barrier_max(self, value=None):
        v = C.c_double(0.0 if value is None else value)
        self._chk(self.M.pfslam_mgpu_barrier_max(self._m, C.byref(v) if value is not None else None), "pfslam_mgpu_barrier_max")
        return v.value

    def stats(self):
        out = (C.c_int * 4)()
        self._chk(self.M.pfslam_mgpu_stats(self._m, out), "pfslam_mgpu_stats")
        return {"collectives": out[0], "balance_builds": out[1], "balance_broadcasts": out[2], "world": out[3]}

    def time_collectives(self, reps=20):
        ms = (C.c_float * 3)()
        self._chk(self.M.pfslam_mgpu_time_collectives(self._m, reps, ms), "pfslam_mgpu_time_collectives")
        return {"pose_blocks": ms[0], "records": ms[1], "weights": ms[2]}

    def close(self):
        if self._m:
            self.M.pfslam_mgpu_destroy(self._m)
            self._m = C.c_void_p(0)


class PfSlam:
    """One handle = one GPU's shard of particles + a replica of the map."""

    def __init__(self, n_particles, n_beams=1081, kd_capacity=1 << 20, device=0, strict_host_mirror=1,
                 free_upload_bug=0, balance_period=100, global_offset=0, global_n=0, shard_stride=0, map_scale=None, map_res=None):
        L = load()
        cfg = Config()
        L.pfslam_default_config(C.byref(cfg))
        cfg.n_particles, cfg.n_beams, cfg.kd_capacity, cfg.device = n_particles, n_beams, kd_capacity, device
        if map_scale is not None:   # (x, y) extent of the map in metres (default 40 x 40, data/map_settings.txt)
            cfg.map_scale_x, cfg.map_scale_y = map_scale
        if map_res is not None:     # (x, y) cell size in metres (default 0.025)
            cfg.map_res_x, cfg.map_res_y = map_res
        cfg.strict_host_mirror, cfg.free_upload_bug, cfg.balance_period = strict_host_mirror, free_upload_bug, balance_period
        cfg.global_offset, cfg.global_n, cfg.shard_stride = global_offset, global_n, shard_stride
        self.cfg = cfg
        self.n, self.nb = n_particles, n_beams
        self._h = C.c_void_p()
        _chk(L.pfslam_create(C.byref(cfg), C.byref(self._h)), "pfslam_create")
        self.L = L

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.L.pfslam_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- uploads
    def set_map(self, nodes):
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
        _chk(self.L.pfslam_set_map(self._h, _p(nodes), len(nodes)), "pfslam_set_map")

    def set_particles(self, particles):
        particles = np.ascontiguousarray(particles, dtype=PARTICLE_DTYPE)
        _chk(self.L.pfslam_set_particles(self._h, _p(particles), len(particles)), "pfslam_set_particles")

    def set_scan(self, scan):
        scan = np.ascontiguousarray(scan, dtype=np.float32)
        _chk(self.L.pfslam_set_scan(self._h, _p(scan), len(scan)), "pfslam_set_scan")

    def set_pose(self, pose):
        pose = np.ascontiguousarray(pose, dtype=np.float32)
        _chk(self.L.pfslam_set_pose(self._h, _p(pose)), "pfslam_set_pose")

    def set_grid(self, grid):
        grid = np.ascontiguousarray(grid, dtype=np.int8)
        _chk(self.L.pfslam_set_grid(self._h, _p(grid), grid.shape[0], grid.shape[1]), "pfslam_set_grid")

    def set_stream(self, stream_ptr):
        _chk(self.L.pfslam_set_stream(self._h, C.c_void_p(stream_ptr)), "pfslam_set_stream")

    def set_variant(self, v):
        _chk(self.L.pfslam_set_variant(self._h, v), "pfslam_set_variant")

    def set_timing(self, enable):
        _chk(self.L.pfslam_set_timing(self._h, int(enable)), "pfslam_set_timing")

    def timers(self):
        out = (C.c_double * 12)()
        _chk(self.L.pfslam_get_timers(self._h, out), "pfslam_get_timers")
        d = {"score_ms": out[0], "score_launches": int(out[1])}
        for k, name in enumerate(("motion", "measurement", "map", "resample", "plan"), start=1):
            d[name + "_ms"] = out[2 * k]
            d[name + "_count"] = int(out[2 * k + 1])
        return d

    def set_lag(self, frames):
        """Frames pfslam_step may leave in flight (0 = every step books its own frame before it returns; default 1)."""
        _chk(self.L.pfslam_set_lag(self._h, frames), "pfslam_set_lag")

    def synchronize(self):
        _chk(self.L.pfslam_synchronize(self._h), "pfslam_synchronize")

    # -- stages
    def shift_particles(self, delta):
        """Odometry hook: every particle and robotPos += (dx, dy, dtheta)."""
        d = np.ascontiguousarray(delta, dtype=np.float32)
        _chk(self.L.pfslam_shift_particles(self._h, _p(d)), "pfslam_shift_particles")

    def motion_update(self, frame):
        _chk(self.L.pfslam_motion_update(self._h, frame), "pfslam_motion_update")

    def score_kd(self, fetch=True):
        if not fetch:
            _chk(self.L.pfslam_score_kd(self._h, None), "pfslam_score_kd")
            return None
        fit = np.empty(self.n, np.float32)
        _chk(self.L.pfslam_score_kd(self._h, _p(fit)), "pfslam_score_kd")
        return fit

    def time_score_grid(self, iters):
        a, b = C.c_float(), C.c_float()
        _chk(self.L.pfslam_time_score_grid(self._h, iters, C.byref(a), C.byref(b)), "pfslam_time_score_grid")
        return a.value, b.value

    def time_score_kd(self, iters):
        ms = C.c_float()
        _chk(self.L.pfslam_time_score_kd(self._h, iters, C.byref(ms)), "pfslam_time_score_kd")
        return ms.value

    def measurement_update(self):
        best, fmin, fmax = C.c_int(), C.c_float(), C.c_float()
        _chk(self.L.pfslam_measurement_update(self._h, C.byref(best), C.byref(fmin), C.byref(fmax)),
             "pfslam_measurement_update")
        return best.value, fmin.value, fmax.value

    def measurement_local(self):
        _chk(self.L.pfslam_measurement_local(self._h), "pfslam_measurement_local")

    def measurement_apply(self, fetch=True):
        if not fetch:  # no read-back, no host synchronisation
            _chk(self.L.pfslam_measurement_apply(self._h, None, None, None), "pfslam_measurement_apply")
            return None
        best, fmin, fmax = C.c_int(), C.c_float(), C.c_float()
        _chk(self.L.pfslam_measurement_apply(self._h, C.byref(best), C.byref(fmin), C.byref(fmax)),
             "pfslam_measurement_apply")
        return best.value, fmin.value, fmax.value

    def icp(self, start=None, fetch=True):
        """start=None: use the device-resident best-particle pose left by measurement_apply.
        fetch=False: launch only (no read-back, no host synchronisation)."""
        if not fetch:
            _chk(self.L.pfslam_icp(self._h, None, None, None), "pfslam_icp")
            return None
        out = np.zeros(3, np.float32)
        dbg = np.zeros(32, np.float32)
        if start is None:
            _chk(self.L.pfslam_icp(self._h, None, _p(out), _p(dbg)), "pfslam_icp")
        else:
            start = np.ascontiguousarray(start, dtype=np.float32)
            _chk(self.L.pfslam_icp(self._h, _p(start), _p(out), _p(dbg)), "pfslam_icp")
        return out, dbg

    def update_map_kd(self):
        _chk(self.L.pfslam_update_map_kd(self._h), "pfslam_update_map_kd")

    def resample(self, frame):
        did, neff = C.c_int(), C.c_float()
        _chk(self.L.pfslam_resample(self._h, frame, C.byref(did), C.byref(neff)), "pfslam_resample")
        return did.value, neff.value

    def maybe_balance(self, frame):
        _chk(self.L.pfslam_maybe_balance(self._h, frame), "pfslam_maybe_balance")

    @property
    def kd_size(self):
        return self.L.pfslam_kd_size(self._h)

    # -- topology graph / loop-closure proposal (kernel.cu:623-795)
    def topology_update(self):
        n = C.c_int()
        _chk(self.L.pfslam_topology_update(self._h, C.byref(n)), "pfslam_topology_update")
        return n.value

    def find_walls(self, a_xy, b_xy):
        a = np.ascontiguousarray(a_xy, np.float32); b = np.ascontiguousarray(b_xy, np.float32)
        n = C.c_int()
        _chk(self.L.pfslam_find_walls(self._h, _p(a), _p(b), C.byref(n)), "pfslam_find_walls")
        return n.value

    def check_loop_closure(self, cap=4096):
        pairs = np.zeros((cap, 2), np.int32)
        n = C.c_int()
        _chk(self.L.pfslam_check_loop_closure(self._h, _p(pairs), cap, C.byref(n)), "pfslam_check_loop_closure")
        return pairs[:min(n.value, cap)].copy()

    def set_topology(self, enable=True):
        """UpdateTopology + CheckLoopClosure at the end of every step() / step_grid() (kernel.cu:1750-1751)."""
        _chk(self.L.pfslam_set_topology(self._h, int(enable)), "pfslam_set_topology")

    def closures(self, cap=65536):
        pairs = np.zeros((cap, 2), np.int32)
        n = C.c_int()
        _chk(self.L.pfslam_get_closures(self._h, _p(pairs), cap, C.byref(n)), "pfslam_get_closures")
        return pairs[:min(n.value, cap)].copy()

    def topology(self, cap=4096):
        nodes = np.zeros((cap, 3), np.float32)
        n, idx = C.c_int(), C.c_int()
        _chk(self.L.pfslam_get_topology(self._h, _p(nodes), cap, C.byref(n), C.byref(idx)), "pfslam_get_topology")
        return nodes[:n.value].copy(), idx.value

    def score_census(self):
        """One counting launch of the score kernel on the current state (see include/pfslam.h)."""
        out = (C.c_ulonglong * 8)()
        _chk(self.L.pfslam_score_census(self._h, out), "pfslam_score_census")
        return {"trips": int(out[0]), "visits": int(out[1]), "tests": int(out[2]), "test_lanes": int(out[3]),
                "uniform_trips": int(out[4]), "prefix_trips": int(out[5]), "redescents": int(out[6]), "redescents_noop": int(out[7])}

    def set_census(self, enable=True):
        """Log what every scoring pass issues (counting instantiation on the same inputs, see include/pfslam.h)."""
        _chk(self.L.pfslam_set_census(self._h, int(enable)), "pfslam_set_census")

    def census_log(self, cap=1024):
        out = np.zeros((cap, 8), np.uint64)
        n = C.c_int()
        _chk(self.L.pfslam_get_census_log(self._h, _p(out), cap, C.byref(n)), "pfslam_get_census_log")
        keys = ("trips", "visits", "tests", "test_lanes", "uniform_trips", "prefix_trips", "redescents", "redescents_noop")
        return [dict(zip(keys, (int(v) for v in row))) for row in out[:min(n.value, cap)]]

    def plan_stats(self):
        out = (C.c_double * 10)()
        _chk(self.L.pfslam_plan_stats(self._h, out), "pfslam_plan_stats")
        keys = ("rows", "path_len", "candidates", "frac_complete", "frac_no_plan", "frac_full", "box_dx", "box_dy", "box_dtheta", "waves")
        return dict(zip(keys, [float(v) for v in out]))

    def cell_stats(self):
        """The persistent lattice-cell rows (all zero when the last scoring pass did not use them)."""
        out = (C.c_double * 16)()
        _chk(self.L.pfslam_cell_stats(self._h, out), "pfslam_cell_stats")
        keys = ("cells", "rows", "candidates", "redescent_candidates", "cells_without_row", "pool_slots", "window_kx", "window_ky",
                "walked_from_root", "extended", "reused", "claimed", "flags", "updates", "wipes", "suspended")
        return dict(zip(keys, [float(v) for v in out]))

    def set_serial(self, on):
        _chk(self.L.pfslam_set_serial(self._h, int(on)), "pfslam_set_serial")

    def check_cells(self):
        """Invariants of the persistent cell rows, checked on the device (include/pfslam.h): counts + violations (must be zero)."""
        out = (C.c_longlong * 16)()
        _chk(self.L.pfslam_debug_check_cells(self._h, out), "pfslam_debug_check_cells")
        keys = ("records", "unwalked", "fresh", "published", "row_words", "pending_words", "fallback_words", "dead",
                "v_unwalked_not_pending", "v_unpublished_word", "v_row_outside_alloc", "v_row_not_candidate", "v_watched_link_has_child",
                "v_bad_link", "v_dead_has_row", "v_counters")
        d = dict(zip(keys, [int(v) for v in out]))
        d["violations"] = sum(v for k, v in d.items() if k.startswith("v_"))
        return d

    def set_trig(self, devlib):
        """1: the device library's cosf / sinf / erfcinvf instead of the pf_math.h specification (include/pfslam.h, pfslam_set_trig)."""
        _chk(self.L.pfslam_set_trig(self._h, 1 if devlib else 0), "pfslam_set_trig")

    def frame_mode(self):
        out = (C.c_int * 4)()
        _chk(self.L.pfslam_frame_mode(self._h, out), "pfslam_frame_mode")
        return {"round5_frame": bool(out[0]), "gates": bool(out[1]), "serial": bool(out[2]), "publish_lag": int(out[3])}

    def set_probe(self, frames):
        _chk(self.L.pfslam_set_probe(self._h, int(frames)), "pfslam_set_probe")

    def probe(self, frames=4096):
        """Wall-clock stamps of the launches of the last round-5 frames: (names, array[f][slot], last ticket).  Absolute values of the device's
        100 MHz wall clock in microseconds (subtract a frame's scan-match stamp for a timeline: tools/frame_probe.py); 0 = that launch did not run in that frame."""
        buf = np.zeros((frames, 32), np.uint64)
        n, last = C.c_int(0), C.c_int(0)
        _chk(self.L.pfslam_get_probe(self._h, _p(buf), frames, C.byref(n), C.byref(last)), "pfslam_get_probe")
        names = []
        for k in range(32):
            s = self.L.pfslam_probe_name(k).decode()
            if not s:
                break
            names.append(s)
        t = buf[:n.value, :len(names)].astype(np.float64) * 0.01
        return names, t, last.value

    def ubench_gather(self):
        out = (C.c_double * 4)()
        _chk(self.L.pfslam_ubench_gather(self._h, out), "pfslam_ubench_gather")
        return {"wave_gathers_per_s": out[0], "cus": int(out[1]), "nominal_ghz": out[2], "cycles_per_wave_gather": out[3]}

    def resample_plan(self, frame):
        did, neff = C.c_int(), C.c_float()
        _chk(self.L.pfslam_resample_plan(self._h, frame, C.byref(did), C.byref(neff)), "pfslam_resample_plan")
        return did.value, neff.value

    def resample_gather(self):
        _chk(self.L.pfslam_resample_gather(self._h), "pfslam_resample_gather")

    def score_grid(self):
        fit = np.empty(self.n, np.int32)
        _chk(self.L.pfslam_score_grid(self._h, _p(fit)), "pfslam_score_grid")
        return fit

    def update_map_grid(self):
        _chk(self.L.pfslam_update_map_grid(self._h), "pfslam_update_map_grid")

    def step(self, frame, scan):
        scan = np.ascontiguousarray(scan, dtype=np.float32)
        _chk(self.L.pfslam_step(self._h, frame, _p(scan)), "pfslam_step")

    def shard_disperse(self, frame, scan):
        """Sharded frame (see include/pfslam.h): scan, re-balance, ICP fork, dispersion; True when the frame only seeded the map."""
        scan = np.ascontiguousarray(scan, dtype=np.float32)
        seeded = C.c_int(0)
        _chk(self.L.pfslam_shard_disperse(self._h, frame, _p(scan), C.byref(seeded)), "pfslam_shard_disperse")
        return bool(seeded.value)

    def shard_score(self):
        _chk(self.L.pfslam_shard_score(self._h), "pfslam_shard_score")

    def shard_weights(self):
        _chk(self.L.pfslam_shard_weights(self._h), "pfslam_shard_weights")

    def shard_finish(self):
        _chk(self.L.pfslam_shard_finish(self._h), "pfslam_shard_finish")

    def shard_stream(self, which):
        """hipStream_t (as an int) collective `which` (0 pose blocks, 1 keys, 2 weights) of the frame being enqueued goes into."""
        st = C.c_void_p(0)
        _chk(self.L.pfslam_shard_stream(self._h, which, C.byref(st)), "pfslam_shard_stream")
        return st.value or 0

    # -- multi-GPU re-balance: one host build per node (include/pfslam.h)
    def set_shard_balance(self, external):
        _chk(self.L.pfslam_set_shard_balance(self._h, 1 if external else 0), "pfslam_set_shard_balance")

    def shard_balance_due(self, frame):
        """(due, n_nodes); books the frames in flight when the re-balance period hits."""
        due, n = C.c_int(0), C.c_int(0)
        _chk(self.L.pfslam_shard_balance_due(self._h, frame, C.byref(due), C.byref(n)), "pfslam_shard_balance_due")
        return bool(due.value), n.value

    def shard_balance_build(self, frame):
        _chk(self.L.pfslam_shard_balance_build(self._h, frame), "pfslam_shard_balance_build")

    def shard_balance_adopt(self):
        _chk(self.L.pfslam_shard_balance_adopt(self._h), "pfslam_shard_balance_adopt")

    def step_grid(self, frame, scan):
        """One frame of the 2-D occupancy-grid variant (motion, grid score + weights, grid update, resample)."""
        scan = np.ascontiguousarray(scan, dtype=np.float32)
        _chk(self.L.pfslam_step_grid(self._h, frame, _p(scan)), "pfslam_step_grid")

    def traverse(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        best = np.empty(len(xyz), np.int32)
        _chk(self.L.pfslam_traverse(self._h, _p(xyz), len(xyz), _p(best)), "pfslam_traverse")
        return best

    def debug_math(self, which, x):
        x = np.ascontiguousarray(x, dtype=np.float32).ravel()
        out = np.empty(len(x) * (2 if which in (0, 6, 7) else 1), np.float32)
        _chk(self.L.pfslam_debug_math(self._h, which, _p(x), len(x), _p(out)), "pfslam_debug_math")
        if which == 6:
            return out.view(np.int32).reshape(-1, 2)
        return out.reshape(-1, 2) if which in (0, 7) else out

    # -- read-back
    @property
    def pose(self):
        out = np.zeros(3, np.float32)
        _chk(self.L.pfslam_get_pose(self._h, _p(out)), "pfslam_get_pose")
        return out

    def particles(self):
        ptr, n = C.c_void_p(), C.c_int()
        _chk(self.L.pfslam_get_particles(self._h, C.byref(ptr), C.byref(n)), "pfslam_get_particles")
        buf = (C.c_char * (32 * n.value)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=PARTICLE_DTYPE).copy()

    def map(self):
        ptr, n = C.c_void_p(), C.c_int()
        _chk(self.L.pfslam_get_map(self._h, C.byref(ptr), C.byref(n)), "pfslam_get_map")
        if n.value == 0:
            return np.zeros(0, NODE_DTYPE)
        buf = (C.c_char * (32 * n.value)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=NODE_DTYPE).copy()

    def grid(self):
        ptr, dx, dy = C.c_void_p(), C.c_int(), C.c_int()
        _chk(self.L.pfslam_get_grid(self._h, C.byref(ptr), C.byref(dx), C.byref(dy)), "pfslam_get_grid")
        buf = (C.c_char * (dx.value * dy.value)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.int8).reshape(dx.value, dy.value).copy()

    def trace(self):
        t = np.zeros(8, np.int32)
        _chk(self.L.pfslam_get_trace(self._h, _p(t)), "pfslam_get_trace")
        return {"best": int(t[0]), "resampled": int(t[1]), "n_wall": int(t[2]), "n_free": int(t[3]),
                "n_insert": int(t[4]), "neff": float(t[5:6].view(np.float32)[0]), "kd_size": int(t[6])}

    def cells(self, which):
        cap = 1600 * 1600
        out = np.empty(cap, np.int32)
        n = C.c_int()
        _chk(self.L.pfslam_get_cells(self._h, which, _p(out), cap, C.byref(n)), "pfslam_get_cells")
        return out[:n.value].copy()

    def device_ptr(self, which):
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        _chk(self.L.pfslam_device_ptr(self._h, which, C.byref(ptr), C.byref(nbytes)), "pfslam_device_ptr")
        return ptr.value, nbytes.value
