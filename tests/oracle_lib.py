"""ctypes doorway to the CPU oracle (oracle/liboracle.so) and, when built, to the
reference-derived checkers under oracle/_ref/.  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

NODE_DTYPE = np.dtype(
    [("axis", "<i4"), ("left", "<i4"), ("right", "<i4"), ("parent", "<i4"),
     ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4")])
PARTICLE_DTYPE = np.dtype(
    {"names": ["x", "y", "theta", "w", "cluster", "map"],
     "formats": ["<f4", "<f4", "<f4", "<f4", "u1", "<u8"],
     "offsets": [0, 4, 8, 12, 16, 24], "itemsize": 32})
assert NODE_DTYPE.itemsize == 32 and PARTICLE_DTYPE.itemsize == 32


class Patch(C.Structure):
    _fields_ = [("scale_x", C.c_float), ("scale_y", C.c_float), ("res_x", C.c_float), ("res_y", C.c_float)]


class SlamConfig(C.Structure):
    _fields_ = [("n_particles", C.c_int), ("n_beams", C.c_int), ("patch", Patch), ("kd_capacity", C.c_int),
                ("strict_host_mirror", C.c_int), ("free_upload_bug", C.c_int), ("balance_period", C.c_int)]


def default_patch():
    return Patch(40.0, 40.0, 0.025, 0.025)


def build_oracle(force=False):
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return so


_lib = None


def P(a, t=None):
    return a.ctypes.data_as(C.c_void_p)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build_oracle())
    vp, i32, f32, u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32
    L.orc_utilhash.restype = u32; L.orc_utilhash.argtypes = [u32]
    L.orc_engine_seed.restype = u32; L.orc_engine_seed.argtypes = [i32, i32, i32]
    L.orc_minstd_next.restype = u32; L.orc_minstd_next.argtypes = [vp]
    L.orc_uniform_real.restype = f32; L.orc_uniform_real.argtypes = [vp, f32, f32]
    L.orc_normal.restype = f32; L.orc_normal.argtypes = [vp, f32, f32]
    L.orc_sincosf.restype = None; L.orc_sincosf.argtypes = [f32, vp, vp]
    L.orc_log.restype = C.c_double; L.orc_log.argtypes = [C.c_double]
    L.orc_ndtri.restype = C.c_double; L.orc_ndtri.argtypes = [C.c_double]
    L.orc_erfcinvf.restype = f32; L.orc_erfcinvf.argtypes = [f32]
    L.orc_asinf.restype = f32; L.orc_asinf.argtypes = [f32]
    L.orc_rsqrtf.restype = f32; L.orc_rsqrtf.argtypes = [f32]
    L.orc_sum_f32.restype = f32; L.orc_sum_f32.argtypes = [vp, i32, i32]
    L.orc_inclusive_scan_f32.restype = None; L.orc_inclusive_scan_f32.argtypes = [vp, i32, vp]
    L.orc_add_noise.restype = None; L.orc_add_noise.argtypes = [vp, i32, i32, i32]
    L.orc_clean_lidar_scan.restype = None; L.orc_clean_lidar_scan.argtypes = [i32, f32, f32, vp, vp]
    L.orc_kd_traverse.restype = i32; L.orc_kd_traverse.argtypes = [vp, f32, f32, f32, vp]
    L.orc_score_kd.restype = None; L.orc_score_kd.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
    L.orc_score_kd_mt.restype = None; L.orc_score_kd_mt.argtypes = [vp, vp, i32, vp, i32, vp, i32]
    L.orc_traverse_batch.restype = None; L.orc_traverse_batch.argtypes = [vp, vp, i32, vp, vp]
    L.orc_minmax_first_f32.restype = None; L.orc_minmax_first_f32.argtypes = [vp, i32, vp, vp]
    L.orc_minmax_first_i32.restype = None; L.orc_minmax_first_i32.argtypes = [vp, i32, vp, vp]
    L.orc_update_weights_f32.restype = None; L.orc_update_weights_f32.argtypes = [vp, i32, vp, f32, i32]
    L.orc_update_weights_i32.restype = None; L.orc_update_weights_i32.argtypes = [vp, i32, vp, f32, i32]
    L.orc_svd3.restype = None; L.orc_svd3.argtypes = [vp, vp, vp, vp]
    L.orc_icp.restype = None; L.orc_icp.argtypes = [vp, vp, vp, vp, i32, vp, vp]
    L.orc_trace_ray.restype = None; L.orc_trace_ray.argtypes = [i32, i32, i32, i32, i32, i32, vp]
    L.orc_get_walls.restype = None; L.orc_get_walls.argtypes = [vp, i32, i32, i32, f32, vp, vp, i32, i32, f32, f32]
    L.orc_masks_to_points.restype = None; L.orc_masks_to_points.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.orc_update_map_kd.restype = None; L.orc_update_map_kd.argtypes = [vp, vp, vp, i32, i32, vp]
    L.orc_test_correspondence.restype = None; L.orc_test_correspondence.argtypes = [vp, vp, vp, i32, vp, vp]
    L.orc_kd_insert_node.restype = None; L.orc_kd_insert_node.argtypes = [vp, vp, i32]
    L.orc_kd_create.restype = None; L.orc_kd_create.argtypes = [vp, i32, vp]
    L.orc_kd_balance.restype = None; L.orc_kd_balance.argtypes = [vp, i32]
    L.orc_resample.restype = i32; L.orc_resample.argtypes = [vp, i32, i32, vp, vp]
    L.orc_weighted_sample_indices.restype = None
    L.orc_weighted_sample_indices.argtypes = [vp, i32, f32, i32, i32, i32, vp]
    L.orc_score_grid.restype = None; L.orc_score_grid.argtypes = [vp, i32, i32, vp, vp, i32, vp, i32, vp]
    L.orc_update_map_grid.restype = None; L.orc_update_map_grid.argtypes = [vp, i32, i32, vp, vp, vp, i32]
    L.orc_topology_init.restype = None; L.orc_topology_init.argtypes = [vp]
    L.orc_topology_update.restype = i32; L.orc_topology_update.argtypes = [vp, vp]
    L.orc_find_walls.restype = i32; L.orc_find_walls.argtypes = [vp, i32, i32, vp, vp, vp]
    L.orc_check_loop_closure.restype = i32; L.orc_check_loop_closure.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32]
    L.orc_slam_create.restype = vp; L.orc_slam_create.argtypes = [vp]
    L.orc_slam_destroy.restype = None; L.orc_slam_destroy.argtypes = [vp]
    L.orc_slam_set_map.restype = None; L.orc_slam_set_map.argtypes = [vp, vp, i32]
    L.orc_slam_step.restype = None; L.orc_slam_step.argtypes = [vp, i32, vp]
    L.orc_slam_step_grid.restype = None; L.orc_slam_step_grid.argtypes = [vp, i32, vp]
    L.orc_slam_step_grid_cpu.restype = None; L.orc_slam_step_grid_cpu.argtypes = [vp, i32, vp]
    L.orc_slam_set_grid.restype = None; L.orc_slam_set_grid.argtypes = [vp, vp]
    L.orc_slam_set_topology.restype = None; L.orc_slam_set_topology.argtypes = [vp, i32]
    L.orc_slam_last_closures.restype = i32; L.orc_slam_last_closures.argtypes = [vp, vp, i32]
    L.orc_slam_topology.restype = vp; L.orc_slam_topology.argtypes = [vp]
    L.orc_slam_grid.restype = vp; L.orc_slam_grid.argtypes = [vp]
    L.orc_slam_set_particles.restype = None; L.orc_slam_set_particles.argtypes = [vp, vp]
    L.orc_slam_shift_particles.restype = None; L.orc_slam_shift_particles.argtypes = [vp, vp]
    L.orc_slam_get_pose.restype = None; L.orc_slam_get_pose.argtypes = [vp, vp]
    L.orc_slam_kd_size.restype = i32; L.orc_slam_kd_size.argtypes = [vp]
    L.orc_slam_tree.restype = vp; L.orc_slam_tree.argtypes = [vp]
    L.orc_slam_particles.restype = vp; L.orc_slam_particles.argtypes = [vp]
    L.orc_slam_last_trace.restype = None; L.orc_slam_last_trace.argtypes = [vp, vp]
    L.orc_slam_last_cells.restype = i32; L.orc_slam_last_cells.argtypes = [vp, i32, vp, i32]
    _lib = L
    return L


# ---------------------------------------------------------------- numpy-level helpers
def sincosf(x):
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    s = np.empty_like(x); c = np.empty_like(x)
    L = lib()
    sf, cf = C.c_float(), C.c_float()
    for i, v in enumerate(x):
        L.orc_sincosf(float(v), C.byref(sf), C.byref(cf))
        s[i], c[i] = sf.value, cf.value
    return s, c


def kd_create(points_xyzw):
    pts = np.ascontiguousarray(points_xyzw, dtype=np.float32).reshape(-1, 4)
    nodes = np.zeros(len(pts), dtype=NODE_DTYPE)
    lib().orc_kd_create(P(pts), len(pts), P(nodes))
    return nodes


def kd_insert(nodes, size, p4):
    p = np.ascontiguousarray(p4, dtype=np.float32)
    lib().orc_kd_insert_node(P(p), P(nodes), size)


def traverse_batch(tree, xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    best = np.empty(len(xyz), np.int32); vis = np.empty(len(xyz), np.int32)
    lib().orc_traverse_batch(P(tree), P(xyz), len(xyz), P(best), P(vis))
    return best, vis


def score_kd(tree, particles, scan, threads=1, stats=False):
    scan = np.ascontiguousarray(scan, dtype=np.float32)
    fit = np.zeros(len(particles), np.float32)
    if threads > 1:
        lib().orc_score_kd_mt(P(tree), P(particles), len(particles), P(scan), len(scan), P(fit), threads)
        return fit
    nv, nb = C.c_uint64(0), C.c_uint64(0)
    lib().orc_score_kd(P(tree), P(particles), len(particles), P(scan), len(scan), P(fit), C.byref(nv), C.byref(nb))
    if stats:
        return fit, nv.value, nb.value
    return fit


def make_particles(n, x=0.0, y=0.0, theta=0.0, w=1.0):
    p = np.zeros(n, dtype=PARTICLE_DTYPE)
    p["x"], p["y"], p["theta"], p["w"] = x, y, theta, w
    return p


def add_noise(particles, frame, idx0=0):
    lib().orc_add_noise(P(particles), len(particles), frame, idx0)
    return particles


def icp(tree, robot, start, scan):
    scan = np.ascontiguousarray(scan, dtype=np.float32)
    robot = np.ascontiguousarray(robot, dtype=np.float32); start = np.ascontiguousarray(start, dtype=np.float32)
    out = np.zeros(3, np.float32); dbg = np.zeros(32, np.float32)
    lib().orc_icp(P(tree), P(robot), P(start), P(scan), len(scan), P(out), P(dbg))
    return out, dbg


def get_walls(scan, cx, cy, theta, dimx=1600, dimy=1600, res=0.025):
    scan = np.ascontiguousarray(scan, dtype=np.float32)
    fm = np.zeros(dimx * dimy, np.uint8); wm = np.zeros(dimx * dimy, np.uint8)
    lib().orc_get_walls(P(scan), len(scan), cx, cy, float(theta), P(fm), P(wm), dimx, dimy, res, res)
    return fm, wm


TOPO_MAX = 4096


class Topology(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("node_idx", C.c_int), ("pos", (C.c_float * 2) * TOPO_MAX),
                ("dist", C.c_float * TOPO_MAX), ("n_edges", C.c_int * TOPO_MAX), ("edges", (C.c_int * 8) * TOPO_MAX)]

    def __init__(self):
        super().__init__()
        lib().orc_topology_init(C.byref(self))

    def update(self, robot):
        r = np.ascontiguousarray(robot, np.float32)
        return lib().orc_topology_update(C.byref(self), P(r))

    def nodes(self):
        return np.array([[self.pos[k][0], self.pos[k][1], self.dist[k]] for k in range(self.n_nodes)], np.float32)

    def loop_closure(self, grid, robot, cap=4096):
        r = np.ascontiguousarray(robot, np.float32)
        pairs = np.zeros((cap, 2), np.int32)
        patch = default_patch()
        n = lib().orc_check_loop_closure(C.byref(self), P(grid), grid.shape[0], grid.shape[1], C.byref(patch), P(r), P(pairs), cap)
        return pairs[:min(n, cap)].copy()


def find_walls(grid, a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    patch = default_patch()
    return lib().orc_find_walls(P(grid), grid.shape[0], grid.shape[1], C.byref(patch), P(a), P(b))


class Slam:
    """Whole-step oracle (particleFilter, kernel.cu:1702-1762)."""

    def __init__(self, n_particles, n_beams=1081, kd_capacity=1 << 20, strict_host_mirror=1,
                 free_upload_bug=0, balance_period=100, patch=None):
        self.cfg = SlamConfig(n_particles, n_beams, patch if patch is not None else default_patch(), kd_capacity, strict_host_mirror,
                              free_upload_bug, balance_period)
        self.h = lib().orc_slam_create(C.byref(self.cfg))
        self.n = n_particles

    def close(self):
        if self.h:
            lib().orc_slam_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_map(self, tree):
        lib().orc_slam_set_map(self.h, P(tree), len(tree))

    def step(self, frame, scan):
        scan = np.ascontiguousarray(scan, dtype=np.float32)
        lib().orc_slam_step(self.h, frame, P(scan))

    def set_particles(self, p):
        assert len(p) == self.n
        lib().orc_slam_set_particles(self.h, P(np.ascontiguousarray(p)))

    def shift_particles(self, delta):
        lib().orc_slam_shift_particles(self.h, P(np.ascontiguousarray(delta, dtype=np.float32)))

    def step_grid(self, frame, scan):
        scan = np.ascontiguousarray(scan, dtype=np.float32)
        lib().orc_slam_step_grid(self.h, frame, P(scan))

    def step_grid_cpu(self, frame, scan):
        """The reference's CPU branches of the 2-D frame loop (kernel.cu:340-369, 487-508, 578-620; H7)."""
        scan = np.ascontiguousarray(scan, dtype=np.float32)
        lib().orc_slam_step_grid_cpu(self.h, frame, P(scan))

    def set_topology(self, enable=True):
        lib().orc_slam_set_topology(self.h, int(enable))

    def closures(self, cap=65536):
        pairs = np.zeros((cap, 2), np.int32)
        n = lib().orc_slam_last_closures(self.h, P(pairs), cap)
        return pairs[:min(n, cap)].copy()

    def topology(self):
        """(nodes as (x, y, dist) rows, index of the current node)"""
        t = Topology.from_address(lib().orc_slam_topology(self.h))
        return t.nodes(), t.node_idx

    def set_grid(self, grid):
        grid = np.ascontiguousarray(grid, dtype=np.int8)
        lib().orc_slam_set_grid(self.h, P(grid))

    @property
    def grid(self):
        ptr = lib().orc_slam_grid(self.h)
        p = self.cfg.patch
        dimx, dimy = int(np.float32(p.scale_x) / np.float32(p.res_x)), int(np.float32(p.scale_y) / np.float32(p.res_y))
        buf = (C.c_char * (dimx * dimy)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.int8).reshape(dimx, dimy).copy()

    @property
    def pose(self):
        out = np.zeros(3, np.float32); lib().orc_slam_get_pose(self.h, P(out)); return out

    @property
    def kd_size(self):
        return lib().orc_slam_kd_size(self.h)

    def tree(self):
        n = self.kd_size
        ptr = lib().orc_slam_tree(self.h)
        buf = (C.c_char * (32 * n)).from_address(ptr)
        return np.frombuffer(buf, dtype=NODE_DTYPE).copy()

    def particles(self):
        ptr = lib().orc_slam_particles(self.h)
        buf = (C.c_char * (32 * self.n)).from_address(ptr)
        return np.frombuffer(buf, dtype=PARTICLE_DTYPE).copy()

    def trace(self):
        t = np.zeros(8, np.int32); lib().orc_slam_last_trace(self.h, P(t))
        return {"best": int(t[0]), "resampled": int(t[1]), "n_wall": int(t[2]), "n_free": int(t[3]),
                "n_insert": int(t[4]), "neff": float(t[5:6].view(np.float32)[0]), "kd_size": int(t[6])}

    def cells(self, which):
        cap = 1600 * 1600
        out = np.empty(cap, np.int32)
        k = lib().orc_slam_last_cells(self.h, which, P(out), cap)
        return out[:k].copy()


# ---------------------------------------------------------------- reference-derived checkers
def ref_kdtree():
    so = os.path.join(ORACLE_DIR, "_ref", "libkdtree_ref.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    L.ref_node_size.restype = C.c_int
    L.ref_kd_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.ref_kd_insert_node.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ref_kd_balance.argtypes = [C.c_void_p, C.c_int]
    return L


def thrust_probe():
    so = os.path.join(ORACLE_DIR, "_ref", "libthrust_probe.so")
    if not os.path.exists(so):
        return None
    try:
        L = C.CDLL(so)
    except OSError:
        return None
    L.tp_minstd.argtypes = [C.c_uint, C.c_int, C.c_void_p]
    L.tp_uniform.argtypes = [C.c_uint, C.c_float, C.c_float, C.c_int, C.c_void_p]
    L.tp_normal3.argtypes = [C.c_uint, C.c_float, C.c_float, C.c_float, C.c_void_p]
    return L
