"""Launcher for oracle/_ref/kernel_ref.hsaco: the REFERENCE's own src/kernel.cu compiled for gfx950, device code only (recipe and
caveats: oracle/kernel_ref_wrap.cpp, oracle/Makefile).  Test infrastructure -- the product never loads it.

Kernel arguments are packed here the way the HIP ABI lays out a kernel's parameters: each at its natural alignment, in order.
glm::ivec2 / vec3 are 4-aligned aggregates of 8 / 12 bytes; Patch (sceneStructs.h:40-45) is
{vec3 scale; vec3 resolution; MAP_TYPE *grid; unsigned char uid;} = 40 bytes, 8-aligned."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import oracle_lib as O

REF = os.path.join(O.ORACLE_DIR, "_ref")
PARTICLE_COUNT = 1000  # kernel.cu:30 (compiled in: the reference's kernels test i < PARTICLE_COUNT)
LIDAR_SIZE = 1081      # kernel.cu:43


def ptr(buf):
    return (struct.pack("<Q", buf.addr if isinstance(buf, DevBuf) else int(buf)), 8)


def i32(v):
    return (struct.pack("<i", int(v)), 4)


def f32(v):
    return (struct.pack("<f", float(v)), 4)


def boolean(v):
    return (struct.pack("<B", 1 if v else 0), 1)


def ivec2(a, b):
    return (struct.pack("<ii", int(a), int(b)), 4)


def vec3(a, b, c):
    return (struct.pack("<fff", float(a), float(b), float(c)), 4)


def patch(scale=(40.0, 40.0, 0.0), res=(0.025, 0.025, 0.0), grid=0, uid=0):
    return (struct.pack("<ffffffQB7x", scale[0], scale[1], scale[2], res[0], res[1], res[2], int(grid), uid), 8)


class DevBuf:
    def __init__(self, rk, arr=None, nbytes=None, dtype=None, shape=None):
        self.rk = rk
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            nbytes, dtype, shape = arr.nbytes, arr.dtype, arr.shape
        self.nbytes, self.dtype, self.shape = int(nbytes), dtype, shape
        p = C.c_void_p()
        assert rk.L.ref_dev_alloc(C.byref(p), self.nbytes) == 0, "device allocation failed"
        self.addr = p.value
        if arr is not None:
            assert rk.L.ref_h2d(C.c_void_p(self.addr), O.P(arr), self.nbytes) == 0
        else:
            assert rk.L.ref_dev_memset(C.c_void_p(self.addr), 0, self.nbytes) == 0

    def get(self):
        out = np.empty(self.shape, self.dtype)
        assert self.rk.L.ref_d2h(O.P(out), C.c_void_p(self.addr), self.nbytes) == 0
        return out

    def at(self, byte_offset):
        return self.addr + int(byte_offset)

    def free(self):
        if self.addr:
            self.rk.L.ref_dev_free(C.c_void_p(self.addr))
            self.addr = 0


class RefKernels:
    def __init__(self, fma=False):
        so = os.path.join(REF, "libhsaco_launcher.so")
        self.hsaco = os.path.join(REF, "kernel_ref_fma.hsaco" if fma else "kernel_ref.hsaco")
        symf = os.path.join(REF, "kernel_ref.symbols")
        if not (os.path.exists(so) and os.path.exists(self.hsaco) and os.path.exists(symf)):
            pytest.skip("oracle/_ref/kernel_ref.hsaco not built (needs /root/reference and hipify-perl at build time)")
        self.L = C.CDLL(so)
        self.L.ref_dev_alloc.argtypes = [C.c_void_p, C.c_size_t]
        self.L.ref_dev_free.argtypes = [C.c_void_p]
        self.L.ref_dev_memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.L.ref_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.L.ref_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.L.ref_launch.argtypes = [C.c_char_p, C.c_char_p, C.c_uint, C.c_uint, C.c_void_p, C.c_size_t]
        self.symbols = open(symf).read().split()
        self.bufs = []

    def sym(self, name, also=""):
        """mangled symbol of a kernel of kernel.cu by its source name (`also`: a substring that picks an overload)"""
        cands = [s for s in self.symbols if (s == name or s.startswith("_Z%d%s" % (len(name), name))) and also in s]
        assert len(cands) == 1, "kernel %s: %r" % (name, cands)
        return cands[0]

    def dev(self, arr):
        b = DevBuf(self, arr=arr)
        self.bufs.append(b)
        return b

    def zeros(self, shape, dtype):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        b = DevBuf(self, nbytes=int(np.prod(shape)) * np.dtype(dtype).itemsize, dtype=np.dtype(dtype), shape=shape)
        self.bufs.append(b)
        return b

    def launch(self, name, n_threads, block, *args, also=""):
        buf = b""
        for data, align in args:
            buf += b"\0" * (-len(buf) % align)
            buf += data
        buf += b"\0" * (-len(buf) % 8)
        cb = C.create_string_buffer(buf, len(buf))
        rc = self.L.ref_launch(self.hsaco.encode(), self.sym(name, also).encode(), (int(n_threads) + block - 1) // block, block, cb, len(buf))
        assert rc == 0, "reference kernel %s failed (code %d)" % (name, rc)

    def close(self):
        for b in self.bufs:
            b.free()
        self.bufs = []

    # ---- typed helpers ----
    def tree_dev(self, tree):
        """KDTree::Node array with one sentinel node in FRONT of it: the reference reads tree[tree[best].parent] with parent == -1
        when the root is the best node (H1).  The sentinel (axis 0, x = +inf) makes that read defined and the search stop --
        the restatement's definition of H1.  Returns (buffer, address of node 0)."""
        ext = np.zeros(len(tree) + 1, O.NODE_DTYPE)
        ext[1:] = tree
        ext[0] = (0, -1, -1, -1, np.inf, np.inf, np.inf, 0.0)
        b = self.dev(ext)
        return b, b.at(O.NODE_DTYPE.itemsize)

    def clean_lidar_scan(self, beam, scan, theta):
        beam = np.ascontiguousarray(beam, np.int32).ravel()
        scan = np.ascontiguousarray(scan, np.float32).ravel()
        theta = np.ascontiguousarray(theta, np.float32).ravel()
        n = len(beam)
        out = self.zeros((n, 2), np.float32)
        db, ds, dt = self.dev(beam), self.dev(scan), self.dev(theta)
        self.launch("ref_probe_clean_lidar_scan", n, 128, ptr(db), ptr(ds), ptr(dt), ptr(out), i32(n))
        r = out.get()
        for b in (db, ds, dt, out):
            b.free()
        return r
