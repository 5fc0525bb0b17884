"""Build recipe for libpfslam_hip.so (hipcc, gfx950 only, in-tree so the .so travels with gpurun)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libpfslam_hip.so")
INC = os.path.join("..", "..", "include", "pfslam.h")
# source -> extra dependencies
UNITS = {
    "pfslam_hip.hip": ["pf_math.h", "kd_device.h", "pfslam_stages.hip.inc", INC],
    "sort_pairs.hip": [],
    "kd_host.cpp": [INC],
}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"] + os.environ.get("PFSLAM_EXTRA_FLAGS", "").split()


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libpfslam_hip.so cannot be built (there is no CPU fallback)")


def _mtime(path):
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _unit_stale(src, deps, obj):
    t = _mtime(obj)
    return t == 0.0 or any(_mtime(os.path.join(CSRC, f)) > t for f in [src] + deps)


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(_mtime(os.path.join(CSRC, f)) > t for src, deps in UNITS.items() for f in [src] + deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    objs = []
    for src, deps in UNITS.items():
        obj = os.path.join(OBJ, src + ".o")
        objs.append(obj)
        if force or _unit_stale(src, deps, obj):
            cmd = [cc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-lpthread", "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
