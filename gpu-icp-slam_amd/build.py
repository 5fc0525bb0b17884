"""Build recipe for libpfslam_hip.so (hipcc, gfx950 only, in-tree so the .so travels with gpurun)."""
import fcntl
import hashlib
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
    "pfslam_hip.hip": ["pf_math.h", "kd_device.h", "kd_cells.hip.inc", "pfslam_stages.hip.inc", "pfslam_frame.hip.inc", INC],
    "kd_host.cpp": [INC],
}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"] + os.environ.get("PFSLAM_EXTRA_FLAGS", "").split()


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libpfslam_hip.so cannot be built (there is no CPU fallback)")


def _digest(files):
    """Content hash of sources + flags: staleness must not depend on file mtimes (the tree is copied between machines)."""
    h = hashlib.sha1(" ".join(FLAGS).encode())
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def _unit_digest(src, deps):
    return _digest([src] + deps)


def _read(path):
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return ""


def stale():
    if not os.path.exists(LIB):
        return True
    return _read(os.path.join(OBJ, "lib.stamp")) != _digest(sorted({f for s, d in UNITS.items() for f in [s] + d}))


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    # one builder at a time: the ranks of a torchrun job import the package concurrently
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not stale():
            return LIB
        cc = hipcc()
        objs = []
        for src, deps in UNITS.items():
            obj = os.path.join(OBJ, src + ".o")
            stamp = obj + ".stamp"
            objs.append(obj)
            want = _unit_digest(src, deps)
            if force or not os.path.exists(obj) or _read(stamp) != want:
                cmd = [cc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                with open(stamp, "w") as fh:
                    fh.write(want)
        tmp = LIB + ".tmp.%d" % os.getpid()
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-lpthread", "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)  # atomic: a concurrent loader never sees a half-written library
        with open(os.path.join(OBJ, "lib.stamp"), "w") as fh:
            fh.write(_digest(sorted({f for s, d in UNITS.items() for f in [s] + d})))
        return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
