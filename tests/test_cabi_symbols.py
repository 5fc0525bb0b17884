"""The C-ABI library loads on a machine without a GPU and exports every symbol include/pfslam.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pfslam.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pfslam_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.load()
    names = declared_symbols()
    assert len(names) >= 35
    for name in names:
        assert hasattr(L, name), "libpfslam_hip.so does not export %s" % name
    assert set(pkg.binding.SYMBOLS) == set(names)


def test_mgpu_library_exports_every_declared_symbol(pkg):
    """include/pfslam_mgpu.h (the sharded frame with its all-gathers on librccl) against host/libpfslam_mgpu.so: loads without a GPU."""
    src = open(os.path.join(ROOT, "include", "pfslam_mgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(pfslam_mgpu_[a-z0-9_]+)\s*\(", src)))
    M = pkg.load_mgpu()
    assert len(names) == 8
    for name in names:
        assert hasattr(M, name), "libpfslam_mgpu.so does not export %s" % name
    assert set(pkg.binding.MGPU_SYMBOLS) == set(names)


def test_no_gpu_is_a_loud_error_not_a_fallback(pkg):
    if pkg.device_count() > 0:
        return  # on the GPU box this is covered by the gpu tests
    try:
        pkg.PfSlam(64)
    except pkg.PfSlamError as e:
        assert "no HIP device" in str(e) or "hip" in str(e).lower()
    else:
        raise AssertionError("creating a handle without a GPU must fail")


def test_struct_layouts_match_reference(pkg):
    # KDTree::Node 32 B (kdtree.hpp:16-27), Particle 32 B with w@12, cluster@16, map@24 (sceneStructs.h:33-38)
    assert pkg.NODE_DTYPE.itemsize == 32
    assert pkg.PARTICLE_DTYPE.itemsize == 32
    assert pkg.PARTICLE_DTYPE.fields["w"][1] == 12
    assert pkg.PARTICLE_DTYPE.fields["cluster"][1] == 16
    assert pkg.PARTICLE_DTYPE.fields["map"][1] == 24
    assert ctypes.sizeof(pkg.Config) == 64


def test_whole_node_thread_budget():
    """One KDTree::Balance per node (rank 0 builds, the others wait in the broadcast): the build may use every usable core, not the
    rank's 1 / LOCAL_WORLD_SIZE share (pfslam_kd_whole_node, set by pfslam_shard_balance_build around its build)."""
    import subprocess, sys, os
    code = ("import importlib, sys; sys.path.insert(0, %r); pkg = importlib.import_module('gpu-icp-slam_amd'); L = pkg.load();"
            "a = L.pfslam_kd_sort_threads(); L.pfslam_kd_whole_node(1); b = L.pfslam_kd_sort_threads(); L.pfslam_kd_whole_node(0);"
            "c = L.pfslam_kd_sort_threads(); print(a, b, c)" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, LOCAL_WORLD_SIZE="8")
    env.pop("PFSLAM_SORT_THREADS", None)
    a, b, c = map(int, subprocess.check_output([sys.executable, "-c", code], env=env).split()[-3:])
    one = int(subprocess.check_output([sys.executable, "-c", code], env=dict(env, LOCAL_WORLD_SIZE="1")).split()[-3])
    assert a == c == max(1, min(one, 64) // 8 if one >= 8 else 1) and b == one


def test_environment_switches_are_the_documented_set():
    """Housekeeping of round 6: the product build reads only the test / diagnostic switches tools/README.md lists; the A/B knobs of past
    experiments go through ab_env(), which is getenv only under -DPF_EXPERIMENTS (not among the build's flags)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    allowed = {"PFSLAM_SERIAL", "PFSLAM_GATES", "PFSLAM_FAULT", "PFSLAM_STABLE_ORDER", "PFSLAM_MARK_EARLY", "PFSLAM_PUBLISH_LAG", "PFSLAM_VARIANT",
               "PFSLAM_PLAN_MIN_N", "PFSLAM_CELL_LIST_CAP", "PFSLAM_CELL_POOL_CAP", "PFSLAM_SORT_THREADS", "PFSLAM_PLAIN_SORT", "PFSLAM_VERBOSE"}
    read = set()
    for f in glob.glob(os.path.join(root, "gpu-icp-slam_amd", "csrc", "*")):
        if f.endswith((".hip", ".inc", ".h", ".cpp")):
            src = open(f).read()
            read |= set(re.findall(r'[^_a-z]getenv\("(PFSLAM_[A-Z0-9_]+)"\)', src))
    assert read <= allowed, "undocumented environment switch in the product build: %s" % sorted(read - allowed)
    readme = open(os.path.join(root, "tools", "README.md")).read()
    assert all(name in readme for name in read)
    build = open(os.path.join(root, "gpu-icp-slam_amd", "build.py")).read()
    assert "PF_EXPERIMENTS" not in build.split("FLAGS =")[1].split("\n")[0]
