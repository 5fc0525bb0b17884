"""CPU-side check of the reference-kernel checker's build products (oracle/_ref/, built by `make -C oracle` when /root/reference is
present): the code object exists beside its symbol list, every kernel of kernel.cu the GPU tests launch is in it, and no scratch
copy of the reference's text is left behind to travel with the tree."""
import os

import pytest

import oracle_lib as O

REF = os.path.join(O.ORACLE_DIR, "_ref")
WANTED = ["kernEvaluateParticlesKD", "findCorrespondenceIndexKD", "findCorrespondenceKD", "kernGetWalls", "kernGetWallsKD", "kernWeightedSample",
          "kernAddNoise", "kernUpdateWeights", "kernCopyWeights", "kernUpdateMapKD", "kernTestCorrespondance", "kernEvaluateParticles",
          "kernUpdateMap"]


def test_reference_kernel_code_object():
    symf = os.path.join(REF, "kernel_ref.symbols")
    if not os.path.exists(symf):
        pytest.skip("oracle/_ref/kernel_ref.hsaco not built (needs /root/reference and hipify-perl at build time)")
    syms = open(symf).read().split()
    for k in WANTED:
        assert any(s.startswith("_Z%d%s" % (len(k), k)) for s in syms), k
    for k in ("ref_probe_clean_lidar_scan", "ref_probe_trace_ray", "ref_probe_hyperplane", "ref_probe_rng"):
        assert k in syms
    assert os.path.getsize(os.path.join(REF, "kernel_ref.hsaco")) > 1 << 20
    assert not os.path.isdir(os.path.join(REF, "build")), "scratch copies of the reference's text must not outlive the build"
    import ctypes
    L = ctypes.CDLL(os.path.join(REF, "libhsaco_launcher.so"))
    for f in ("ref_launch", "ref_dev_alloc", "ref_dev_free", "ref_h2d", "ref_d2h", "ref_dev_memset", "ref_svd3_gpu"):
        assert hasattr(L, f)


def test_reference_kernel_code_object_is_not_stale():
    """The code object the GPU tests launch was built from the reference's text AS IT IS NOW: oracle/Makefile leaves the sha256 of
    kernel.cu and of every header it includes beside the code object; here (where /root/reference exists) they are recomputed."""
    import hashlib
    ref_src = "/root/reference/src"
    shaf = os.path.join(REF, "kernel_ref.sha256")
    if not os.path.isdir(ref_src) or not os.path.exists(os.path.join(REF, "kernel_ref.hsaco")):
        pytest.skip("needs /root/reference and a built oracle/_ref/kernel_ref.hsaco")
    assert os.path.exists(shaf), "oracle/_ref/kernel_ref.sha256 missing: rebuild with `make -C oracle`"
    rows = [l.split() for l in open(shaf).read().splitlines() if l.strip()]
    assert any(name == "kernel.cu" for _, name in rows)
    for digest, name in rows:
        with open(os.path.join(ref_src, name), "rb") as fh:
            assert hashlib.sha256(fh.read()).hexdigest() == digest, "%s changed since kernel_ref.hsaco was built: `make -C oracle`" % name
    assert os.path.getmtime(os.path.join(REF, "kernel_ref.hsaco")) >= os.path.getmtime(shaf) - 600
