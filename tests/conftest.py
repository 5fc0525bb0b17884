import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name has a hyphen, hence importlib)."""
    return importlib.import_module("gpu-icp-slam_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def small_world(pkg):
    """4k-point synthetic map (tree built by the product's host code), segments, a scan."""
    pts, segs = pkg.synth.make_map_points(4000, seed=11)
    tree = pkg.kd_create(pts)
    scan = pkg.synth.make_scan(segs, (0.1, -0.2, 0.3), seed=12)
    return {"pts": pts, "segs": segs, "tree": tree, "scan": scan}
