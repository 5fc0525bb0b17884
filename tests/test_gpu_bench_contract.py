"""bench.py prints ONE JSON line with the contract's fields (run small so it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line_has_contract_fields():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                                   "--particles", "5000", "--map-points", "20000", "--cpu-sample", "256"], cwd=ROOT).decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # the kernel is bound by the CU's gather path, not by HBM or MFMA: the block says so and every fraction in it is <= 1
    assert r["bound"] == "l1_gather" and r["unit"] == "GB/s" and r["launches"] == 3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] <= 1.02, r["frac"]
    g = r["gathers"]
    assert g["wave_gathers_16B_per_launch"] > 0 and 1.0 <= g["lanes_active_per_trip"] <= 64.0
    assert r["ubench"]["wave_gathers_per_s"] > 1e9 and r["ubench"]["cus"] >= 1
    assert r["hbm"] is None or 0.0 <= r["hbm"]["frac"] <= 1.0     # no committed PMC summary for this small test workload
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["alg_equiv"]["bytes_per_eval"] > 20
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and 1 <= c["cores"] <= c["logical_cpus"] and c["value"] > 0 and "sample" in c
    assert c["cpu_model"] and c["single_thread_value"] > 0
    lr = d["long_run"]
    assert lr["frames"] == 100 and lr["first_frame"] % 100 == 6 and lr["value"] > 0
    ph = d["phases_ms"]
    assert all(ph[k] >= 0 for k in ("motion", "measurement", "map", "resample")) and ph["measurement"] > ph["motion"]
    assert abs(d["value"] - 5000 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


def _torchrun(nproc, port, *bench_args):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3",
           "--warmup", "1", "--particles", "4096", "--map-points", "20000"] + list(bench_args)
    out = subprocess.check_output(cmd, cwd=ROOT, stderr=subprocess.STDOUT, timeout=600).decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_under_torchrun_rccl_one_rank():
    """The driver's launch form; one rank over RCCL exercises the sharded engine and the nccl process group."""
    d = _torchrun(1, 29551)
    assert d["n_gpus"] == 1 and d["config"]["particles_global"] == 4096 and d["roofline"]["launches"] == 3
    assert "cpu_baseline" in d


def test_bench_under_torchrun_two_ranks_on_one_gpu():
    """Two ranks (gloo, both on GPU 0 -- RCCL refuses two ranks per device): whole-job value counts both shards, rank 0
    prints the only line, and the roofline of rank 0's launches is still there."""
    d = _torchrun(2, 29552, "--backend", "gloo", "--same-device")
    assert d["n_gpus"] == 2 and d["config"]["particles_global"] == 8192 and d["scaling"] == "weak"
    assert abs(d["value"] - 8192 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    assert d["roofline"]["launches"] == 3 and "cpu_baseline" not in d
