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
    assert 0.0 < r["frac_of_nominal_peak"] <= 1.02 and r["peak_nominal"] > 0
    g = r["census"]
    # the gathers are counted on the timed launches themselves: a replay of the same frames, bit-identical state
    assert g["launches"] == 3 and g["replay_identical"] is True
    assert g["wave_gathers_16B_per_launch"] > 0 and 1.0 <= g["lanes_active_per_trip"] <= 64.0
    b = g["gather_bytes_per_launch"]
    assert 0 < b["min"] <= b["mean"] <= b["max"]
    assert abs(r["achieved_lane_bytes"] - b["mean"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved_lane_bytes"] < 1e-9
    assert abs(r["achieved"] - g["wave_gathers_per_launch"] * 1024.0 / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-9   # every wave gather as 16 B x 64
    assert r["ubench"]["wave_gathers_per_s"] > 1e9 and r["ubench"]["cus"] >= 1
    assert r["pmc"] is None                                       # no committed PMC summary for this small test workload
    assert r["traffic"] is None
    assert r["alg_equiv"]["bytes_per_eval"] > 20
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and 1 <= c["cores"] <= c["logical_cpus"] and c["value"] > 0 and "sample" in c
    assert c["cpu_model"] and c["single_thread_value"] > 0
    lr = d["long_run"]
    assert lr["frames"] == 100 and lr["first_frame"] % 100 == 6 and lr["value"] > 0
    assert d["value_long_run"] == lr["value"] and list(d)[:3] == ["metric", "value", "value_long_run"]
    ph = d["phases_ms"]
    assert all(ph[k] >= 0 for k in ("motion", "measurement", "map", "resample")) and ph["measurement"] > ph["motion"]
    assert abs(d["value"] - 5000 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


def _torchrun(nproc, port, *bench_args):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3",
           "--warmup", "1", "--particles", "4096", "--map-points", "20000"] + list(bench_args)
    # (the process-group bootstrap has been seen to hang on some leases -- 2 of 8 boxes in round 5, never
    # reproducible in a loop of the same command on another box --: a watchdog inside bench.py, a bounded wait here, and one retry on a fresh port)
    out = None
    for attempt in range(2):
        cmd[cmd.index("--master-port") + 1] = str(port + 20 * attempt)
        try:
            out = subprocess.check_output(cmd, cwd=ROOT, stderr=subprocess.STDOUT, timeout=300, env=dict(os.environ, PFSLAM_BENCH_WATCHDOG="240")).decode()
            break
        except (subprocess.TimeoutExpired, subprocess.CalledProcessError) as e:
            last = e
    assert out is not None, getattr(last, "output", b"")[-3000:]
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
    assert d["roofline"]["launches"] == 3
    # the N > 1 line explains itself -- every rank's own wall clock, the collectives timed on their own, and ONE KDTree::Balance in the
    # whole job (the long-run leg contains frame 105).  The CPU leg is rank 0 at N = 1 only (the other ranks would wait for it).
    assert "cpu_baseline" not in d and len(d["per_rank_ms_per_step"]) == 2
    assert set(d["collectives"]["ms"]) == {"pose_blocks", "records", "weights"} and all(len(v) == 2 for v in d["collectives"]["ms"].values())
    assert d["balance"]["host_builds_on_rank0"] == d["balance"]["broadcasts"] >= 1


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun: bench.py starts the two ranks itself (gloo, both on GPU 0) and rank 0's
    line is the only one printed."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--particles", "4096",
           "--map-points", "20000", "--backend", "gloo", "--same-device"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.check_output(cmd, cwd=ROOT, stderr=subprocess.STDOUT, timeout=600, env=env).decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["particles_global"] == 8192 and d["roofline"]["launches"] == 3
    assert d["roofline"]["census"]["replay_identical"] is True


def test_roofline_census_agrees_with_committed_pmc():
    """The default workload's census (counted by the replay inside bench.py) against the newest matching rocprofv3 summary under
    profiles/: TA_BUFFER_READ_WAVEFRONTS_sum per launch is the counter's view of the same wave gathers.  A reader must be able to
    recompute `frac` from that file alone."""
    import glob
    if not glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_pmc_score_kd.json")):
        pytest.skip("no round-3+ PMC summary of the default workload committed yet")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"], cwd=ROOT, timeout=900).decode()
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    r = d["roofline"]
    assert r["census"]["replay_identical"] is True and r["launches"] == 20
    p = r["pmc"]
    assert p is not None and p["ta_buffer_read_wavefronts_per_launch"], "profiles/ has no matching PMC summary with the TA counters"
    assert abs(p["census_over_pmc_wavefronts"] - 1.0) < 0.10, p
    # recompute the fraction from the PMC file alone: every wave gather priced as a 16-byte one (the counter cannot tell the
    # 4-byte parent-index gathers apart) / the timed launches' duration in the kernel-trace pass / the line's peak
    pm = json.load(open(os.path.join(ROOT, p["source"])))
    frac_pmc = pm["avg_per_launch"]["TA_BUFFER_READ_WAVEFRONTS_sum"] * 1024.0 / (pm["kernel_ms"] * 1e-3) / 1e9 / r["peak"]
    assert abs(frac_pmc - p["frac_from_pmc_only"]) < 1e-9
    # the counter agrees with the census to 10 % (above); the two KERNEL TIMES are from different runs, and since the cell rows persist
    # the scan-match kernel's time differs by up to ~10 % from run to run on one box (tools/experiments/r04/README.md): 20 %
    assert abs(frac_pmc / r["frac_all_gathers_as_16B"] - 1.0) < 0.20, (frac_pmc, r["frac_all_gathers_as_16B"])
    # round 5: `frac` itself prices every wave gather as one 16 B x 64 lanes request (what the counter sees); the lane-level bytes --
    # the 4-byte gathers (cell-table words, parent indices: a third of all wave gathers) at 256 B -- stay beside it
    assert r["frac"] == r["frac_all_gathers_as_16B"] and r["frac_lane_bytes"] <= r["frac"] <= 1.6 * r["frac_lane_bytes"]
    # the pmc block says that it is replayed from profiles/, and its VALU figure is the calibrated one (no assumed cycle count)
    assert p["replayed"] is True and p["source"].startswith("profiles/")
    v = p["valu_issue_busy_mix_weighted"]
    assert v and 0.3 < v["lo"] <= v["hi"] < 1.0 and v["calibration"].startswith("profiles/")
    assert abs(p["sq_active_inst_valu_over_insts_valu"] - 1.0) < 0.02
    # where the frame's time goes, from the frame's own kernels
    assert d["frame"]["frames"] >= 10 and 60.0 < d["frame"]["chain_us_mean"] < 170.0, d["frame"]
