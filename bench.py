#!/usr/bin/env python3
"""bench.py -- particle-scan evaluations per second of the MI355X particle-filter SLAM step.

  python bench.py --gpus N --steps K --warmup W
      N > 1: either started by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
      (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or started plainly -- then bench.py launches its own N ranks
      (one child process per GPU with the same environment variables set) and relays rank 0's line.

A "step" is one full particleFilter() frame of the KD / point-cloud path (kernel.cu:1702-1762): dispersion,
1081-beam scan-match of every particle against the KD map, min/max/argmax + weight update, single-step ICP/SVD
pose, Bresenham map update (with the insert of the new walls, on the device), Neff + weighted resample.  Workload (BASELINE.json
north_star / configs[2], synthetic because data/train_lidar*.mat is absent from the reference checkout):
100 000 particles per GPU x 1081-beam synthetic scans against a 100 000-point KD map.  Particles shard over the
GPUs (weak scaling); the map and scan are replicated; the merges are three small RCCL all-gathers per frame.
BASELINE configs[3]'s per-GPU share is `--particles 125000 --map-points 500000`.

One JSON line on rank 0:  value = particles scored per second over the whole job, with the map, particles and
all state resident in HBM (the 4.3 KB scan per frame is the step API's input and is inside the timed region).

"roofline" prices the dominant kernel (the scan-match kernel) against the resource that binds it -- the CU's gather path (texture
addresser: 4 lane addresses per clock for 64-bit and wider loads, i.e. 64 B/clk per CU for the 16-byte node records).
Everything in it is measured on THE TIMED LAUNCHES: the kernel time by HIP events around them, the gathers they issue by a
REPLAY -- a second handle steps through the same frames (the frame loop is deterministic: same particles, scans and map bit for
bit, checked) with the counting instantiation of the kernel run behind every scoring pass (pfslam_set_census) --, the chip's
gather rate by a micro-benchmark (pfslam_ubench_gather).  `frac` is stated against that measured rate AND against the nominal
256 CU x 64 B/clk x clock.  Sub-blocks: "pmc" = counters of the newest matching rocprofv3 summary under profiles/ (never a
fixed file): HBM bytes against the 8 TB/s peak, TA busy, and TA_BUFFER_READ_WAVEFRONTS, which must agree with the census;
"alg_equiv" = SURVEY 8d's algorithmic bytes (B_valid x V x 32 B + 20 B per evaluation), which are L1/L2 hits and therefore NOT a
fraction of anything.
"cpu_baseline" times the CPU oracle's restatement of the same scoring loop on this box's usable host cores (a reported
baseline, not the target); "long_run" is the same step over a whole 100-frame KDTree::Balance cycle (its rate is also the
line's third key, `value_long_run`).
"""
import argparse
import glob
import importlib
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FIRST_FRAME = 6        # frame numbers seed the RNG and decide the frame % 100 == 5 re-balance; see "long_run" in the line


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=100000, help="particles per GPU")
    ap.add_argument("--map-points", type=int, default=100000)
    ap.add_argument("--cpu-sample", type=int, default=0, help="particles in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs, the long-run leg and the extras (profiling runs)")
    ap.add_argument("--pmc-file", default="", help="rocprofv3 PMC summary to take roofline.traffic from (default: newest matching profiles/r*_pmc_score_kd*.json)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--backend", default="", help="torch.distributed backend of the process group (default: gloo with --collectives rccl -- the group only "
                    "carries the RCCL job id and each rank's wall time, the frame's collectives are the job's own RCCL communicators --, nccl with --collectives torch)")
    ap.add_argument("--same-device", action="store_true", help="debug: every rank uses GPU 0 (--collectives torch --backend gloo)")
    ap.add_argument("--cloud-sigma", type=float, default=0.0, help="start from a Gaussian cloud of this spread (m; heading: sigma / 8 rad) instead of the dispersed start cloud")
    ap.add_argument("--topology", type=int, default=0, help="1 | 2: UpdateTopology + CheckLoopClosure inside the frame (kernel.cu:1750-1751; BASELINE configs[4]); "
                    "the line then carries the graph size and the loop-closure proposals of the last timed frame")
    ap.add_argument("--collectives", default="rccl", choices=["rccl", "torch"],
                    help="who issues the frame's all-gathers: rccl = libpfslam_mgpu.so launches them straight into the frame's own streams (default); "
                         "torch = torch.distributed with the frame's stream as the current stream (debugging; what --backend gloo / --same-device use)")
    ap.add_argument("--dry-collectives", action="store_true", help="also time the frame's collectives on their own (per-rank wall times in the line); with one rank: the fields, no traffic")
    return ap.parse_args()


def host_cpu():
    """Usable host cores (scheduler affinity AND the cgroup CPU quota) and the CPU model."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    if quota:
        cores = max(1, min(cores, int(quota + 0.5)))
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return cores, model, os.cpu_count() or cores


def cpu_baseline(O, tree, particles, scan, sample, map_points):
    """Oracle restatement of EvaluateParticleKD (the reference has no CPU version of it) on a bounded sample: one thread
    (the reference's CPU style) and one thread per usable core."""
    cores, model, logical = host_cpu()
    one = min(len(particles), 512)
    t0 = time.perf_counter()
    fit1, visits, valid = O.score_kd(tree, particles[:one], scan, stats=True)
    t1 = time.perf_counter() - t0
    rate1 = one / t1
    # does the lease really deliver `cores` cores?  try the affinity count and fall back to what scales
    if sample <= 0:  # ~10-20 s of CPU core time, at least 64 particles per thread
        sample = int(min(len(particles), max(one, 64 * cores, rate1 * 12)))
    best = None
    for _ in range(3):  # best of 3 (thread start-up noise)
        t0 = time.perf_counter()
        O.score_kd(tree, particles[:sample], scan, threads=cores)
        tm = time.perf_counter() - t0
        best = tm if best is None else min(best, tm)
    return {
        "value": sample / best, "unit": "particle-scan evals/s", "cores": cores, "kind": "port",
        "cpu_model": model, "logical_cpus": logical,
        "single_thread_value": rate1, "all_core_speedup": (sample / best) / rate1,
        "sample": "%d particles x 1081 beams on the end-of-run %d-point map, oracle restatement of kernEvaluateParticlesKD "
                  "(gcc -O3 -mavx2 -mfma), %d pthreads = usable cores (sched affinity and cgroup quota; %d logical CPUs); "
                  "single thread: %.0f evals/s on %d particles" % (sample, map_points, cores, logical, rate1, one),
    }, visits / max(valid, 1), valid / one


def pmc_busy(pmc, avgp, name, units, clock_ghz, quad):
    """Busy fraction of `units` hardware units from a cycle counter summed over them: counter (x 4 if it counts quad-cycles) / units /
    cycles, cycles = the kernel's mean duration inside the PMC pass that collected the counter x the measured clock."""
    v, ms = avgp.get(name), (pmc.get("pass_kernel_ms") or {}).get(name)
    if not v or not ms or not clock_ghz:
        return None
    return v * (4.0 if quad else 1.0) / units / (ms * 1e-3 * clock_ghz * 1e9)


def valu_mix_busy(pmc, avgp, cus, clock_ghz):
    """VALU issue cycles of the scan-match kernel, mix-weighted: SQ_INSTS_VALU_* counts x the measured cost of each kind
    (profiles/rNN_valu_calibration.json, tools/valu_calibrate.sh) / SIMDs / kernel cycles.  INT32 mixes 2.4-cycle adds and 4.2-cycle
    shifts / bit-field ops: bracketed."""
    try:
        cal_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_calibration.json")))
        if not cal_files or not clock_ghz:
            return None
        kinds = json.load(open(cal_files[-1]))["kinds"]
        c = lambda name: kinds[name]["cycles_per_wave_instruction_per_simd"]
        fast, slow = (c("v_fma_f32") + c("v_mul_f32") + c("v_add_f32")) / 3.0, (c("v_cndmask_b32") + c("v_cvt_f32_i32") + c("v_fma_f64") + c("v_mul_f64")) / 4.0
        total = avgp["SQ_INSTS_VALU"]
        f32 = avgp.get("SQ_INSTS_VALU_ADD_F32", 0.0) + avgp.get("SQ_INSTS_VALU_MUL_F32", 0.0) + avgp.get("SQ_INSTS_VALU_FMA_F32", 0.0)
        i32 = avgp.get("SQ_INSTS_VALU_INT32", 0.0)
        rest = total - f32 - i32
        ms = (pmc.get("pass_kernel_ms") or {}).get("SQ_INSTS_VALU") or pmc.get("kernel_ms")
        cycles = ms * 1e-3 * clock_ghz * 1e9 * 4.0 * cus
        return {"lo": (f32 * fast + i32 * fast + rest * slow) / cycles, "hi": (f32 * fast + i32 * slow + rest * slow) / cycles,
                "cycles_fp32_add_mul_fma": fast, "cycles_other": slow, "insts_fp32_add_mul_fma": f32, "insts_int32": i32, "insts_other": rest,
                "calibration": os.path.relpath(cal_files[-1], ROOT)}
    except Exception:
        return None


def find_pmc(explicit, n_local, map_points):
    """Newest committed rocprofv3 PMC summary of k_score_kd for THIS workload (profiles/rNN_pmc_score_kd*.json)."""
    if explicit:
        return explicit if os.path.exists(explicit) else None
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_score_kd*.json")):
        m = re.match(r"r(\d+)_", os.path.basename(f))
        try:
            d = json.load(open(f))
        except Exception:
            continue
        wl = d.get("workload", {"particles": 100000, "map_points": 100000})  # r01 predates the key: default workload
        if wl.get("particles") != n_local or wl.get("map_points") != map_points:
            continue
        key = (int(m.group(1)) if m else 0, os.path.getmtime(f))
        if best is None or key > best[0]:
            best = (key, f)
    return best[1] if best else None


def extras(pkg, O, tree, pts, scan, device):
    """Side measurements for DESIGN.md (not part of the contract line's metric): the host map structure next to the
    reference's own kdtree.cpp (oracle/_ref, kind "reference"), and the 2-D grid path (BASELINE configs[0-1]) with the
    reference's CPU branches (H7) timed beside pfslam_step_grid."""
    ex = {}
    try:
        t0 = time.perf_counter(); pkg.kd_create(pts); t1 = time.perf_counter() - t0
        ex["kd_create_ms"] = {"points": len(pts), "product_host": t1 * 1e3}
        ref = O.ref_kdtree()
        if ref is not None:
            buf = np.zeros(len(pts), O.NODE_DTYPE)
            t0 = time.perf_counter(); ref.ref_kd_create(O.P(pts), len(pts), O.P(buf)); t2 = time.perf_counter() - t0
            ex["kd_create_ms"]["reference_kdtree_cpp"] = t2 * 1e3
            ex["kd_create_ms"]["identical_output"] = bool(buf.tobytes() == pkg.kd_create(pts).tobytes())
        # grid path: 10 k particles, 1600x1600 int8 grid rasterised from the same walls
        n = 10000
        grid = np.full((1600, 1600), -100, np.int8)
        gx = np.clip(np.round(800 + pts[:, 0] / 0.025).astype(int), 0, 1599); gy = np.clip(np.round(800 + pts[:, 1] / 0.025).astype(int), 0, 1599)
        grid[gx, gy] = 113
        h = pkg.PfSlam(n, device=device)
        p = O.make_particles(n); O.add_noise(p, 1)
        h.set_grid(grid); h.set_particles(p); h.set_scan(scan)
        h.score_grid(); h.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            h.L.pfslam_score_grid(h._h, None)
        h.synchronize()
        dt = (time.perf_counter() - t0) / 20
        fit = h.score_grid()
        want = np.zeros(256, np.int32)
        import ctypes as C
        patch = O.default_patch()
        O.lib().orc_score_grid(O.P(grid), 1600, 1600, C.byref(patch), O.P(p[:256].copy()), 256, O.P(scan), 1081, O.P(want))
        t0 = time.perf_counter()
        O.lib().orc_score_grid(O.P(grid), 1600, 1600, C.byref(patch), O.P(p[:2048].copy()), 2048, O.P(scan), 1081, O.P(np.zeros(2048, np.int32)))
        tc = time.perf_counter() - t0
        ex["grid_path_10k_particles"] = {"gpu_evals_per_s": n / dt, "ms_per_call_incl_minmax_weights": dt * 1e3,
                                         "cpu_oracle_1thread_evals_per_s": 2048 / tc, "parity_sample_ok": bool((fit[:256] == want).all())}
        h.close()
        # whole 2-D frame loop (BASELINE configs[1]: 10 k particles) and configs[0]'s 50 particles, on a short drive
        _, frames = pkg.synth.corridor_sequence(60, seed=5)
        for nn, key in ((10000, "grid_step_10k_particles"), (50, "grid_step_50_particles")):
            h = pkg.PfSlam(nn, device=device)
            for f in range(1, 11):
                h.step_grid(f, frames[f - 1][1])
            h.synchronize()
            t0 = time.perf_counter()
            for f in range(11, 61):
                h.step_grid(f, frames[f - 1][1])
            h.synchronize()
            dt = (time.perf_counter() - t0) / 50
            ex[key] = {"ms_per_frame": dt * 1e3, "evals_per_s": nn / dt}
            h.close()
        # BASELINE configs[0]: 50 particles on the reference's own CPU branches (GPU_* == 0, kernel.cu:340-369, 578-620,
        # 487-508; H7 semantics), one thread -- next to pfslam_step_grid at 50 particles above
        for label, fn in (("cpu_branch_reference_semantics", "step_grid_cpu"), ("gpu_branch_semantics_on_cpu", "step_grid")):
            o = O.Slam(50)
            step = getattr(o, fn)
            for f in range(1, 11):
                step(f, frames[f - 1][1])
            t0 = time.perf_counter()
            for f in range(11, 61):
                step(f, frames[f - 1][1])
            ex["grid_step_50_particles"]["cpu_oracle_1thread_ms_per_frame_" + label] = (time.perf_counter() - t0) / 50 * 1e3
            o.close()
        ex["grid_step_50_particles"]["note"] = ("configs[0]: reference CPU 2-D path (oracle restatement of the GPU_*==0 branches, "
                                                "1 thread) vs pfslam_step_grid on the GPU, 50 particles, 50 frames")
    except Exception as e:  # extras must never break the contract line
        ex["error"] = repr(e)
    return ex


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (what torch.distributed.run would do: one
    process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set) and wait for them.  Rank 0 prints the
    line; the children inherit stdout / stderr."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PFSLAM_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                r = p.poll()
                if r is None:
                    continue
                procs.remove(p)
                if r != 0:          # one rank failed: the others would wait in a collective forever
                    rc = rc or r
                    for q in procs:
                        q.kill()
            time.sleep(0.05)
    finally:
        for q in procs:             # only ever the exact children started above
            q.kill()
    return rc


def main():
    # (multi-process GPU work on this host driver needs dmabuf IPC -- RCCL's hipIpcGetMemHandle fails otherwise; read by the HSA runtime
    # when it initialises, i.e. before the package's library makes its first HIP call)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("PFSLAM_BENCH_WATCHDOG"):  # debugging aid: every thread's traceback after N seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PFSLAM_BENCH_WATCHDOG"]), exit=True)
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        if not a.same_device:
            ndev = importlib.import_module("gpu-icp-slam_amd").device_count()
            if a.gpus > ndev:
                raise SystemExit("--gpus %d but only %d device(s) are visible: refusing to launch ranks" % (a.gpus, ndev))
        raise SystemExit(self_launch(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if a.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(a.gpus, 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start bench.py plainly (it launches its own ranks) or with "
                         "`python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...`" % (a.gpus, world, a.gpus, a.gpus))
    pkg = importlib.import_module("gpu-icp-slam_amd")
    if pkg.device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: libpfslam_hip.so has no CPU fallback")
    if not a.same_device and max(a.gpus, 1) > pkg.device_count():
        raise SystemExit("--gpus %d but only %d device(s) are visible: refusing to start (a rank without a GPU of its own would hang "
                         "the others in ncclCommInitRank)" % (a.gpus, pkg.device_count()))

    dist = None
    torch = None
    distributed = world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ)  # torchrun, even with 1 rank
    if a.same_device and a.collectives == "rccl":
        a.collectives = "torch"          # (RCCL refuses two ranks on one device)
    native = distributed and a.collectives == "rccl"
    backend = a.backend or ("gloo" if (native or a.same_device) else "nccl")
    cuda_tensors = False                 # the process group moves device tensors (only with torch's own collectives)
    if distributed:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # one node: the bootstrap over loopback (the container's other interfaces may not route)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            cuda_tensors = True
        else:
            # With the job's own RCCL communicators (libpfslam_mgpu.so) the process group is bookkeeping on the host: the job id, each rank's
            # wall time.  No torch CUDA context, no second RCCL instance with streams of its own beside the frame's four.
            dist.init_process_group(backend, rank=rank, world_size=world)
            if not native:
                torch.cuda.set_device(local_rank)

    # ---- synthetic workload (identical on every rank) -------------------------------------------
    n_local = a.particles
    n_global = n_local * world
    pts, segs = pkg.synth.make_map_points(a.map_points, seed=1)
    tree = pkg.kd_create(pts)
    long_run = not a.no_cpu_baseline
    # long-run leg: filler up to the next frame % 100 == 6, 100 timed frames, 20 frames for the phase split
    n_frames = a.warmup + 3 * a.steps + (100 + 100 + 20 if long_run else 0)   # (+ 2 x steps: the frame-probe leg and the steady-state window; long run: filler up to the next frame % 100 == 6, then 100 + 20)

    def one_scan(f):
        return pkg.synth.make_scan(segs, (0.002 * f, 0.001 * f, 0.0004 * f), seed=2000 + f)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=host_cpu()[0]) as pool:  # numpy releases the GIL in the ray-casting ufuncs
        scans = list(pool.map(one_scan, range(n_frames)))

    cap = a.map_points + (1 << 18)

    fallback = {}   # native RCCL asked for and not available on every rank: {"native": False, "reason": ...}

    def make_engine():
        if distributed:
            from importlib import import_module
            sharded = import_module("gpu-icp-slam_amd.sharded")
            job_id = None
            if native and world > 1:   # rank 0 makes the job's RCCL id (two communicators: particle stream, chain stream); the process group carries it
                box = [pkg.mgpu_make_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                job_id = box[0]
            e, why = None, ""
            if native:
                try:
                    e = sharded.ShardedSlam(pkg, n_global, rank, world, device=local_rank, kd_capacity=cap, dist=dist, torch=None, native=True, native_id=job_id)
                except Exception as ex:   # ncclCommInitRank / the warm-up collectives failed on this rank
                    why = "%s: %s" % (type(ex).__name__, ex)
                if world > 1:             # ... then nobody uses them: the ranks agree over the process group, and the line says so
                    ok = torch.tensor([0 if e is None else 1], dtype=torch.int32)
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                    if int(ok.item()) == 0:
                        reasons = [None] * world
                        dist.all_gather_object(reasons, why)
                        fallback["reason"] = "; ".join("rank %d: %s" % (r, w) for r, w in enumerate(reasons) if w) or "a rank failed"
                        if e is not None:
                            e.close()
                        e = None
                elif e is None:
                    raise SystemExit("libpfslam_mgpu.so: " + why)
            if e is None:                 # torch.distributed issues the collectives (--collectives torch, or the fallback above)
                if native:
                    fallback["native"] = False
                    torch.cuda.set_device(local_rank)
                e = sharded.ShardedSlam(pkg, n_global, rank, world, device=local_rank, kd_capacity=cap, dist=dist, torch=torch, native=False, native_id=None)
        else:
            e = pkg.PfSlam(n_local, kd_capacity=cap, device=local_rank)
        e.set_map(tree)
        if a.variant:
            e.set_variant(a.variant)
        if a.topology:
            e.set_topology(a.topology)
        # initial condition: a particle cloud already dispersed around the start pose (5 dispersion steps), so that the
        # first scoring launches behave like steady state instead of scoring 100 k coincident particles
        for f in range(1, 6):
            e.motion_update(f)
        if a.cloud_sigma > 0 and not distributed:  # a wide start cloud (the organisation follows the cloud's spread)
            p = e.particles().copy()
            rs = np.random.RandomState(7)
            p["x"] = rs.normal(0, a.cloud_sigma, len(p)).astype(np.float32)
            p["y"] = rs.normal(0, a.cloud_sigma, len(p)).astype(np.float32)
            p["theta"] = rs.normal(0, a.cloud_sigma / 8.0, len(p)).astype(np.float32)
            e.set_particles(p)
        return e

    eng = make_engine()
    e0 = eng.eng if hasattr(eng, "eng") else eng

    def barrier():
        eng.synchronize()
        if getattr(eng, "native", None) is not None:
            eng.native.barrier_max()     # hipDeviceSynchronize + an all-reduce on the job's own RCCL communicator (a process-group barrier costs ~1 ms: 10 % of a 20-step window)
        elif dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(k0, count, frame0):
        barrier()
        t0 = time.perf_counter()
        for k in range(count):
            eng.step(frame0 + k, scans[k0 + k])
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if cuda_tensors else "cpu")
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)                       # every rank's own wall time: a SCALE record explains itself
            per_rank_s.append([float(v.item()) for v in every])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    per_rank_s = []

    def time_collectives(reps=20):
        """The frame's collectives on their own, same buffers and sizes as the frame issues them, `reps` each, bracketed by events on
        the stream the frame runs on (ms per call, this rank).  One rank: the sizes only -- nothing moves."""
        sizes = {"pose_blocks_bytes_per_rank": 3 * eng.stride * 4 if hasattr(eng, "stride") else 3 * n_local * 4,
                 "records_bytes_per_rank": 16, "weights_bytes_per_rank": (eng.stride if hasattr(eng, "stride") else n_local) * 4,
                 "balance_broadcast_bytes": 28 * int(e0.kd_size) + 16}
        res = {"sizes": sizes, "world": world, "ms": None, "issued_by": "libpfslam_mgpu.so (ncclAllGather into the frame's own streams)" if getattr(eng, "native", None) is not None else "torch.distributed",
               "note": "pose blocks: on the particle stream, under the scan-match kernel; the 16-byte records (a shard's packed keys): on the chain "
                       "stream between the reduce and the walls -- the one collective on the frame's critical chain; weights: on the particle stream; "
                       "the broadcast happens once per KDTree::Balance (100 frames).  ms = each collective on its own, back to back, per call"}
        if dist is None or world == 1:
            return res
        if getattr(eng, "native", None) is not None:
            ms = eng.native.time_collectives(reps)
        else:
            b = eng.buf
            local, glob = b.pose_blocks()
            ops = {"pose_blocks": lambda: dist.all_gather_into_tensor(glob, local),
                   "records": lambda: dist.all_gather_into_tensor(b.packs, b.pack),
                   "weights": lambda: dist.all_gather_into_tensor(b.gw, b.w)}
            ms = {}
            for name, op in ops.items():
                op(); torch.cuda.synchronize(); dist.barrier()
                e_a, e_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e_a.record()
                for _ in range(reps):
                    op()
                e_b.record(); torch.cuda.synchronize()
                ms[name] = e_a.elapsed_time(e_b) / reps
        t = torch.tensor([ms[k] for k in sorted(ms)], dtype=torch.float64, device="cuda" if cuda_tensors else "cpu")
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        res["ms"] = {k: [float(v[i].item()) for v in every] for i, k in enumerate(sorted(ms))}
        return res

    frame = FIRST_FRAME
    for k in range(a.warmup):
        eng.step(frame, scans[k]); frame += 1
    barrier()
    eng.set_timing(1)
    dt = timed(a.warmup, a.steps, frame)
    frame += a.steps
    timers = eng.timers()
    eng.set_timing(0)
    trace = eng.trace()
    pose_timed = np.array(eng.pose, np.float32)
    # how the timed frames ran: the round-5 frame (four in-order streams), its cross-stream edges as device-word gates or as events (the
    # start-up self-test decides; a second handle in the process withdraws the gates), one-stream mode
    fm = e0.frame_mode()
    frame_mode_timed = {"round5_frame": fm["round5_frame"], "edges": "gates" if fm["gates"] else "events", "one_stream": fm["serial"],
                        "sharded_frame": bool(distributed), "collectives": ("rccl, launched into the frame's own streams" if getattr(eng, "native", None) is not None else "torch.distributed") if world > 1 else None,
                        **({"collectives_fallback": fallback.get("reason", "")} if fallback else {})}
    topo_info = None
    if a.topology:   # replicated state: every rank holds the same graph and proposes the same pairs
        nodes_t, idx_t = eng.topology()
        topo_info = {"mode": a.topology, "nodes": int(len(nodes_t)), "node": int(idx_t), "closures_last_frame": int(len(eng.closures()))}

    # ---- where the frame's time goes: the next `steps` frames on the SAME handle with the frame probe on (pfslam_set_probe: the first thread
    # of every launch of a round-5 frame stores the wall clock; no events, no profiler) -- outside the timed region
    frame_probe, probe_frames = None, 0
    if not distributed:
        try:
            e0.set_probe(a.steps + 8)
            for k in range(a.steps):
                eng.step(frame, scans[a.warmup + a.steps + k]); frame += 1; probe_frames += 1
            eng.synchronize()
            names, tp, _ = e0.probe(a.steps)
            e0.set_probe(0)
            if len(tp) and "C scan-match" in names:
                sc, rd = names.index("C scan-match"), names.index("C reduce")
                ok = (tp[:, sc] > 0) & (tp[:, rd] > 0)
                good = ok[1:] & ok[:-1]
                if good.any():
                    chain = (tp[1:, sc] - tp[:-1, rd])[good]
                    rel = np.where(tp > 0, tp - tp[:, sc:sc + 1], np.nan)
                    frame_probe = {"frames": int(good.sum()), "chain_us_mean": float(chain.mean()), "chain_us_min": float(chain.min()), "chain_us_max": float(chain.max()),
                                   "scan_match_start_to_next_us": float((tp[1:, sc] - tp[:-1, sc])[good].mean()),
                                   "launch_start_us_after_scan_match_start": {names[k]: float(np.nanmean(rel[:, k])) for k in range(len(names)) if not np.all(np.isnan(rel[:, k]))},
                                   "note": "chain = start of a frame's reduce -> start of the next frame's scan-match kernel: what sits between two scan-match "
                                           "kernels (reduce, walls + insert, cell rows, the edge to the particle chain); wall-clock stamps of the launches' first "
                                           "threads over the %d frames BEHIND the timed window, same handle" % a.steps}
        except Exception as e:  # the probe must never break the contract line
            frame_probe = {"error": repr(e)}
    # ---- steady state: the same K steps once more, further into the run (every rank takes part).  The contract window above starts right
    # behind the warm-up, while the synthetic start cloud is still settling (the run's most expensive stretch); this one is the rate the
    # loop runs at from then on.  Reported beside `value`, never in its place.
    steady = None
    k_steady = a.warmup + a.steps + probe_frames
    if k_steady + a.steps <= n_frames:
        dts = timed(k_steady, a.steps, frame)
        steady = {"value": n_global * a.steps / dts, "ms_per_step": dts / a.steps * 1e3, "frames": "%d..%d" % (frame, frame + a.steps - 1), "unit": "particle-scan evals/s"}
        frame += a.steps
        probe_frames += a.steps   # (the long-run leg goes on behind these frames)

    # ---- census replay: what exactly did the timed launches issue?  A second handle steps through the same frames -- the frame
    # loop is deterministic, so its particles, scans and map are the timed run's, bit for bit (checked on the pose) -- with the
    # counting instantiation of the scan-match kernel behind every scoring pass (pfslam_set_census).  Every rank takes part.
    rep = make_engine()
    r0 = rep.eng if hasattr(rep, "eng") else rep
    r0.set_census(True)
    for k in range(a.warmup + a.steps):
        rep.step(FIRST_FRAME + k, scans[k])
    rep.synchronize()
    census_all = r0.census_log()
    replay_identical = bool((np.array(rep.pose, np.float32).view(np.int32) == pose_timed.view(np.int32)).all())
    r0.set_census(False)
    r0.close()
    del rep, r0

    coll = time_collectives() if (world > 1 or a.dry_collectives) else None   # every rank takes part
    out = None
    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = n_global * a.steps / dt
        out = {
            "metric": "particle-scan evals/sec (1081 beams x N particles), full particleFilter step, KD path",
            "value": value, "value_long_run": None, "unit": "particle-scan evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic 1081-beam scans, %d particles/GPU, %d-point KD map, full SLAM step "
                                   "(disperse+score+weights+ICP/SVD+map update+resample)" % (n_local, a.map_points),
                       "particles_global": n_global, "parallelism": "particles sharded x%d, map replicated" % world,
                       "frames": "%d..%d (KDTree::Balance falls on frame %% 100 == 5: not inside this window, see long_run)" % (frame - a.steps, frame - 1),
                       "kd_size_end": trace.get("kd_size")},
        }
        if topo_info is not None:
            out["config"]["topology"] = topo_info
        out["config"]["frame_mode"] = frame_mode_timed
        if steady is not None:
            out["steady_state"] = steady
        if frame_probe is not None:
            out["frame"] = frame_probe
        if per_rank_s:
            out["per_rank_ms_per_step"] = [v / a.steps * 1e3 for v in per_rank_s[0]]   # each rank's own wall clock over the timed window
        if coll is not None:
            out["collectives"] = coll
        if a.cloud_sigma > 0:
            out["config"]["cloud_sigma_m"] = a.cloud_sigma
        import oracle_lib as O
        # ---- roofline of the dominant kernel (rank 0's launches) -------------------------------------
        tree_end = np.ascontiguousarray(e0.map(), dtype=O.NODE_DTYPE)
        p_end = np.ascontiguousarray(e0.particles(), dtype=O.PARTICLE_DTYPE)
        last_scan = scans[a.warmup + a.steps - 1]
        if a.no_cpu_baseline or world > 1:   # (the CPU leg: rank 0 at N = 1 only -- the other ranks would wait for it)
            _, visits, valid = O.score_kd(tree_end, p_end[:128], last_scan, stats=True)
            vbar, bvalid = visits / max(valid, 1), valid / 128
        else:
            cb, vbar, bvalid = cpu_baseline(O, tree_end, p_end, last_scan, a.cpu_sample, len(tree_end))
            out["cpu_baseline"] = cb
        launches = max(timers["score_launches"], 1)
        kern_s = max(timers["score_ms"] / launches, 1e-9) * 1e-3
        plan_ms = timers["plan_ms"] / max(timers["plan_count"], 1)
        plan_stats = e0.plan_stats()
        cell_stats = e0.cell_stats()
        ub = e0.ubench_gather()
        # wave-level gathers per launch, from the replay's census of the SAME launches.  A descent-loop trip is one 16-byte wave
        # gather; a parent-hyperplane test one 16-byte + one 4-byte wave gather.  Bytes are lane-level: 64 lanes x record size.
        cen_timed = census_all[a.warmup:a.warmup + a.steps]

        cells_kernel = bool(cell_stats["rows"])

        def per_launch(c):
            # wave-level gathers: 16-byte = descent-loop trips / row slots + the node record of a parent-hyperplane test; 4-byte = the
            # parent index of such a test + (cell-row kernel, whose census books them in "uniform_trips") the cell-table words
            g16, g4 = c["trips"] + c["tests"], c["tests"] + (c["uniform_trips"] if cells_kernel else 0)
            return g16, g4, g16 * 64 * 16 + g4 * 64 * 4

        pl = [per_launch(c) for c in cen_timed] or [(0, 0, 0)]
        g16 = float(np.mean([v[0] for v in pl])); g4 = float(np.mean([v[1] for v in pl]))
        gbytes = [v[2] for v in pl]
        gather_bytes = float(np.mean(gbytes))                  # lane-level bytes the TA path moves per launch
        lane_visits = float(np.mean([c["visits"] for c in cen_timed])) if cen_timed else 0.0
        trips = float(np.mean([c["trips"] for c in cen_timed])) if cen_timed else 0.0
        # every wave gather priced as ONE request of the gather path (16 B x 64 lanes): a wave gather costs the texture addresser the same ~16
        # cycles whatever its width (profiles/r01_ubench_gather_rate.txt), and this is what a reader recomputes from the hardware counter
        # alone (TA_BUFFER_READ_WAVEFRONTS_sum x 1024 B / kernel time / peak).  The lane-level bytes (4-byte table words as 4 bytes) are kept
        # beside it as frac_lane_bytes.
        achieved_lane = gather_bytes / kern_s / 1e9
        achieved = (g16 + g4) * 1024.0 / kern_s / 1e9
        peak_measured = ub["wave_gathers_per_s"] * 1024.0 / 1e9
        pmc_path = find_pmc(a.pmc_file, n_local, a.map_points)
        traffic, clock_ghz, pmc, avgp = None, None, None, {}
        if pmc_path:
            try:
                pmc = json.load(open(pmc_path))
                avgp = pmc.get("avg_per_launch", {})
                traffic = pmc.get("hbm_bytes_per_launch")
                clock_ghz = pmc.get("measured_clock_ghz")
                gui, kms = avgp.get("GRBM_GUI_ACTIVE"), pmc.get("kernel_ms")
                if not clock_ghz and gui and kms:                # summaries of rounds 1-2
                    clock_ghz = gui / 8.0 / (kms * 1e-3) / 1e9   # GRBM_GUI_ACTIVE sums the 8 XCDs
            except Exception:
                traffic, pmc, avgp = None, None, {}
        ghz = clock_ghz or ub["nominal_ghz"]
        peak_nominal = ub["cus"] * 64.0 * ghz                  # 64 B/clk per CU (4 lanes x 16 B), GB/s
        alg_bytes_per_eval = bvalid * vbar * 32.0 + 20.0     # B_valid x V x 32 B + 16 B in + 4 B out
        ta_wf = avgp.get("TA_BUFFER_READ_WAVEFRONTS_sum")
        out["roofline"] = {
            "bound": "l1_gather", "achieved": achieved, "peak": peak_measured, "unit": "GB/s", "frac": achieved / peak_measured,
            "traffic": traffic,
            "traffic_source": ("replayed from " + os.path.relpath(pmc_path, ROOT) + " (rocprofv3 PMC passes of this command on another lease; not measured in this run)") if pmc_path else None,
            "frac_of_nominal_peak": achieved / peak_nominal,
            "frac_lane_bytes": achieved_lane / peak_measured, "achieved_lane_bytes": achieved_lane,
            "frac_all_gathers_as_16B": achieved / peak_measured,  # (= frac since round 5; the key of rounds 3-4 is kept)
            "peak_nominal": peak_nominal, "peak_nominal_clock_ghz": ghz,
            "peak_nominal_clock_source": ("GRBM_GUI_ACTIVE / 8 / kernel time of " + os.path.relpath(pmc_path, ROOT)) if clock_ghz else "nominal clock (hipDeviceProp)",
            "kernel": "k_score_kd_cells" if cell_stats["rows"] else "k_score_kd_plan" if plan_stats["rows"] else "k_score_kd",
            "kernel_ms": kern_s * 1e3, "launches": launches,
            "kernel_evals_per_s": n_local / kern_s,
            "census": {"launches": len(cen_timed), "replay_identical": replay_identical,
                       "gather_bytes_per_launch": {"min": float(min(gbytes)), "mean": gather_bytes, "max": float(max(gbytes))},
                       "wave_gathers_16B_per_launch": g16, "wave_gathers_4B_per_launch": g4,
                       "wave_gathers_per_launch": g16 + g4,
                       "lane_visits_per_launch": lane_visits, "lanes_active_per_trip": lane_visits / max(trips, 1.0),
                       "first": cen_timed[0] if cen_timed else None, "last": cen_timed[-1] if cen_timed else None,
                       "oracle_min_wave_gathers": n_local / 64.0 * bvalid * vbar,
                       "note": "counted on the timed launches themselves: a second handle replays frames %d..%d (bit-identical state, "
                               "`replay_identical`) with the counting instantiation of the kernel behind every scoring pass; "
                               "wave_gathers_per_launch = 16-byte + 4-byte wave gathers = what TA_BUFFER_READ_WAVEFRONTS_sum counts for "
                               "the same launches (pmc.census_over_pmc_wavefronts).  Cell-row kernel: trips = row slots (requested four "
                               "at a time) + node records of the generic tail, uniform_trips = cell-table words, prefix_trips = queries "
                               "that found their cell's row, redescents = queries that went on generically"
                               % (FIRST_FRAME, FIRST_FRAME + a.warmup + a.steps - 1)},
            "cells": dict(cell_stats, kernel_ms=plan_ms,
                          per_update={k: cell_stats[k] / max(cell_stats["updates"], 1.0) for k in ("walked_from_root", "extended", "reused", "claimed")},
                          note="persistent lattice-cell rows (pfslam_cell_stats, csrc/kd_cells.hip.inc) at the end of the timed window: cells "
                               "claimed since the last wipe (one 384-byte record each), live rows (one per sub-cell), and since the wipe: cells "
                               "walked from the root (once each), extensions (a link of the cell had gained a node in a frame's insert: a few "
                               "hops, rows re-cut), looks that found a cell unchanged = reused, per_update = the same per k_cells_update.  "
                               "kernel_ms = what this stream spends between the pose boxes and the scan-match kernel (waiting for the "
                               "previous frame's map update + k_cells_update; marking and the walks of new cells run on streams of their own)") if cell_stats["rows"] else None,
            "plan": None if cell_stats["rows"] else dict(plan_stats, kernel_ms=plan_ms,
                         note="shared-prefix plan of the LAST timed launch (pfslam_plan_stats): one planning lane per (wave, beam) walks "
                              "the root path common to the wave's 64 queries and keeps only the nodes that can be nearest for some "
                              "lane; kernel_ms = k_group_box + k_plan, which run before the scan-match kernel"),
            "definition_version": 2,   # 1 (rounds 1-4): frac / achieved priced 4-byte gathers at 256 B -- that figure is frac_lane_bytes / achieved_lane_bytes now; do not compare `frac` across versions
            "definition": "achieved = wave gathers one timed launch issues (census of the timed launches, mean) x 1024 B (every wave gather "
                          "as one 16 B x 64 lanes request of the gather path, whatever its width) / HIP-event time of the same launches "
                          "(mean); frac_lane_bytes prices the 4-byte gathers (cell-table words, parent indices) at 256 B instead; peak = wave-gather rate of this chip measured in this process (pfslam_ubench_gather: cache-resident "
                          "table, 8 waves/SIMD) x 1024 B; frac_of_nominal_peak prices the same bytes against CUs x 64 B/clk x clock.  "
                          "The cell-row kernel of round 3 replaced the per-query tree walk by a handful of row slots per query, and what "
                          "binds it now is the vector ALUs and the gather path together (pmc.valu_issue_busy_mix_weighted, pmc.ta_busy): `frac` says how much of the gather path it still uses, "
                          "not how far it is from its own ceiling",
            "ubench": dict(ub, note="the measured rate is what the same gather instruction sustains on this box, wave-uniform addresses"),
            "pmc": None if not pmc else {
                "source": os.path.relpath(pmc_path, ROOT), "replayed": True,
                "replayed_note": "every field of this block is computed from the committed rocprofv3 summary named in `source` (PMC passes of `python bench.py "
                                 "--no-cpu-baseline` on another lease, tools/profile_round.sh), NOT measured in this run; measured live in this run: ms_per_step, "
                                 "kernel_ms, the census, ubench, frame",
                "measured_clock_ghz": clock_ghz,
                "valu_issue_busy_mix_weighted": valu_mix_busy(pmc, avgp, ub["cus"], clock_ghz),
                "ta_buffer_read_wavefronts_per_launch": ta_wf,
                "census_over_pmc_wavefronts": ((g16 + g4) / ta_wf) if ta_wf else None,
                "ta_busy": (avgp["TA_TA_BUSY_sum"] / ub["cus"] / (avgp["GRBM_GUI_ACTIVE"] / 8.0)) if avgp.get("TA_TA_BUSY_sum") and avgp.get("GRBM_GUI_ACTIVE") else None,
                "ta_cycles_frac": (avgp["TA_BUFFER_TOTAL_CYCLES_sum"] / ub["cus"] / (avgp["GRBM_GUI_ACTIVE"] / 8.0)) if avgp.get("TA_BUFFER_TOTAL_CYCLES_sum") and avgp.get("GRBM_GUI_ACTIVE") else None,
                "valu_insts_per_simd_cycle": ((avgp.get("SQ_INSTS_VALU", 0.0) / (4.0 * ub["cus"])) / (avgp["GRBM_GUI_ACTIVE"] / 8.0)) if avgp.get("GRBM_GUI_ACTIVE") else None,
                "valu_issue_frac_if_4_cycles_each": (4.0 * (avgp.get("SQ_INSTS_VALU", 0.0) / (4.0 * ub["cus"])) / (avgp["GRBM_GUI_ACTIVE"] / 8.0)) if avgp.get("GRBM_GUI_ACTIVE") else None,
                # MEASURED busy cycles (round 4): SQ_ACTIVE_INST_VALU counts quad-cycles a SIMD spends executing vector ALU instructions
                # (rocprofiler's VALUBusy = 4 x SQ_ACTIVE_INST_VALU / SIMDs / cycles); cycles = that pass's own kernel time x the measured clock
                "sq_active_inst_valu_over_insts_valu": (avgp["SQ_ACTIVE_INST_VALU"] / avgp["SQ_INSTS_VALU"]) if avgp.get("SQ_ACTIVE_INST_VALU") and avgp.get("SQ_INSTS_VALU") else None,
                "vmem_issue_busy_measured": pmc_busy(pmc, avgp, "SQ_ACTIVE_INST_VMEM", 4.0 * ub["cus"], clock_ghz, quad=True),
                "scalar_busy_measured": pmc_busy(pmc, avgp, "SQ_ACTIVE_INST_SCA", 4.0 * ub["cus"], clock_ghz, quad=True),
                "sq_busy_measured": pmc_busy(pmc, avgp, "SQ_BUSY_CYCLES", 32.0, clock_ghz, quad=False),
                "wave_cycles_waiting": (avgp["SQ_WAIT_ANY"] / avgp["SQ_WAVE_CYCLES"]) if avgp.get("SQ_WAIT_ANY") and avgp.get("SQ_WAVE_CYCLES") else None,
                "valu_insts_by_type": {k[len("SQ_INSTS_VALU_"):].lower(): avgp[k] for k in sorted(avgp) if k.startswith("SQ_INSTS_VALU_")} or None,
                # the line's gather fraction again, from profiles/ alone: every wave gather the counter saw as a 16-byte one, over the
                # kernel time of the same file, against the gather rate measured live in this process
                "frac_from_pmc_only": (ta_wf * 1024.0 / (pmc.get("kernel_ms") * 1e-3) / 1e9 / peak_measured) if ta_wf and pmc.get("kernel_ms") else None,
                "hbm": None if traffic is None else {"bytes_per_launch": traffic, "achieved": traffic / kern_s / 1e9, "peak": HBM_PEAK_GBS,
                                                     "unit": "GB/s", "frac": traffic / kern_s / 1e9 / HBM_PEAK_GBS},
                "note": "rocprofv3 PMC passes of `python bench.py --no-cpu-baseline` (tools/profile_round.sh), averaged over the TIMED "
                        "launches of that process (dispatches [warmup, warmup + steps) of the timed instantiation).  ta_busy = TA_TA_BUSY_sum / CUs / (GRBM_GUI_ACTIVE / 8 XCDs); "
                        "ta_cycles_frac = TA_BUFFER_TOTAL_CYCLES_sum / CUs / kernel cycles; hbm = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950 "
                        "correction of MI355X_MICROARCH.md) -- the map records are cache resident, compulsory HBM traffic is ~20 B per "
                        "evaluation; census_over_pmc_wavefronts compares this run's census with the counter (1.0 = agreement; the census "
                        "also counts the parent-index reads of the generic tail, global loads a BUFFER counter does not see); "
                        "SQ_ACTIVE_INST_VALU ticks ONCE per VALU instruction on this part (sq_active_inst_valu_over_insts_valu = 1.00; the same in "
                        "every kind of tools/ubench/valu_rate, profiles/r05_valu_calibration.json), so '4 x the counter' is an instruction count "
                        "priced at 4 cycles, not a busy measurement (it gives 1.68 for a pure v_fma_f32 stream); valu_issue_busy_mix_weighted = the "
                        "instruction counts by kind (SQ_INSTS_VALU_*) x the cycles per wave-instruction per SIMD MEASURED for each kind by that "
                        "calibration (fp32 add / mul / fma 2.4, everything else 4.2; the INT32 class holds both kinds: lo / hi) / SIMDs / the kernel's "
                        "cycles; valu_issue_frac_if_4_cycles_each is the round-3 estimate kept beside it; sq_busy_measured = "
                        "SQ_BUSY_CYCLES / 32 shader engines x XCDs / cycles"},
            "alg_equiv": {"bytes_per_eval": alg_bytes_per_eval, "mean_node_visits": vbar, "valid_beams": bvalid,
                          "GBs": alg_bytes_per_eval * n_local / kern_s / 1e9,
                          "note": "SURVEY 8d's algorithmic node bytes (B_valid x V x 32 B + 20 B): served from L1/L2, "
                                  "not an HBM statement and not a fraction of any peak"},
        }

    # ---- the same step over a whole balance cycle (every rank takes part; rank 0 reports) ----------
    if dist is not None:
        dist.barrier()   # rank 0 has just put its report together: the others enqueue nothing while it does (a frame's stream gates wait behind the collectives)
    if long_run:
        # continue to the next frame % 100 == 6, then time exactly 100 frames: one KDTree::Balance (frame % 100 == 5) inside
        k = a.warmup + a.steps + probe_frames
        while frame % 100 != 6:
            eng.step(frame, scans[k]); frame += 1; k += 1
        if k + 100 <= n_frames:
            dtl = timed(k, 100, frame)
            frame += 100; k += 100
            if out is not None:
                out["value_long_run"] = n_global * 100 / dtl
                out["long_run"] = {"frames": 100, "first_frame": frame - 100, "ms_per_step": dtl / 100 * 1e3,
                                   "value": n_global * 100 / dtl, "unit": "particle-scan evals/s",
                                   "includes": "one KDTree::Balance (host re-build + upload of the whole map, kernel.cu:1707-1711) "
                                               "and the map growing between two balances", "kd_size_end": eng.trace().get("kd_size")}
        if not distributed and k + 20 <= n_frames:  # per-phase split, as the reference prints it (kernel.cu:1754-1759)
            eng.set_timing(2)
            for j in range(20):
                eng.step(frame, scans[k + j]); frame += 1
            eng.synchronize()
            t = eng.timers()
            eng.set_timing(0)
            if out is not None:
                out["phases_ms"] = {name: t[name + "_ms"] / max(t[name + "_count"], 1) for name in ("motion", "measurement", "map", "resample")}
                out["phases_ms"]["note"] = ("HIP events on the step's stream, mean of 20 frames: motion = dispersion; measurement = lane "
                                            "order + scan-match + reduce/min/max + weights + Neff (ICP runs under it); map = device chain incl. the "
                                            "insert of the new walls, kept on the main stream while the phases are timed (it runs on the aux "
                                            "stream otherwise); resample = the five gated launches, averaged over all frames")
    if out is not None and hasattr(eng, "balance_builds"):
        out["balance"] = {"host_builds_on_rank0": eng.balance_builds, "broadcasts": eng.balance_broadcasts,
                          "note": "KDTree::Balance once per node: rank 0 re-builds the map on the host, the device arrays (28 B per node) "
                                  "are broadcast, the other ranks adopt them (include/pfslam.h, pfslam_shard_balance_*)"}
    if out is not None:
        if world == 1 and not a.no_cpu_baseline:
            import oracle_lib as O
            out["extras"] = extras(pkg, O, tree, pts, scans[0], local_rank)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
