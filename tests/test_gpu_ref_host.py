"""The reference's HOST half as a checker (round 6): oracle/_ref/libkernel_ref_host.so is /root/reference/src/kernel.cu compiled whole --
particleFilter and everything under it, host functions and kernels -- with its kdtree.cpp / utilities.cpp / scene.cpp
(oracle/kernel_ref_host_wrap.cpp has the recipe, the ONE host function it has to add -- CUDA's host-side rsqrt, which svd3.h needs -- and
what that means for the claim).  The reference's own frame loop runs here on the MI355X, PARTICLE_COUNT = 1000 particles, and the product
is stepped beside it frame by frame FROM THE REFERENCE'S STATE (its host particle array, robotPos and tree are pushed into the product in
front of every frame), so that every frame is compared on identical inputs and the reference's own run-to-run effects (H3: the in-place
resample races; H4: non-atomic map weights; thrust::reduce's summation order) cannot accumulate.

Pinned by this, beyond the kernels (tests/test_gpu_ref_kernels.py): the step order (kernel.cu:1702-1762), PFMotionUpdate's upload of the
host array (400-418), PFMeasurementUpdateKD's min / max / argmax, `(int)min` weights and half-array read-back (1311-1348), the ICP host
sequence incl. the host-side svd (993-1093), PFUpdateMapKD's double loop, list order, ROUND_FRAC snapping, H6 upload length and INSERT ORDER
(1406-1540: the tree's byte layout), PFResample's Neff / threshold / scan (447-511).
The product runs with the device library's transcendentals (pfslam_set_trig 1: what the reference's text compiles to here) and the
reference's H6 behaviour (free_upload_bug = 1)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
REF = os.path.join(O.ORACLE_DIR, "_ref")
NB = 1081

SCENE_TXT = """// Camera
CAMERA
RES         800 800
FOVY        45
FILE        map0
EYE         0.0 0.0 25
LOOKAT      0 0 0
UP          0 1 0

// Patch size in meters
MAP
SIZE \t40 40
RES\t.025
"""


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def report(line):
    print(line)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "ref_host_agreement.txt"), "a") as fh:
            fh.write(line + "\n")
    except OSError:
        pass


class RefHost:
    def __init__(self, tmp_path, scans):
        path = os.path.join(REF, "libkernel_ref_host.so")
        if not os.path.exists(path):
            pytest.skip("oracle/_ref/libkernel_ref_host.so not built (needs /root/reference and hipify-perl at build time)")
        self.L = C.CDLL(path)
        self.n = self.L.refhost_particle_count()
        scene = tmp_path / "scene.txt"
        scene.write_text(SCENE_TXT)  # same format as the reference's data/map_settings.txt
        f = tmp_path / "scans.f32"
        np.ascontiguousarray(scans, np.float32).tofile(str(f))
        self.frames = self.L.refhost_init(str(scene).encode(), str(f).encode())
        assert self.frames == len(scans)
        self.cap = 1 << 20

    def step(self, frame):
        assert self.L.refhost_step(frame) == 0

    def get(self):
        p = np.zeros(self.n, O.PARTICLE_DTYPE)
        pose = np.zeros(3, np.float32)
        nodes = np.zeros(self.cap, O.NODE_DTYPE)
        n = C.c_int(0)
        assert self.L.refhost_get(O.P(p), O.P(pose), O.P(nodes), self.cap, C.byref(n)) == 0
        assert n.value <= self.cap
        return p, pose, nodes[:n.value].copy()

    def close(self):
        self.L.refhost_free()


def test_reference_frame_loop_against_the_product(pkg, tmp_path):
    n_frames = 40
    segs, seq = pkg.synth.corridor_sequence(n_frames + 1, seed=5)
    scans = np.stack([s for _, s in seq]).astype(np.float32)   # the reference reads scans[frame] (kernel.cu:1716); frames start at 1 (H9)
    assert pkg.device_count() > 0   # (the product's library talks to the runtime first: loaded the other way round, its device query fails)
    h = pkg.PfSlam(1000, kd_capacity=1 << 18, free_upload_bug=1, strict_host_mirror=1)
    ref = RefHost(tmp_path, scans)
    assert ref.n == 1000            # PARTICLE_COUNT, kernel.cu:30
    h.set_trig(1)
    stats = {"frames": 0, "pose_bits": 0, "pose_max_err": 0.0, "tree_struct": 0, "tree_weights": 0, "particles_no_resample": 0, "no_resample_frames": 0,
             "resample_frames": 0, "resample_decision": 0, "stage_tree": 0, "stage_frames": 0, "weights_lost": 0}
    for f in range(1, n_frames + 1):
        p0, pose0, t0 = ref.get()               # the state the reference's frame f starts from ...
        if len(t0):
            h.set_map(t0)                       # ... pushed into the product
        h.set_particles(p0)
        h.set_pose(pose0)
        ref.step(f)
        h.step(f, scans[f])
        p1, pose1, t1 = ref.get()
        hp, hpose, ht, tr = h.particles().copy(), np.asarray(h.pose, np.float32), h.map().copy(), h.trace()
        stats["frames"] += 1
        # pose = best particle + ICP increment: the sums of the ICP are thrust::reduce's on the reference's side (order unspecified)
        stats["pose_bits"] += int((bits(hpose) == bits(pose1)).all())
        stats["pose_max_err"] = max(stats["pose_max_err"], float(np.abs(hpose - pose1).max()))
        assert np.abs(hpose - pose1).max() <= 1e-5, "frame %d: pose %s vs the reference's %s" % (f, hpose, pose1)
        # the tree: node count, positions and links (= the insert ORDER of PFUpdateMapKD's host loop), then the weights
        assert len(ht) == len(t1), "frame %d: %d nodes vs the reference's %d" % (f, len(ht), len(t1))
        same_struct = all((ht[k] == t1[k]).all() for k in ("axis", "left", "right", "parent")) and all((bits(ht[k]) == bits(t1[k])).all() for k in ("x", "y", "z"))
        stats["tree_struct"] += int(same_struct)
        wdiff = int((bits(ht["w"]) != bits(t1["w"])).sum())
        stats["tree_weights"] += int(wdiff == 0)
        stats["weights_lost"] += wdiff
        # did the frame resample?  (the reference's particles all carry w == 1 behind a resample: kernel.cu:441-442)
        ref_resampled = bool((p1["w"] == 1.0).all()) and f > 1
        stats["resample_decision"] += int(bool(tr["resampled"]) == ref_resampled or f == 1)
        if f > 1 and not ref_resampled:
            stats["no_resample_frames"] += 1
            ok = all((bits(hp[k]) == bits(p1[k])).all() for k in ("x", "y", "theta", "w"))
            stats["particles_no_resample"] += int(ok)
        elif f > 1:
            stats["resample_frames"] += 1
        # the map update on its own, AT THE REFERENCE'S POSE (takes the ICP sums' order out of the comparison): PFUpdateMapKD's lists, snapping,
        # H6 upload, weight passes, new-wall test and insert order must give the reference's tree, byte for byte in structure
        if len(t0):
            g = pkg.PfSlam(64, kd_capacity=1 << 18, free_upload_bug=1)
            g.set_trig(1)
            g.set_map(t0); g.set_scan(scans[f]); g.set_pose(pose1)
            g.update_map_kd()
            gt = g.map().copy()
            g.close()
            stats["stage_frames"] += 1
            ok = len(gt) == len(t1) and all((gt[k] == t1[k]).all() for k in ("axis", "left", "right", "parent")) and all((bits(gt[k]) == bits(t1[k])).all() for k in ("x", "y", "z"))
            stats["stage_tree"] += int(ok)
            assert ok, "frame %d: PFUpdateMapKD at the reference's pose gives another tree" % f
    h.close()
    ref.close()
    report("reference frame loop (kernel.cu whole, %d frames x 1000 particles, product stepped from the reference's state): %s" % (n_frames, stats))
    assert stats["stage_tree"] == stats["stage_frames"] > 0
    assert stats["tree_struct"] >= stats["frames"] - 2          # (a pose differing in its last place may move a cell's snapped coordinate)
    assert stats["resample_decision"] == stats["frames"]
    assert stats["particles_no_resample"] == stats["no_resample_frames"]
