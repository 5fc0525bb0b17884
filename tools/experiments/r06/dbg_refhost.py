import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("gpu-icp-slam_amd")
order = sys.argv[1] if len(sys.argv) > 1 else "ref_first"
print("devices before:", pkg.device_count(), flush=True)
if order == "prod_first":
    h = pkg.PfSlam(1000, kd_capacity=1 << 18)
    print("product handle ok", flush=True)
L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libkernel_ref_host.so"))
print("devices after load:", pkg.device_count(), flush=True)
import test_gpu_ref_host as T
open("/tmp/scene.txt", "w").write(T.SCENE_TXT)
segs, seq = pkg.synth.corridor_sequence(12, seed=5)
np.minimum(np.stack([s for _, s in seq]).astype(np.float32), np.float32(19.0)).tofile("/tmp/scans.f32")
n = L.refhost_init(b"/tmp/scene.txt", b"/tmp/scans.f32")
print("refhost_init ->", n, "devices after init:", pkg.device_count(), flush=True)
for f in range(1, 10):
    print("step", f, L.refhost_step(f), flush=True)
print("devices after step:", pkg.device_count(), flush=True)
h2 = pkg.PfSlam(1000, kd_capacity=1 << 18)
print("second product handle ok")
