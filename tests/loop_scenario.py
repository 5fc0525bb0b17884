"""A closed synthetic loop for the end-to-end map / loop-closure comparison (BASELINE configs[4], SURVEY 8f #2 + #4).

The robot drives a 6 x 6 m square (24 m, 0.1 m per frame) and overlaps its start by 2 m: 260 frames.  The reference's filter
has no motion model besides a 1.5 cm diffusion per frame (kernel.cu:375-397) and cannot follow 0.1 m steps by itself, so --
exactly as an odometry prior would -- the particle cloud is re-centred on the simulated pose before every frame through the
stage API both sides have (set_particles).  Everything else is the frame loop: particleFilter with UpdateTopology +
CheckLoopClosure enabled where the reference has them commented out (kernel.cu:1750-1751).

`run(engine, ...)` drives anything with the PfSlam / oracle Slam interface and returns the per-frame record and the final
exports; the same function produces the golden fixture (oracle, CPU) and checks the product (GPU)."""
import numpy as np

N_FRAMES = 260
STEP = 0.1
SIDE = 6.0
N_PARTICLES = 48


def trajectory(n_frames=N_FRAMES):
    """Poses (x, y, theta) along the square, starting at (-3, -3) heading +x; theta follows the side."""
    out = []
    for f in range(n_frames):
        s = (f * STEP) % (4 * SIDE)
        side, d = int(s // SIDE), s % SIDE
        if side == 0: x, y, th = -3.0 + d, -3.0, 0.0
        elif side == 1: x, y, th = 3.0, -3.0 + d, np.pi / 2
        elif side == 2: x, y, th = 3.0 - d, 3.0, np.pi
        else: x, y, th = -3.0, 3.0 - d, -np.pi / 2
        out.append(np.array([x, y, th], np.float32))
    return out


def world(pkg, seed=7, n_points=6000):
    _, segs = pkg.synth.make_segments(n_points, seed, half_extent=12.0, clear_radius=0.0)
    # keep the driven square itself free of walls: drop segments that come within 0.4 m of the path
    keep = []
    for x0, y0, x1, y1 in segs:
        xs, ys = np.linspace(x0, x1, 40), np.linspace(y0, y1, 40)
        d = np.minimum.reduce([np.hypot(np.clip(xs, -3, 3) - xs, ys + 3), np.hypot(np.clip(xs, -3, 3) - xs, ys - 3),
                               np.hypot(xs + 3, np.clip(ys, -3, 3) - ys), np.hypot(xs - 3, np.clip(ys, -3, 3) - ys)])
        if d.min() > 0.4:
            keep.append((x0, y0, x1, y1))
    return np.asarray(keep, np.float64)


def scans(pkg, n_frames=N_FRAMES, seed=7):
    segs = world(pkg, seed)
    return [pkg.synth.make_scan(segs, tuple(float(v) for v in p), seed=31000 + f) for f, p in enumerate(trajectory(n_frames))]


def run(engine, make_particles, frame_scans, grid_path=False, n_frames=N_FRAMES):
    """Returns (records, topology nodes, node index): records[f] = (pose bits x3, kd size, resampled, closure pairs as a tuple)."""
    engine.set_topology(True)
    traj = trajectory(n_frames)
    rec = []
    for f in range(1, n_frames + 1):
        p = traj[f - 1]
        engine.set_particles(make_particles(p))   # the odometry prior: cloud re-centred on the simulated pose
        if grid_path:
            engine.step_grid(f, frame_scans[f - 1])
        else:
            engine.step(f, frame_scans[f - 1])
        t = engine.trace()
        pose = np.asarray(engine.pose, np.float32)
        pairs = engine.closures()
        rec.append((tuple(pose.view(np.int32).tolist()), int(t.get("kd_size", 0)), int(t["resampled"]), tuple(map(tuple, pairs.tolist()))))
    return rec
