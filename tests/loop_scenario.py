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


# ---- free-running variant (golden_v4): no re-centring of the cloud -----------------------------------------------------------
# The robot starts at the origin of ITS OWN map frame (the first scan seeds the map at robotPos = 0, kernel.cu:1714-1717) and the
# harness hands the filter the commanded motion of every frame as an odometry increment -- a rigid shift of every pose the filter
# holds (pfslam_shift_particles; the reference's filter has no motion model besides its 1.5 cm diffusion).  The particle cloud is
# never replaced: diversity, weights and resampling history carry over from frame to frame, 100 000 particles.
N_PARTICLES_FREE = 100000


def trajectory_free(n_frames=N_FRAMES):
    """The square loop of `trajectory` in the robot's start frame: pose 0 = (0, 0, 0)."""
    t = trajectory(n_frames)
    return [np.array([p[0] - t[0][0], p[1] - t[0][1], p[2] - t[0][2]], np.float32) for p in t]


def run_free(engine, frame_scans, grid_path=False, n_frames=N_FRAMES, look_every=1, topology_mode=1):
    """Free-running closed loop: shift by the commanded motion, step.  records[f] as in `run`; the engine is only looked at every
    `look_every` frames (in between the frames stay enqueued -- with the topology calls inside the frame loop they are booked at
    once anyway; topology_mode 2 = the product's lagged booking, pfslam_set_topology)."""
    engine.set_topology(topology_mode)
    traj = trajectory_free(n_frames)
    rec = []
    for f in range(1, n_frames + 1):
        if f > 1:
            engine.shift_particles(traj[f - 1] - traj[f - 2])
        if grid_path:
            engine.step_grid(f, frame_scans[f - 1])
        else:
            engine.step(f, frame_scans[f - 1])
        if f % look_every and f != n_frames:
            continue
        t = engine.trace()
        pose = np.asarray(engine.pose, np.float32)
        pairs = engine.closures()
        rec.append((tuple(pose.view(np.int32).tolist()), int(t.get("kd_size", 0)), int(t["resampled"]), tuple(map(tuple, pairs.tolist()))))
    return rec
