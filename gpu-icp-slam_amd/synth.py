"""Seeded synthetic workloads: the reference's real inputs
(data/train_lidar*.mat) are absent from its checkout, so maps, scans and particle sets are
generated here -- identically for the CPU oracle, the CPU baseline and the GPU path.

  map   : random axis-aligned wall segments (0.5-8 m) in a 40x40 m area, snapped to the 0.025 m
          grid exactly as the reference snaps map points (ROUND_FRAC: k * 0.025f), de-duplicated,
          integer occupancy weights in [-100, 113], shuffled.
  scan  : 1081 ranges, -135..+135 deg in 0.25 deg steps, ray-cast against the same segments from a
          pose, clipped to [0.1, 30] m (so some beams exceed the +-20 m reject), + N(0, 0.01).
"""
import numpy as np

RES = np.float32(0.025)
N_BEAMS = 1081


def make_segments(n_points, seed=1, half_extent=19.0, clear_radius=1.5):
    """Return (cells[K,2] int32 grid coordinates, segments[S,4] float64 x0,y0,x1,y1)."""
    rng = np.random.RandomState(seed)
    half_cells = int(round(half_extent / float(RES)))
    seen = set()
    cells = []
    segs = []
    while len(cells) < n_points:
        ix0 = int(rng.randint(-half_cells, half_cells))
        iy0 = int(rng.randint(-half_cells, half_cells))
        length = int(round(rng.uniform(0.5, 8.0) / float(RES)))
        horiz = bool(rng.randint(0, 2))
        if horiz:
            xs = np.arange(ix0, min(ix0 + length, half_cells)); ys = np.full_like(xs, iy0)
        else:
            ys = np.arange(iy0, min(iy0 + length, half_cells)); xs = np.full_like(ys, ix0)
        r2 = (xs * float(RES)) ** 2 + (ys * float(RES)) ** 2
        keep = r2 > clear_radius ** 2
        xs, ys = xs[keep], ys[keep]
        if len(xs) == 0:
            continue
        # contiguous runs only (the clear disc may cut a segment in two): keep the longest run
        brk = np.where(np.diff(xs if horiz else ys) != 1)[0]
        if len(brk):
            runs = np.split(np.arange(len(xs)), brk + 1)
            run = max(runs, key=len)
            xs, ys = xs[run], ys[run]
        added = 0
        for cx, cy in zip(xs.tolist(), ys.tolist()):
            if (cx, cy) not in seen:
                seen.add((cx, cy)); cells.append((cx, cy)); added += 1
        if added:
            segs.append((xs[0] * float(RES), ys[0] * float(RES), xs[-1] * float(RES), ys[-1] * float(RES)))
    cells = np.asarray(cells[:n_points], dtype=np.int32)
    return cells, np.asarray(segs, dtype=np.float64)


def make_map_points(n_points, seed=1):
    """K x 4 float32 (x, y, 0, w) map points + the wall segments they came from."""
    cells, segs = make_segments(n_points, seed)
    rng = np.random.RandomState(seed + 1000)
    pts = np.zeros((len(cells), 4), np.float32)
    pts[:, 0] = cells[:, 0].astype(np.float32) * RES
    pts[:, 1] = cells[:, 1].astype(np.float32) * RES
    pts[:, 3] = rng.randint(-100, 114, size=len(cells)).astype(np.float32)
    rng.shuffle(pts)
    return pts, segs


def beam_angles(theta=0.0):
    j = np.arange(N_BEAMS, dtype=np.float64)
    return np.deg2rad(-135.0 + 0.25 * j) + float(theta)


def make_scan(segs, pose=(0.0, 0.0, 0.0), seed=2, noise=0.01, rmin=0.1, rmax=30.0):
    """Ray-cast 1081 beams from `pose` against axis-aligned segments (treated as thin walls)."""
    px, py, th = [float(v) for v in pose]
    ang = beam_angles(th)
    dx, dy = np.cos(ang)[:, None], np.sin(ang)[:, None]
    x0, y0, x1, y1 = [segs[:, k][None, :] for k in range(4)]
    horiz = (y0 == y1)
    with np.errstate(divide="ignore", invalid="ignore"):
        # horizontal wall y = y0, x in [x0, x1]
        t_h = (y0 - py) / dy
        xh = px + t_h * dx
        ok_h = horiz & (t_h > 0) & (xh >= np.minimum(x0, x1) - 0.0125) & (xh <= np.maximum(x0, x1) + 0.0125)
        # vertical wall x = x0, y in [y0, y1]
        t_v = (x0 - px) / dx
        yv = py + t_v * dy
        ok_v = (~horiz) & (t_v > 0) & (yv >= np.minimum(y0, y1) - 0.0125) & (yv <= np.maximum(y0, y1) + 0.0125)
    t = np.where(ok_h, t_h, np.inf)
    t = np.minimum(t, np.where(ok_v, t_v, np.inf))
    r = t.min(axis=1)
    r = np.where(np.isfinite(r), r, rmax)
    rng = np.random.RandomState(seed)
    r = r + rng.normal(0.0, noise, size=r.shape)
    return np.clip(r, rmin, rmax).astype(np.float32)


def make_weird_scan(seed=3):
    """A scan with edge cases: zero ranges, ranges beyond the 20 m reject, huge values."""
    rng = np.random.RandomState(seed)
    r = rng.uniform(0.0, 35.0, N_BEAMS).astype(np.float32)
    r[::97] = 0.0
    r[5::101] = 1000.0
    return r


def corridor_sequence(n_frames, seed=5, n_points=4000):
    """A short seeded drive: returns (segments, list of (pose, scan))."""
    _, segs = make_segments(n_points, seed)
    out = []
    for f in range(n_frames):
        pose = (0.02 * f, 0.01 * f, 0.004 * f)
        out.append((pose, make_scan(segs, pose, seed=seed * 100 + f)))
    return segs, out
