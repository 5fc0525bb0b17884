"""The device-side map insert (k_test_new, DESIGN.md "frame pipeline") resolves a frame's new walls in parallel claim rounds
instead of the reference's sequential KDTree::InsertNode loop (kdtree.cpp:69-105, kernel.cu:1512-1517).  This is a plain
Python model of exactly those rounds -- every point claims the link it fell off with its list position, the smallest position
wins, the others continue below the winner -- checked against the oracle's sequential insert on random and adversarial
(sorted, duplicated, one-leaf) batches.  It pins the ALGORITHM on the CPU; the kernel itself is pinned by the -m gpu replays."""
import numpy as np

import oracle_lib as O


def insert_in_rounds(tree, pts):
    """tree: NODE_DTYPE array (modified copy returned), pts: (m, 4) float32 in list order."""
    old, m = len(tree), len(pts)
    out = np.concatenate([tree, np.zeros(m, tree.dtype)])
    key = lambda i, axis: (out["x"][i], out["y"][i], out["z"][i])[axis]
    cur = np.zeros(m, np.int64)            # node every pending point currently stands on (0 = root)
    placed = np.zeros(m, bool)
    rounds = 0
    while not placed.all():
        rounds += 1
        claims = {}
        for k in np.flatnonzero(~placed):  # descend until a link is empty, claim it with the list position
            at = int(cur[k])
            while True:
                axis = int(out["axis"][at])
                side = "left" if pts[k][axis] < key(at, axis) else "right"
                nxt = int(out[side][at])
                if nxt < 0:
                    break
                at = nxt
            cur[k] = at
            claims[(at, side)] = min(claims.get((at, side), m), k)
        for k in np.flatnonzero(~placed):  # resolve: the smallest position hangs there, the others go on below it
            at = int(cur[k])
            axis = int(out["axis"][at])
            side = "left" if pts[k][axis] < key(at, axis) else "right"
            w = claims[(at, side)]
            if w == k:
                out[side][at] = old + k
                out[old + k] = ((axis + 1) % 3, -1, -1, at, pts[k][0], pts[k][1], pts[k][2], pts[k][3])
                placed[k] = True
            else:
                cur[k] = old + w
    return out, rounds


def sequential(tree, pts):
    out = np.concatenate([tree, np.zeros(len(pts), tree.dtype)])
    for k, p in enumerate(pts):
        O.kd_insert(out, len(tree) + k, p)
    return out


def batch(rng, m, kind):
    p = np.zeros((m, 4), np.float32)
    if kind == "random":
        p[:, 0] = rng.randint(-400, 400, m) * np.float32(0.025)
        p[:, 1] = rng.randint(-400, 400, m) * np.float32(0.025)
    elif kind == "sorted_wall":      # cells of one wall in index order: one long chain
        p[:, 0] = np.float32(7.5)
        p[:, 1] = np.arange(m) * np.float32(0.025) - np.float32(3.0)
    elif kind == "duplicates":
        p[:, 0] = rng.randint(-3, 3, m) * np.float32(0.025)
        p[:, 1] = rng.randint(-3, 3, m) * np.float32(0.025)
    else:                             # "one_leaf": everything lands below the same leaf
        p[:, 0] = np.float32(19.0) + rng.randint(0, 40, m) * np.float32(0.025)
        p[:, 1] = np.float32(19.0) + rng.randint(0, 40, m) * np.float32(0.025)
    p[:, 3] = -100.0
    return p


def test_claim_rounds_equal_the_sequential_insert():
    rng = np.random.RandomState(5)
    base = np.zeros((300, 4), np.float32)
    base[:, 0] = rng.randint(-200, 200, 300) * np.float32(0.025)
    base[:, 1] = rng.randint(-200, 200, 300) * np.float32(0.025)
    tree = O.kd_create(base)
    deepest = 0
    for kind in ("random", "sorted_wall", "duplicates", "one_leaf"):
        for m in (1, 2, 17, 120):
            pts = batch(rng, m, kind)
            got, rounds = insert_in_rounds(tree, pts)
            assert got.tobytes() == sequential(tree, pts).tobytes(), (kind, m)
            deepest = max(deepest, rounds)
            # and a second batch on top of the first (new nodes are old nodes of the next frame)
            pts2 = batch(rng, m, "random")
            got2, _ = insert_in_rounds(got, pts2)
            assert got2.tobytes() == sequential(got, pts2).tobytes(), (kind, m, "second batch")
    assert deepest > 20  # the sorted wall really is a chain: one round per node
