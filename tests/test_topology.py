"""Topology graph + loop-closure proposal (kernel.cu:623-795, the rows SURVEY 8f marks "next"): oracle known answers on
the CPU, product vs oracle on the GPU."""
import numpy as np
import pytest

import oracle_lib as O


def loop_trajectory():
    """A 12 x 12 m square driven in 0.5 m steps and closed: the graph distance back to the start exceeds 20 m while the
    map distance drops below 6 m."""
    pts = []
    for k in range(24): pts.append((0.5 * k, 0.0))
    for k in range(24): pts.append((12.0, 0.5 * k))
    for k in range(24): pts.append((12.0 - 0.5 * k, 12.0))
    for k in range(22): pts.append((0.0, 12.0 - 0.5 * k))
    return [np.array([x - 6.0, y - 6.0, 0.0], np.float32) for x, y in pts]


def walled_grid():
    grid = np.full((1600, 1600), -100, np.int8)
    grid[800 - 100:800 + 100, 800] = 113     # a confident wall through the middle of the square (x in [-2.5, 2.5), y = 0)
    grid[900, 700:1100] = 31                 # barely confident
    grid[1000, 700:1100] = 30                # NOT above WALL_CONFIDENCE
    return grid


def test_oracle_topology_known_answers(oracle):
    t = O.Topology()
    created = [t.update(p) for p in loop_trajectory()]
    nodes = t.nodes()
    # node 0 is the origin (particleFilterInit); the start pose (-6, -6) is 8.5 m away -> node 1 at once, then a node
    # whenever the robot is > 2.5 m from every node: 3.0 m further along the first edge
    assert created[:8] == [1, 0, 0, 0, 0, 0, 1, 0]
    assert np.allclose(nodes[0], (0, 0, np.float32(np.sqrt(np.float32(72.0))) + 45.0)) and np.allclose(nodes[1, :2], (-6, -6)) and np.allclose(nodes[2, :2], (-3, -6))
    assert t.n_nodes == len(nodes) and t.node_idx == t.n_nodes - 1
    assert nodes[-1, 2] == 0.0 and (np.diff(nodes[1:, 2]) < 0).all()     # graph distance to the current node decreases along the path
    grid = walled_grid()
    assert O.find_walls(grid, (-6.0, -3.0), (6.0, -3.0)) == 0
    assert O.find_walls(grid, (0.0, -5.0), (0.0, 5.0)) == 1               # crosses the 113 wall once
    assert O.find_walls(grid, (0.0, 0.5), (7.0, 0.5)) == 1                # crosses the 31 row, the 30 row does not count
    pairs = t.loop_closure(grid, loop_trajectory()[-1])
    assert len(pairs) > 0 and (pairs[:, 0] < t.n_nodes).all()


@pytest.mark.gpu
def test_topology_and_loop_closure_match_oracle_on_gpu(pkg):
    assert pkg.device_count() > 0
    grid = walled_grid()
    h = pkg.PfSlam(64)
    h.set_grid(grid)
    t = O.Topology()
    rng = np.random.RandomState(0)
    for p in loop_trajectory():
        p = (p + np.array([rng.normal(0, 0.01), rng.normal(0, 0.01), 0], np.float32)).astype(np.float32)
        h.set_pose(p)
        n = h.topology_update()
        t.update(p)
        assert n == t.n_nodes
        got = h.check_loop_closure()
        want = t.loop_closure(grid, p)
        assert got.tolist() == want.tolist()
    nodes, idx = h.topology()
    assert idx == t.node_idx and (nodes.view(np.int32) == t.nodes().view(np.int32)).all()
    for _ in range(200):
        a, b = rng.uniform(-19, 19, 2), rng.uniform(-19, 19, 2)
        assert h.find_walls(a, b) == O.find_walls(grid, a, b)
    # rays leaving the grid are clipped exactly like traceRay
    assert h.find_walls((-25.0, 0.0), (25.0, 3.0)) == O.find_walls(grid, (-25.0, 0.0), (25.0, 3.0))
    h.close()
