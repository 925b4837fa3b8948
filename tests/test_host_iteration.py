"""The k-d tree forms of the host iteration's neighbour searches (oracle/host_iteration.py: what bench.py's cpu_baseline
times at BASELINE's sizes) against the brute-force oracle forms the parity tests use: same edge sets, same sums."""
import numpy as np
import torch

from oracle.host_iteration import KdPhysicsOracle, distance_loss_kdtree
from oracle.physics_oracle import PhysicsOracle, distance_loss_oracle


def _cloud(n, seed, box=6.0):
    rng = np.random.RandomState(seed)
    x = rng.uniform(0, box, size=(n, 3))
    x[5] = x[9]  # coincident points
    return torch.tensor(x, dtype=torch.float64)


def test_kdtree_edges_equal_brute_force():
    y, x = _cloud(400, 0), _cloud(300, 1)
    for r in (0.7, 2.0):
        a = PhysicsOracle._edges(y, x, r)
        b = KdPhysicsOracle._edges(y, x, r)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        a = PhysicsOracle._edges(x, x, r)  # radius_graph with self loops
        b = KdPhysicsOracle._edges(x, x, r)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and int((a[0] == a[1]).sum()) == x.shape[0]


def test_kdtree_physics_terms_equal_brute_force():
    x = (_cloud(500, 2, box=8.0) / 100.0).requires_grad_(True)
    prev = _cloud(500, 3, box=8.0)
    vis = _cloud(900, 4, box=8.0)
    imass = torch.ones(500, 1, dtype=torch.float64)
    for cls in (PhysicsOracle, KdPhysicsOracle):
        po = cls()
        out = po.visual_xyz_from_nn(x, prev, vis).sum() + po.gas_constraints_from_exyz_nn(x, imass).sum()
        g, = torch.autograd.grad(out, x)
        if cls is PhysicsOracle:
            ref, gref = out.detach(), g
        else:
            assert abs(float(out - ref)) <= 1e-9 * abs(float(ref)) and (g - gref).abs().max() <= 1e-9 * gref.abs().max()


def test_kdtree_distance_loss_equals_the_dense_form():
    x = _cloud(700, 5, box=1.0).numpy()
    for thr in (0.05, 0.2):
        l0, g0 = distance_loss_oracle(x, thr)
        l1, g1 = distance_loss_kdtree(x, thr)
        assert abs(l0 - l1) <= 1e-10 * max(l0, 1.0) and np.abs(g0 - g1).max() <= 1e-10 * max(np.abs(g0).max(), 1.0)
