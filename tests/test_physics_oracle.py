"""CPU: oracle/physics_oracle.py vs golden vectors produced by the reference's own gm_dynamics.py."""
import os

import numpy as np
import pytest
import torch

from oracle.physics_oracle import PhysicsOracle

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "physics.npz"))


def _state(tag):
    H, K, p0, secs, sf, eps, bmy = G[f"consts_{tag}"]
    o = PhysicsOracle(H=float(H), p0=float(p0), secs=float(secs), scale_factor=float(sf), eps=float(eps),
                      buoyancy_max_y=float(bmy))
    t = {k: torch.tensor(G[f"{k}_{tag}"]) for k in ("x_prev", "x_est", "x_nn", "imass", "buoyancy", "force", "visual_xyz")}
    return o, t


@pytest.mark.parametrize("tag", ["a", "b"])
def test_physics_oracle_matches_reference(tag):
    o, t = _state(tag)

    def run(fn, w):
        x = t["x_nn"].clone().requires_grad_(True)
        val = fn(x)
        (val * torch.tensor(w)).sum().backward()
        return val.detach().numpy(), x.grad.numpy()

    v, g = run(lambda x: o.gas_constraints_from_exyz_nn(x, t["imass"]), G[f"w_gas_{tag}"])
    assert np.allclose(v, G[f"p_ratio_{tag}"], rtol=2e-6, atol=1e-7)
    assert np.allclose(g, G[f"d_gas_{tag}"], rtol=1e-4, atol=1e-4 * np.abs(G[f"d_gas_{tag}"]).max())
    v, g = run(lambda x: o.gas_constraints_from_vel_nn_guess(x, t["x_prev"], t["imass"], t["buoyancy"], t["force"]),
               G[f"w_next_{tag}"])
    assert np.allclose(v, G[f"p_ratio_next_{tag}"], rtol=2e-6, atol=1e-7)
    assert np.allclose(g, G[f"d_next_{tag}"], rtol=1e-4, atol=1e-4 * np.abs(G[f"d_next_{tag}"]).max())
    v, g = run(lambda x: o.visual_xyz_from_nn(x, t["x_prev"], t["visual_xyz"]), G[f"w_vis_{tag}"])
    assert np.allclose(v, G[f"vis_{tag}"], rtol=1e-6, atol=1e-6)
    assert np.allclose(g, G[f"d_vis_{tag}"], rtol=1e-4, atol=1e-4 * np.abs(G[f"d_vis_{tag}"]).max())
    guess = o.guess_hidden_particles_from_nn(t["x_nn"], t["x_prev"], t["buoyancy"], t["force"]).numpy()
    assert np.allclose(guess, G[f"guess_{tag}"], rtol=1e-6, atol=1e-6)
    assert np.allclose(o.poly6(torch.tensor(G[f"poly6_r2_{tag}"])).numpy(), G[f"poly6_{tag}"], rtol=1e-6)


@pytest.mark.parametrize("tag", ["plume", "blob", "sparse"])
def test_distance_loss_oracle_matches_reference(tag):
    """The float64 restatement against the reference's distance_loss on float64 copies of the same points; the
    reference's own fp32 evaluation (cdist's matrix-multiply form) agrees with both only to its cancellation noise."""
    from oracle.physics_oracle import distance_loss_oracle
    D = np.load(os.path.join(os.path.dirname(__file__), "golden", "distance_loss.npz"))
    loss, grad = distance_loss_oracle(D[f"pos_{tag}"], D[f"thr_{tag}"])
    g64 = D[f"grad64_{tag}"]
    scale = np.abs(g64).max() + 1e-30
    assert abs(loss - float(D[f"loss64_{tag}"])) <= 1e-9 * max(1.0, abs(loss))
    assert np.abs(grad - g64).max() <= 1e-9 * scale + 1e-30
    if tag != "sparse":
        assert loss > 0
        assert abs(float(D[f"loss32_{tag}"]) - loss) <= 1e-3 * loss          # the reference's fp32 noise
        assert np.abs(D[f"grad32_{tag}"] - g64).max() <= 2e-2 * scale


def test_capped_edges_keep_the_first_k_by_index():
    """max_num_neighbors as torch_cluster's CUDA kernel applies it (oracle header): hand case on a line."""
    from oracle.physics_oracle import PhysicsOracle
    x = torch.tensor([[0.0, 0, 0], [0.5, 0, 0], [1.0, 0, 0], [1.5, 0, 0], [5.0, 0, 0]])
    row, col = PhysicsOracle._edges_capped(x, x, 1.2, 2)
    kept = {q: [int(c) for r, c in zip(row, col) if int(r) == q] for q in range(5)}
    # within 1.2: 0 -> {0,1,2}; 1 -> {0,1,2,3}; 2 -> {0,1,2,3}; 3 -> {1,2,3}; 4 -> {4}; first two by index each
    assert kept == {0: [0, 1], 1: [0, 1], 2: [0, 1], 3: [1, 2], 4: [4]}
    o = PhysicsOracle(H=1.2, p0=1.0, knn_k=2)
    im = torch.ones(5, 1)
    w = lambda a, b: float(o.poly6(torch.tensor(float(a - b) ** 2)))  # noqa: E731
    # the kept edge (query q, neighbour i) adds at i (radius_graph's source_to_target swap + index_add_ on row)
    want = [w(0, 0) + w(0.5, 0) + w(1.0, 0), w(0, 0.5) + w(0.5, 0.5) + w(1.0, 0.5) + w(1.5, 0.5), w(1.5, 1.0), 0.0, w(0, 0)]
    got = o.p_ratio(x, im).squeeze(1)
    assert torch.allclose(got, torch.tensor(want), rtol=1e-6)
    # a cap no list reaches changes nothing
    big, none = PhysicsOracle(H=1.2, p0=1.0, knn_k=50), PhysicsOracle(H=1.2, p0=1.0)
    assert torch.allclose(big.p_ratio(x, im), none.p_ratio(x, im), rtol=1e-6)
    v = torch.tensor([[0.6, 0.1, 0.0], [4.8, 0, 0]])
    xp = x - 0.05
    assert torch.allclose(big.visual_xyz_from_nn(x / 100, xp, v), none.visual_xyz_from_nn(x / 100, xp, v))
    xq = x * 0.9 - 0.05  # velocities that differ from particle to particle: the kept subset shows in the average
    assert not torch.allclose(o.visual_xyz_from_nn(x / 100, xq, v)[0], none.visual_xyz_from_nn(x / 100, xq, v)[0])
