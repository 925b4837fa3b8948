"""CPU: oracle/physics_oracle.py::PbfOracle vs golden vectors produced by the reference's own gm_dynamics.py
(guess_hidden_particles -> project_gas_constraints x n -> confirm_guess_hidden_particles ->
update_visual_particles; tests/golden/pbf.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle.physics_oracle import PbfOracle

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pbf.npz"))


def pbf_state(tag):
    H, p0, secs, sf, eps, k, relax, K_P, E_P, DQ_P, alpha, bmy, decay, iters = (float(v) for v in G[f"consts_{tag}"])
    o = PbfOracle(H=H, p0=p0, secs=secs, scale_factor=sf, eps=eps, buoyancy_max_y=bmy, k=k, relaxation=relax, K_P=K_P,
                  E_P=int(E_P), DQ_P=DQ_P)
    t = {k_: torch.tensor(G[f"{k_}_{tag}"]) for k_ in ("xyz0", "velocity0", "force0", "buoyancy0", "imass", "counts0",
                                                       "visual0", "gravity")}
    return o, t, dict(alpha=alpha, decay=decay, iters=int(iters))


def close(a, ref, rtol=2e-5):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - ref).max() <= rtol * (np.abs(ref).max() + 1e-12)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_pbf_oracle_matches_reference(tag):
    o, t, c = pbf_state(tag)
    cnt = o.neighbor_counts(t["xyz0"]).numpy()
    assert (cnt == G[f"neighbor_counts_{tag}"]).all() and ((cnt >= 1) == G[f"keep_mask_{tag}"]).all()
    vel, buo, force, est = o.guess_hidden_particles(t["xyz0"], t["velocity0"], t["force0"], t["buoyancy0"], t["gravity"],
                                                    c["alpha"], c["decay"])
    for name, got in (("velocity1", vel), ("buoyancy1", buo), ("force1", force), ("estimate1", est)):
        assert close(got.numpy(), G[f"{name}_{tag}"], 1e-6), name
    counts = torch.zeros_like(t["counts0"]) + c["iters"]  # guess zeroes the counts, update_solver_counts x iters
    assert np.array_equal(counts.numpy(), G[f"counts2_{tag}"])
    for it in range(c["iters"]):
        est, force, p_ratio, lambdas = o.project_gas_constraints(est, vel, force, t["imass"], counts)
        assert close(est.numpy(), G[f"estimate_it{it}_{tag}"]), f"estimate {it}"
        assert close(force.numpy(), G[f"force_it{it}_{tag}"]), f"force {it}"
        assert abs(float(p_ratio.mean()) - float(G[f"p_ratio_mean_it{it}_{tag}"])) < 1e-5
        assert abs(float(lambdas.mean()) - float(G[f"lambdas_mean_it{it}_{tag}"])) < 1e-5 * max(1.0, abs(float(lambdas.mean())))
    xyz, vel3 = o.confirm_guess_hidden_particles(t["xyz0"], est)
    assert close(xyz.numpy(), G[f"xyz3_{tag}"]) and close(vel3.numpy(), G[f"velocity3_{tag}"])
    vis = o.update_visual_particles(t["visual0"], est, vel3)
    assert close(vis.numpy(), G[f"visual3_{tag}"])
