"""-m gpu: the fused PBF predictor / solver (fnx_pbf_*, fnx_visual_advect behind GaussianModel's own method names)
against golden vectors produced by the reference's gm_dynamics.py (tests/golden/pbf.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pbf.npz"))


def close(a, ref, rtol=2e-5):
    a, ref = a.detach().cpu().double().numpy(), np.asarray(ref, np.float64)
    return np.abs(a - ref).max() <= rtol * (np.abs(ref).max() + 1e-12)


def _model(tag):
    from fluidnexus_amd.gaussian_splatting.gm_dynamics import GaussianModel
    H, p0, secs, sf, eps, k, relax, K_P, E_P, DQ_P, alpha, bmy, decay, iters = (float(v) for v in G[f"consts_{tag}"])
    gm = GaussianModel()
    gm.setup_constants(H=H, KNN_K=100, p0=p0, secs=secs, k=k, buoyancy_max_y=bmy)
    gm.setup_solver_constants(alpha=alpha, buoyancy_decay_rate=decay, min_neighbors=1, gravity=G[f"gravity_{tag}"].reshape(3))
    assert gm.scale_factor == sf and gm.EPSILON == eps and (gm.RELAXATION, gm.K_P, gm.E_P, gm.DQ_P) == (relax, K_P, E_P, DQ_P)
    t = lambda n: torch.tensor(G[f"{n}_{tag}"]).cuda()  # noqa: E731
    gm._xyz, gm._estimate_xyz, gm._velocity, gm._force = t("xyz0"), t("xyz0").clone(), t("velocity0"), t("force0")
    gm._buoyancy, gm._imass, gm._counts, gm._visual_xyz = t("buoyancy0"), t("imass"), t("counts0"), t("visual0")
    return gm, int(iters)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_pbf_frame_step_matches_reference(tag):
    gm, iters = _model(tag)
    cnt = gm.neighbor_counts().cpu().numpy()
    assert (cnt == G[f"neighbor_counts_{tag}"]).all()
    gm.guess_hidden_particles(stable=False, use_wind=False)
    for name, got in (("velocity1", gm._velocity), ("buoyancy1", gm._buoyancy), ("force1", gm._force),
                      ("estimate1", gm._estimate_xyz), ("counts1", gm._counts)):
        assert close(got, G[f"{name}_{tag}"], 1e-6), name
    for _ in range(iters):
        gm.update_solver_counts()
    assert np.array_equal(gm._counts.cpu().numpy(), G[f"counts2_{tag}"])
    for it in range(iters):
        assert gm.project_gas_constraints() == {}
        assert close(gm._estimate_xyz, G[f"estimate_it{it}_{tag}"]), f"estimate after solver iteration {it}"
        assert close(gm._force, G[f"force_it{it}_{tag}"]), f"force after solver iteration {it}"
    moved = np.abs(G[f"estimate_it{iters - 1}_{tag}"] - G[f"estimate1_{tag}"]).max()
    assert moved > 0.05  # the solver really moved the particles
    assert np.abs(gm._estimate_xyz.cpu().numpy() - G[f"estimate_it{iters - 1}_{tag}"]).max() <= 1e-4 * moved + 2e-5
    gm.confirm_guess_hidden_particles()
    assert close(gm._xyz, G[f"xyz3_{tag}"]) and close(gm._velocity, G[f"velocity3_{tag}"], 1e-4)
    gm.update_visual_particles()
    assert close(gm._visual_xyz, G[f"visual3_{tag}"])
    assert np.abs(gm._visual_xyz.cpu().numpy() - G[f"visual3_{tag}"]).max() <= 1e-4 * np.abs(G[f"visual3_{tag}"] - G[f"visual0_{tag}"]).max() + 2e-5


def test_remove_invalid_particles_and_stable_mode():
    gm, _ = _model("a")
    n0 = gm._xyz.shape[0]
    gm._particle_id = torch.arange(n0, device="cuda")
    gm.remove_invalid_particles()
    keep = G["keep_mask_a"]
    assert gm._xyz.shape[0] == int(keep.sum()) < n0
    assert torch.equal(gm._particle_id.cpu(), torch.arange(n0)[torch.tensor(keep)])
    for name in ("_estimate_xyz", "_buoyancy", "_force", "_velocity", "_imass", "_counts"):
        assert getattr(gm, name).shape[0] == gm._xyz.shape[0]
    # stable mode: 0.01 s step and unit downward buoyancy (alpha = -1), gm_dynamics.py:980-983
    v0 = gm._velocity.clone()
    f0 = gm._force.clone()
    gm.guess_hidden_particles(stable=True)
    exp_v = v0 + (torch.tensor([[0.0, 9.8, 0.0]], device="cuda") * 0.01 + 0.01 * f0)
    assert torch.allclose(gm._velocity, exp_v, rtol=1e-6, atol=1e-6)
    assert torch.allclose(gm._estimate_xyz, gm._xyz + 0.01 * gm._velocity, rtol=1e-6, atol=1e-6)
    with pytest.raises(NotImplementedError):
        gm.guess_hidden_particles(use_wind=True)
