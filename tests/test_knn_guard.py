"""KNN_K guard (gm.knn_k_report / assert_within_knn_k): host-side check that no neighbour list of the current state
reaches the cap at which the reference's torch_cluster searches truncate (gm_dynamics.py:1081-1515)."""
import numpy as np
import pytest
import torch

from fluidnexus_amd.gaussian_splatting.gm_dynamics import GaussianModel


def _model(spacing, n=6, knn_k=100):
    gm = GaussianModel(device="cpu")
    gm.setup_constants(H=2.0, KNN_K=knn_k)
    ax = np.arange(n) * spacing
    xyz = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    gm._xyz = torch.from_numpy(xyz)
    gm._estimate_xyz = gm._xyz.clone()
    gm._estimate_xyz_nn = (gm._xyz / gm.scale_factor).clone()
    gm._buoyancy = torch.zeros_like(gm._xyz)
    gm._force = torch.zeros_like(gm._xyz)
    gm._visual_xyz = gm._xyz[:50] + 0.25
    return gm


def test_report_counts_lists_like_the_reference_searches():
    gm = _model(spacing=1.0)
    rep = gm.assert_within_knn_k()
    # unit lattice, H = 2 (strict): self + 6 + 12 + 8 = 27 lattice points at distance < 2 of an interior particle
    assert rep["hidden_at_estimate"] == 27 and rep["hidden_at_guess"] == 27
    assert 27 <= rep["hidden_per_visual"] <= 40 and rep["within_cap"]


def test_dense_cloud_is_flagged():
    gm = _model(spacing=0.5, n=8)
    rep = gm.knn_k_report()
    assert rep["hidden_at_estimate"] > 100 and not rep["within_cap"]
    with pytest.raises(RuntimeError, match="KNN_K"):
        gm.assert_within_knn_k()
    gm.setup_constants(H=2.0, KNN_K=512)
    assert gm.knn_k_report()["within_cap"]
