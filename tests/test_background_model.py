"""CPU: background-stage model against golden vectors produced by the reference's own
gaussian_splatting/gm_background.py (tests/golden/background.npz): optimiser surgery of densify / clone / split /
prune (same seed -> same torch.normal samples on the CPU), densification statistics, opacity reset, pruning helpers."""
import os
from types import SimpleNamespace

import numpy as np
import torch

from fluidnexus_amd.gaussian_splatting.gm_background import _PARAMS, GaussianModel

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "background.npz"))
ARGS = SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                       position_lr_max_steps=30000, color_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)


def close(a, ref, rtol=1e-6):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return a.shape == ref.shape and np.abs(a - ref).max() <= rtol * (np.abs(ref).max() + 1e-30)


def test_background_stage_sequence_matches_reference():
    gm = GaussianModel()
    gm.spatial_lr_scale = 1.3
    N = G["init_xyz"].shape[0]
    for name, attr in _PARAMS:
        setattr(gm, attr, torch.nn.Parameter(torch.tensor(G[f"init_{name}"]).requires_grad_(True)))
    gm.max_radii2D = torch.zeros(N)
    gm.training_setup(ARGS)
    assert [g["name"] for g in gm.optimizer.param_groups] == ["xyz", "color", "opacity", "scaling", "rotation"]
    assert abs(gm.update_learning_rate(1000) - float(G["lr_xyz_1000"])) < 1e-12
    for name, attr in _PARAMS:
        getattr(gm, attr).grad = torch.tensor(G[f"grad_{name}"])
    gm.optimizer.step()
    gm.optimizer.zero_grad(set_to_none=True)
    for it in range(2):
        vs = torch.zeros(N, 3, requires_grad=True)
        vs.grad = torch.tensor(G[f"vs_grad{it}"])
        gm.add_densification_stats(vs, torch.tensor(G[f"filter{it}"]))
    assert close(gm.xyz_gradient_accum.numpy(), G["accum"]) and np.array_equal(gm.denom.numpy(), G["denom"])
    gm.max_radii2D = torch.tensor(G["max_radii2D"])
    torch.manual_seed(123)
    gm.densify_and_prune(0.0002, 0.005, 2.0, 20)
    assert gm.get_xyz.shape[0] == int(G["n_after_densify"])
    for name, attr in _PARAMS:
        p = getattr(gm, attr)
        assert isinstance(p, torch.nn.Parameter) and p.requires_grad and p.is_leaf
        assert close(p.detach().numpy(), G[f"dp_{name}"]), name
        st = gm.optimizer.state[p]                      # the state follows the new Parameter object
        assert close(st["exp_avg"].numpy(), G[f"dp_exp_avg_{name}"]), name
        assert close(st["exp_avg_sq"].numpy(), G[f"dp_exp_avg_sq_{name}"]), name
    assert len(gm.optimizer.state) == 5
    assert [gm.xyz_gradient_accum.shape[0], gm.denom.shape[0], gm.max_radii2D.shape[0]] == list(G["dp_stats_shapes"])
    gm.reset_opacity()
    assert close(gm._opacity.detach().numpy(), G["reset_opacity"])
    assert float(gm.optimizer.state[gm._opacity]["exp_avg"].abs().max()) == float(G["reset_exp_avg_opacity_absmax"]) == 0.0
    gm.prune_large_points()
    assert gm.get_xyz.shape[0] == int(G["n_after_large"])
    gm._valid_min_y, gm._valid_max_z = -0.2, 0.1
    gm.prune_near_points()
    assert gm.get_xyz.shape[0] == int(G["n_after_near"])
    gm.set_cam_locations(G["cams"])
    assert close(gm.smoke_to_cams_dist.numpy(), G["smoke_to_cams_dist"])
    gm.prune_near_cam_points()
    assert gm.get_xyz.shape[0] == int(G["n_after_cam"])
    assert close(gm._xyz.detach().numpy(), G["final_xyz"])
    # the optimiser still steps on the surviving tensors
    for _, attr in _PARAMS:
        getattr(gm, attr).grad = torch.ones_like(getattr(gm, attr))
    before = gm._xyz.detach().clone()
    gm.optimizer.step()
    assert not torch.equal(before, gm._xyz.detach())


def test_background_ply_roundtrip_and_capture_restore(tmp_path):
    gm = GaussianModel()
    pcd = SimpleNamespace(points=np.random.RandomState(0).uniform(-1, 1, size=(40, 3)))
    gm.create_from_pcd(pcd, 2.0, device="cpu")
    assert torch.allclose(gm.get_opacity, torch.full((40, 1), 0.1)) and torch.allclose(gm.get_scaling, torch.full((40, 3), float(np.exp(-5.9))))
    gm.training_setup(ARGS)
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    gm.save_ply(path)
    other = GaussianModel()
    other.load_ply(path, device="cpu")
    for _, attr in _PARAMS:
        assert torch.equal(getattr(other, attr).detach(), getattr(gm, attr).detach()), attr
    snap = gm.capture()
    third = GaussianModel()
    third.restore(snap, ARGS)
    assert third.spatial_lr_scale == 2.0 and torch.equal(third._xyz, gm._xyz)
