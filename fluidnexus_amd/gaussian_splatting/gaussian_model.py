"""`gm_gs`: the vanilla 3D-Gaussian model with spherical-harmonics colour (reference
FluidDynamics/gaussian_splatting/gaussian_model.py:21-200) -- the only model whose colour reaches the rasteriser as SH
coefficients (`renderer.render`, pipe.py:74-98), i.e. the SH -> RGB path of csrc/raster_forward.hip / raster_backward.hip.
No shipped configuration selects it (SURVEY finding 3); this is the part of it the SH pipe and its optimisation need:
state tensors in the reference's layout, activations, getters, initialisation from a point cloud, optimiser groups and
the position learning-rate schedule.  Densification / PLY I/O of the vanilla model are not part of the hot path."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .. import physics
from ..utils.general_utils import build_scaling_rotation, get_expon_lr_func, inv_sigmoid, strip_symmetric
from ..utils.sh_utils import rgb2sh


class GaussianModel:
    def __init__(self, sh_degree: int = 3, device="cuda"):
        self.device = torch.device(device)
        self.active_sh_degree, self.max_sh_degree = 0, int(sh_degree)
        e = torch.empty(0, device=self.device)
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = e
        self.max_radii2D = self.xyz_gradient_accum = self.denom = e
        self.optimizer, self.percent_dense, self.spatial_lr_scale = None, 0, 0
        self.scaling_activation, self.scaling_inverse_activation = torch.exp, torch.log
        self.opacity_activation, self.opacity_inverse_activation = torch.sigmoid, inv_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    # getters the SH pipe reads (pipe.py:57-80)
    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: s.scaling_activation(s._scaling))
    get_rotation = property(lambda s: s.rotation_activation(s._rotation))
    get_opacity = property(lambda s: s.opacity_activation(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))  # [P, M, 3]

    def get_covariance(self, scaling_modifier=1):
        L = build_scaling_rotation(scaling_modifier * self.get_scaling, self._rotation)
        return strip_symmetric(L @ L.transpose(1, 2))

    def one_up_sh_degree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def create_from_pcd(self, pcd, spatial_lr_scale: float):
        """:125-146: DC coefficient from the point colours, log-scales from the mean distance to the three nearest
        points (simple-knn -> fnx_knn_mean_dist2), identity rotations, opacity 0.1."""
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.tensor(np.asarray(pcd.points), dtype=torch.float32, device=self.device)
        M = (self.max_sh_degree + 1) ** 2
        feats = torch.zeros(pts.shape[0], 3, M, device=self.device)
        feats[:, :, 0] = rgb2sh(torch.tensor(np.asarray(pcd.colors), dtype=torch.float32, device=self.device))
        dist2 = torch.clamp_min(physics.knn_mean_dist2(pts), 0.0000001)
        rots = torch.zeros(pts.shape[0], 4, device=self.device)
        rots[:, 0] = 1
        self._xyz = nn.Parameter(pts.requires_grad_(True))
        self._features_dc = nn.Parameter(feats[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(feats[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3).requires_grad_(True))
        self._rotation = nn.Parameter(rots.requires_grad_(True))
        self._opacity = nn.Parameter(inv_sigmoid(0.1 * torch.ones(pts.shape[0], 1, device=self.device)).requires_grad_(True))
        self.max_radii2D = torch.zeros(pts.shape[0], device=self.device)

    def training_setup(self, training_args):
        """:148-171: one Adam group per attribute (eps 1e-15), exponential schedule on the positions."""
        P = self.get_xyz.shape[0]
        self.percent_dense = training_args.percent_dense
        self.xyz_gradient_accum = torch.zeros(P, 1, device=self.device)
        self.denom = torch.zeros(P, 1, device=self.device)
        a = training_args
        groups = [("xyz", self._xyz, a.position_lr_init * self.spatial_lr_scale), ("f_dc", self._features_dc, a.feature_lr),
                  ("f_rest", self._features_rest, a.feature_lr / 20.0), ("opacity", self._opacity, a.opacity_lr),
                  ("scaling", self._scaling, a.scaling_lr), ("rotation", self._rotation, a.rotation_lr)]
        self.optimizer = torch.optim.Adam([{"params": [p], "lr": lr, "name": n} for n, p, lr in groups], lr=0.0, eps=1e-15)
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=a.position_lr_init * self.spatial_lr_scale,
                                                    lr_final=a.position_lr_final * self.spatial_lr_scale,
                                                    lr_delay_mult=a.position_lr_delay_mult, max_steps=a.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        for g in self.optimizer.param_groups:
            if g["name"] == "xyz":
                g["lr"] = self.xyz_scheduler_args(iteration)
                return g["lr"]
