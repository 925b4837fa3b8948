"""Background-stage Gaussian model: the static scene Gaussians trained before the fluid stages
(FluidDynamics/gaussian_splatting/gm_background.py).  Same attribute and method names as the
reference class, so `train_background.py`'s call sequence (training_setup, add_densification_stats,
densify_and_prune, reset_opacity, prune_*, save_ply / load_ply) runs unchanged; rendering goes through
`renderer.render_background` on the MI355X rasteriser.

Everything here is tensor bookkeeping in PyTorch (no kernels of its own) and device-agnostic: tensors follow the
device of `_xyz`, where the reference hard-codes "cuda".  The five trainable tensors and their optimiser state
are handled by one generic routine (`_rebuild`) instead of one copy per operation."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn

from ..utils.general_utils import build_rotation, build_scaling_rotation, get_expon_lr_func, inv_sigmoid, strip_symmetric
from ..utils.ply_io import read_vertex_ply, write_vertex_ply
from ..utils.sh_utils import rgb2sh

# optimiser group name -> attribute (gm_background.py:158-164)
_PARAMS = (("xyz", "_xyz"), ("color", "_color"), ("opacity", "_opacity"), ("scaling", "_scaling"), ("rotation", "_rotation"))


class GaussianModel:
    def __init__(self, *args, **kwargs):
        e = torch.empty(0)
        self.active_sh_degree = self.max_sh_degree = 0
        self._xyz = self._color = self._scaling = self._rotation = self._opacity = e
        self.max_radii2D = self.xyz_gradient_accum = self.denom = e
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        # activations (gm_background.py:22-37)
        self.scaling_activation, self.scaling_inverse_activation = torch.exp, torch.log
        self.opacity_activation, self.opacity_inverse_activation = torch.sigmoid, inv_sigmoid
        self.rotation_activation = torch.nn.functional.normalize
        self.covariance_activation = lambda scaling, modifier, rotation: strip_symmetric(
            (lambda L: L @ L.transpose(1, 2))(build_scaling_rotation(modifier * scaling, rotation)))

    # -- getters (:91-112) --------------------------------------------------------------------------
    get_xyz = property(lambda s: s._xyz)
    get_color = property(lambda s: s._color)
    get_scaling = property(lambda s: s.scaling_activation(s._scaling))
    get_rotation = property(lambda s: s.rotation_activation(s._rotation))
    get_opacity = property(lambda s: s.opacity_activation(s._opacity))

    def get_covariance(self, scaling_modifier=1):
        return self.covariance_activation(self.get_scaling, scaling_modifier, self._rotation)

    def one_up_sh_degree(self):
        pass  # plain colours, no SH bands in this model (:113-115)

    @property
    def _dev(self):
        return self._xyz.device

    def capture(self):
        """:53-66"""
        return (self.active_sh_degree, self._xyz, self._color, self._scaling, self._rotation, self._opacity,
                self.max_radii2D, self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(), self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        """:68-88"""
        (self.active_sh_degree, self._xyz, self._color, self._scaling, self._rotation, self._opacity, self.max_radii2D,
         accum, denom, opt_dict, self.spatial_lr_scale) = model_args
        self.training_setup(training_args)
        self.xyz_gradient_accum, self.denom = accum, denom
        self.optimizer.load_state_dict(opt_dict)

    # -- construction ---------------------------------------------------------------------------------
    def create_from_pcd(self, pcd, spatial_lr_scale, device="cuda"):
        """:117-143: grey 0.7, log-scale -5.9, identity rotation, opacity 0.1 on the given points."""
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.tensor(np.asarray(pcd.points)).float().to(device)
        n = pts.shape[0]
        rots = torch.zeros((n, 4), device=device)
        rots[:, 0] = 1
        values = dict(xyz=pts, color=torch.zeros((n, 3), device=device) + 0.7, scaling=torch.zeros((n, 3), device=device) - 5.9,
                      rotation=rots, opacity=inv_sigmoid(0.1 * torch.ones((n, 1), device=device)))
        for name, attr in _PARAMS:
            setattr(self, attr, nn.Parameter(values[name].contiguous().requires_grad_(True)))
        self.max_radii2D = torch.zeros(n, device=device)
        self._valid_min_y, self._valid_max_z = -0.04, -0.45
        self._object_ball_center = torch.tensor([0.328, 0.378, -0.28], device=device).view(1, 3)
        self._object_ball_radius = 0.11 + 0.02

    def set_cam_locations(self, cam_locations):
        """:145-148"""
        self.smoke_location = torch.tensor([0.328, -0.04, -0.34], device=self._dev).view(1, 3)
        self.cam_locations = torch.from_numpy(np.asarray(cam_locations)).to(self._dev)
        self.smoke_to_cams_dist = torch.norm(self.smoke_location.unsqueeze(1) - self.cam_locations.unsqueeze(0), dim=2)

    def set_near_params(self, optim_args):
        self._valid_min_y, self._valid_max_z = optim_args.valid_min_y, optim_args.valid_max_z

    def training_setup(self, training_args):
        """:154-174: one Adam group per tensor (eps 1e-15), exponential position schedule."""
        self.percent_dense = training_args.percent_dense
        self._reset_stats()
        lrs = dict(xyz=training_args.position_lr_init * self.spatial_lr_scale, color=training_args.color_lr,
                   opacity=training_args.opacity_lr, scaling=training_args.scaling_lr, rotation=training_args.rotation_lr)
        self.optimizer = torch.optim.Adam([{"params": [getattr(self, attr)], "lr": lrs[name], "name": name}
                                           for name, attr in _PARAMS], lr=0.0, eps=1e-15)
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=training_args.position_lr_init * self.spatial_lr_scale,
                                                    lr_final=training_args.position_lr_final * self.spatial_lr_scale,
                                                    lr_delay_mult=training_args.position_lr_delay_mult,
                                                    max_steps=training_args.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        """:176-182"""
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                group["lr"] = self.xyz_scheduler_args(iteration)
                return group["lr"]

    # -- scene files (:184-267) -----------------------------------------------------------------------
    def construct_list_of_attributes(self):
        c = self._color.shape[1]
        return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(c)] + [f"f_rest_{i}" for i in range(c)]
                + ["opacity"] + [f"scale_{i}" for i in range(self._scaling.shape[1])]
                + [f"rot_{i}" for i in range(self._rotation.shape[1])] + [f"color_{i}" for i in range(c)])

    def save_ply(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        xyz = self._xyz.detach().cpu().numpy().copy()
        xyz[:, :2] *= -1.0  # x, y negated in the file (supersplat convention, :209-210)
        color = self._color.detach().cpu().numpy()
        cols = np.concatenate((xyz, np.zeros_like(xyz), rgb2sh(color), np.zeros_like(color), self._opacity.detach().cpu().numpy(),
                               self._scaling.detach().cpu().numpy(), self._rotation.detach().cpu().numpy(), color), axis=1)
        write_vertex_ply(path, self.construct_list_of_attributes(), cols)

    def load_ply(self, path, device="cuda"):
        names, col = read_vertex_ply(path)
        xyz = np.stack((col["x"] * -1.0, col["y"] * -1.0, col["z"]), axis=1)

        def group(prefix):
            ks = sorted((n for n in names if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
            return np.stack([col[k] for k in ks], axis=1)

        values = dict(xyz=xyz, color=group("color_"), opacity=col["opacity"][..., np.newaxis], scaling=group("scale_"),
                      rotation=group("rot"))
        for name, attr in _PARAMS:
            setattr(self, attr, nn.Parameter(torch.tensor(values[name], dtype=torch.float, device=device).requires_grad_(True)))
        self.active_sh_degree = self.max_sh_degree

    # -- optimiser surgery (:269-352): every operation maps each tensor p -> f(name, p), its Adam moments -> g(m) ---
    def _rebuild(self, new_param, new_moment):
        for group in self.optimizer.param_groups:
            assert len(group["params"]) == 1
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            param = nn.Parameter(new_param(group["name"], old).requires_grad_(True))
            if state is not None:
                state["exp_avg"], state["exp_avg_sq"] = new_moment(group["name"], state["exp_avg"]), new_moment(
                    group["name"], state["exp_avg_sq"])
                self.optimizer.state[param] = state
            group["params"][0] = param
            setattr(self, dict(_PARAMS)[group["name"]], param)

    def _reset_stats(self):
        n = self.get_xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((n, 1), device=self._dev)
        self.denom = torch.zeros((n, 1), device=self._dev)

    def replace_tensor_to_optimizer(self, tensor, name):
        """:269-282: swap one tensor in, zero its Adam moments."""
        out = {}
        for group in self.optimizer.param_groups:
            if group["name"] != name:
                continue
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            param = nn.Parameter(tensor.requires_grad_(True))
            if state is not None:
                state["exp_avg"], state["exp_avg_sq"] = torch.zeros_like(tensor), torch.zeros_like(tensor)
                self.optimizer.state[param] = state
            group["params"][0] = param
            out[name] = param
        return out

    def prune_points(self, mask):
        """:303-317: drop the points where mask is True (tensors, moments, statistics)."""
        keep = ~mask
        self._rebuild(lambda _n, p: p[keep], lambda _n, m: m[keep])
        self.xyz_gradient_accum, self.denom, self.max_radii2D = self.xyz_gradient_accum[keep], self.denom[keep], self.max_radii2D[keep]

    def densification_postfix(self, new_xyz, new_color, new_opacities, new_scaling, new_rotation):
        """:352-378: append new points with zero moments; the statistics start over."""
        ext = dict(xyz=new_xyz, color=new_color, opacity=new_opacities, scaling=new_scaling, rotation=new_rotation)
        self._rebuild(lambda n, p: torch.cat((p, ext[n]), dim=0), lambda n, m: torch.cat((m, torch.zeros_like(ext[n])), dim=0))
        self._reset_stats()
        self.max_radii2D = torch.zeros(self.get_xyz.shape[0], device=self._dev)

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        """:404-418: small Gaussians with a large view-space gradient are duplicated in place."""
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & (
            torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent)
        self.densification_postfix(self._xyz[sel], self._color[sel], self._opacity[sel], self._scaling[sel], self._rotation[sel])

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        """:380-402: large ones are replaced by N samples of themselves, 1.6x smaller."""
        n = self.get_xyz.shape[0]
        padded = torch.zeros(n, device=self._dev)
        padded[: grads.shape[0]] = grads.squeeze()
        sel = (padded >= grad_threshold) & (torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
        stds = self.get_scaling[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=self._dev), std=stds)
        rots = build_rotation(self._rotation[sel]).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[sel].repeat(N, 1)
        new_scaling = self.scaling_inverse_activation(self.get_scaling[sel].repeat(N, 1) / (0.8 * N))
        self.densification_postfix(new_xyz, self._color[sel].repeat(N, 1), self._opacity[sel].repeat(N, 1), new_scaling,
                                   self._rotation[sel].repeat(N, 1))
        self.prune_points(torch.cat((sel, torch.zeros(N * int(sel.sum()), device=self._dev, dtype=torch.bool))))

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, **kwargs):
        """:420-436"""
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent)
        prune = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            prune = prune | (self.max_radii2D > max_screen_size) | (self.get_scaling.max(dim=1).values > 0.1 * extent)
        self.prune_points(prune)
        if self._dev.type == "cuda":
            torch.cuda.empty_cache()

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """:473-477: accumulate the norm of the 2D-mean gradient of the visible points."""
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1

    def reset_opacity(self):
        """:227-230: clamp opacities to <= 0.01 and forget their Adam moments."""
        new = inv_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        self._opacity = self.replace_tensor_to_optimizer(new, "opacity")["opacity"]

    # -- scene-specific pruning helpers (:438-471) ---------------------------------------------------------
    def check_outside_object(self):
        return torch.sum((self.get_xyz - self._object_ball_center) ** 2, dim=1) > self._object_ball_radius ** 2

    def prune_near_points(self, prune_near_with_object=False):
        mask = (self.get_xyz[:, 2] > self._valid_max_z) & (self.get_xyz[:, 1] > self._valid_min_y)
        if prune_near_with_object:
            mask = mask & self.check_outside_object()
        self.prune_points(mask)

    def prune_near_cam_points(self):
        d = torch.norm(self.get_xyz.unsqueeze(1) - self.cam_locations.unsqueeze(0), dim=2)
        self.prune_points(torch.any(d < self.smoke_to_cams_dist, dim=1))

    def prune_large_points(self):
        self.prune_points(self.get_scaling.max(dim=1).values > 0.03)
