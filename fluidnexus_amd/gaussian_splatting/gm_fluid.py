"""GaussianModel of the ScalarReal scenes (FluidDynamics/gaussian_splatting/gm_fluid.py): the fluid model without a
separate background set, rendered through render_fluid and the 1-channel rasteriser.

The reference's gm_fluid.py is a sibling copy of gm_dynamics.py; a normalised diff of the two leaves these
differences, which is all this subclass states: no `_gs_*` background group and no load_ply; the first-frame particle
clouds have hard-coded geometry (`create_particles_visual()` :468-487, `create_particles_hidden()` :490-528);
`prepare_emitter_points()` takes no arguments (hard-coded nozzle, :594-632); `load_visual` has no colour replication flag; `save_all` has no re-simulation flag; `record_time` times a solver step
with device events (:898-900, 988-995).  Physics terms, PBF solver, gradient caches and checkpoint formats are the
shared implementation of .gm_dynamics."""
from __future__ import annotations

import numpy as np
import torch

from .gm_dynamics import GaussianModel as _DynamicsModel


class GaussianModel(_DynamicsModel):
    # hard-coded geometry of the ScalarReal plume (gm_fluid.py:470-475, 492-497), world units
    VISUAL_INIT = dict(num_pts=600, radius_max=0.03, x_mid=0.34, y_min=-0.01, y_max=0.04, z_mid=-0.225)
    HIDDEN_INIT = dict(radius_max=0.1, delta=0.009, x_mid=0.34, y_min=-0.02, y_max=0.08, z_mid=-0.225)

    @torch.no_grad()
    def create_particles_visual(self):
        """:468-487: 600 visual particles in a disc-shaped slab above the inlet, draws in the reference's order."""
        g = self.VISUAL_INIT
        n = g["num_pts"]
        y = np.random.uniform(g["y_min"], g["y_max"], (n, 1))
        radius = np.random.random((n, 1)) * g["radius_max"]
        theta = np.random.random((n, 1)) * 2 * np.pi
        xyz = np.concatenate((radius * np.cos(theta) + g["x_mid"], y, radius * np.sin(theta) + g["z_mid"]), axis=1)
        self._visual_xyz = torch.from_numpy(xyz).float().to(self.device)
        self._visual_grid = None
        self.visual_particles_created = True

    @torch.no_grad()
    def create_particles_hidden(self):
        """:490-528: lattice of spacing 0.009 inside the pillar of radius 0.1, at rest."""
        g = self.HIDDEN_INIT
        self.init_hidden_velocity = 0.0
        self._init_hidden_state(self._pillar_lattice(g["x_mid"], g["z_mid"], g["radius_max"], g["y_min"], g["y_max"],
                                                     g["delta"]))

    # hard-coded nozzle of the ScalarReal scenes (gm_fluid.py:594-632): the dynamics model reads it from model_args
    EMITTER = dict(emitter_hidden_delta=0.015, emitter_visual_delta=0.00625, init_x_mid=0.34, init_z_mid=-0.225,
                   emitter_center_y_hidden=-0.02, emitter_center_y_visual=-0.01, emitter_visual_radius_ratio=4,
                   emitter_hidden_radius_ratio=6)

    @torch.no_grad()
    def prepare_emitter_points(self):
        """:594-632: NO arguments in this model (entries_scalar_real/train_physical_particle.py calls it so) -- the same
        one-layer disc lattices as the dynamics model's, from the constants above."""
        from types import SimpleNamespace
        return super().prepare_emitter_points(SimpleNamespace(**self.EMITTER))

    def load_visual(self, checkpoint_path, frame_idx, scale=True, device="cuda"):
        return super().load_visual(checkpoint_path, frame_idx, scale=scale, color_3ch=False, device=device)

    def save_all(self, checkpoint_path, frame_idx):
        return super().save_all(checkpoint_path, frame_idx)

    def load_ply(self, *args, **kwargs):
        raise AttributeError("gm_fluid has no background Gaussians (gm_fluid.py defines no load_ply)")

    @torch.no_grad()
    def project_gas_constraints(self):
        """:898-995: with record_time the solver step is bracketed by device events and reports "elapsed_time" (ms),
        which costs a host sync like the reference's torch.cuda.synchronize()."""
        if not getattr(self, "record_time", False):
            return super().project_gas_constraints()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        ret = super().project_gas_constraints()
        end.record()
        end.synchronize()
        ret["elapsed_time"] = start.elapsed_time(end)
        return ret
