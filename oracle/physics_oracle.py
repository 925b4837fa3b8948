"""CPU restatement (PyTorch fp32, brute-force neighbour search) of the reference's physics-informed
losses -- TEST INFRASTRUCTURE ONLY.  Follows FluidDynamics/gaussian_splatting/gm_dynamics.py:
poly6 :188-191, get_guess_hidden_particles_from_nn :1014-1030, get_gas_constraints_from_exyz_nn
:1269-1294, get_gas_constraints_from_vel_nn_guess :1296-1320, get_visual_xyz_from_nn :1453-1498.

PARITY STATUS: pinned by tests/golden/physics.npz, which holds outputs and autograd gradients of
the reference's own methods (imported with the missing third-party modules stubbed, see
tests/golden/gen_reference_golden.py).  The edge list comes from torch_cluster.radius{,_graph}
(1.6.3, not vendored): parity is unpinned at that boundary; here an edge is every ordered pair
with distance < r (self-loop included), and clouds are kept below KNN_K neighbours per particle.
"""
from __future__ import annotations

import numpy as np
import torch


class PhysicsOracle:
    def __init__(self, H=2.0, p0=1.5, secs=0.033, scale_factor=100.0, eps=1e-8, buoyancy_max_y=0.0):
        self.H, self.p0, self.secs, self.scale_factor, self.EPSILON = H, p0, secs, scale_factor, eps
        self.H2 = H ** 2
        self.poly6_term1 = 315.0 / (64.0 * np.pi * H ** 9)  # gm_dynamics.py:130
        self.buoyancy_max_y = buoyancy_max_y

    def poly6(self, r2):
        return (r2 < self.H2) * self.poly6_term1 * ((self.H2 - r2) ** 3)

    @staticmethod
    def _edges(y, x, r):
        """row indexes y (queries), col indexes x: all pairs with ||y_row - x_col|| < r."""
        d = torch.cdist(y.detach().double(), x.detach().double())
        return torch.nonzero(d < r, as_tuple=True)

    def p_ratio(self, xyz_scaled, imass):
        N = xyz_scaled.shape[0]
        row, col = self._edges(xyz_scaled, xyz_scaled, self.H)
        diff = xyz_scaled[row] - xyz_scaled[col]
        vals = self.poly6(torch.sum(diff ** 2, dim=1))
        pi = torch.zeros(N, dtype=xyz_scaled.dtype).index_add_(0, row, vals)
        return pi.unsqueeze(1) / imass / self.p0

    def gas_constraints_from_exyz_nn(self, x_nn, imass):
        return self.p_ratio(x_nn * self.scale_factor, imass)

    def guess_hidden_particles_from_nn(self, x_nn, x_prev, buoyancy, force):
        if self.buoyancy_max_y > 0.0:
            cur_buoyancy = buoyancy * (1.0 - (x_nn[:, 1:2] / self.buoyancy_max_y))
        else:
            cur_buoyancy = buoyancy
        tmp_velocity = (x_nn * self.scale_factor - x_prev) / self.secs
        estimate_velocity = tmp_velocity + cur_buoyancy * self.secs + self.secs * force
        return x_nn * self.scale_factor + self.secs * estimate_velocity

    def gas_constraints_from_vel_nn_guess(self, x_nn, x_prev, imass, buoyancy, force):
        return self.p_ratio(self.guess_hidden_particles_from_nn(x_nn, x_prev, buoyancy, force), imass)

    def visual_xyz_from_nn(self, x_nn, x_prev, visual_xyz):
        visual_xyz = visual_xyz.detach()
        est = x_nn * self.scale_factor
        vel = (est - x_prev) / self.secs
        V = visual_xyz.shape[0]
        row, col = self._edges(visual_xyz, est, self.H)
        diff = visual_xyz[row] - est[col]
        p6 = self.poly6(torch.sum(diff ** 2, dim=1))
        vv = torch.zeros(V, 3, dtype=est.dtype).index_add_(0, row, vel[col] * p6.unsqueeze(-1))
        sp = torch.zeros(V, dtype=est.dtype).index_add_(0, row, p6).clamp_min(self.EPSILON)
        return visual_xyz + vv * self.secs / sp.unsqueeze(-1)
