"""Autograd wrappers of the fused particle-physics kernels (include/fnx_physics.h).

`density_ratio`  == the arithmetic of GaussianModel.get_gas_constraints_from_exyz_nn after the
                    scaling (gm_dynamics.py:1276-1292): radius_graph + poly6 + index_add_ + 2 divisions;
`visual_from_hidden` == GaussianModel.get_visual_xyz_from_nn after the scaling (gm_dynamics.py:1463-1496).
Both fuse the neighbour search; see the header for the neighbour rule and the KNN_K caveat.
"""
from __future__ import annotations

import torch

from . import _physics_lib as PL


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("fluidnexus_amd physics: tensors must be on a HIP device (no CPU path)")
    t = t.float() if t.dtype != torch.float32 else t
    return t if t.is_contiguous() else t.contiguous()


class HashGrid:
    """Opaque uniform hash grid over a point cloud (cell edge = H)."""

    def __init__(self, xyz: torch.Tensor, cell: float):
        lib = PL.physics()
        xyz = _req(xyz.detach())
        self.N = xyz.shape[0]
        self.cell = float(cell)
        self.blob = torch.empty(lib.fnx_grid_bytes(self.N), dtype=torch.uint8, device=xyz.device)
        PL.check(lib.fnx_grid_build(xyz.data_ptr() if self.N else None, self.N, self.cell, self.blob.data_ptr(),
                                    _stream()))


class _DensityRatio(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, imass, H, p0, grid):
        lib = PL.physics()
        xyz, imass = _req(xyz), _req(imass)
        N = xyz.shape[0]
        if grid is None:
            grid = HashGrid(xyz, H)
        out = torch.empty(N, 1, dtype=torch.float32, device=xyz.device)
        PL.check(lib.fnx_density_forward(xyz.data_ptr(), N, imass.data_ptr(), H, p0, grid.blob.data_ptr(),
                                         out.data_ptr(), _stream()))
        ctx.save_for_backward(xyz, imass, grid.blob)
        ctx.consts = (H, p0)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = PL.physics()
        xyz, imass, blob = ctx.saved_tensors
        H, p0 = ctx.consts
        N = xyz.shape[0]
        g = _req(g)
        dx = torch.empty_like(xyz)
        PL.check(lib.fnx_density_backward(xyz.data_ptr(), N, imass.data_ptr(), H, p0, blob.data_ptr(), g.data_ptr(),
                                          dx.data_ptr(), _stream()))
        return dx, None, None, None, None


def density_ratio(xyz, imass, H, p0, grid=None):
    """p_ratio [N,1] of positions xyz [N,3] (scaled units), inverse masses imass [N,1].
    `grid`: an up-to-date HashGrid over xyz (cell = H) to reuse; built here when None."""
    return _DensityRatio.apply(xyz, imass, float(H), float(p0), grid)


class _VisualFromHidden(torch.autograd.Function):
    @staticmethod
    def forward(ctx, visual, hidden, hidden_prev, H, secs, eps, visual_grid, hgrid):
        lib = PL.physics()
        visual, hidden, hidden_prev = _req(visual), _req(hidden), _req(hidden_prev)
        V, N = visual.shape[0], hidden.shape[0]
        if hgrid is None:
            hgrid = HashGrid(hidden, H)
        out = torch.empty_like(visual)
        sum_w = torch.empty(V, dtype=torch.float32, device=visual.device)
        wvel = torch.empty(V, 3, dtype=torch.float32, device=visual.device)
        PL.check(lib.fnx_visual_interp_forward(visual.data_ptr(), V, hidden.data_ptr(), hidden_prev.data_ptr(), N, H,
                                               secs, eps, hgrid.blob.data_ptr(), out.data_ptr(), sum_w.data_ptr(),
                                               wvel.data_ptr(), _stream()))
        if visual_grid is None:
            visual_grid = HashGrid(visual, H)
        ctx.save_for_backward(visual, hidden, hidden_prev, sum_w, wvel, visual_grid.blob)
        ctx.consts = (H, secs, eps)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = PL.physics()
        visual, hidden, hidden_prev, sum_w, wvel, vblob = ctx.saved_tensors
        H, secs, eps = ctx.consts
        g = _req(g)
        dh = torch.empty_like(hidden)
        PL.check(lib.fnx_visual_interp_backward(visual.data_ptr(), visual.shape[0], hidden.data_ptr(),
                                                hidden_prev.data_ptr(), hidden.shape[0], H, secs, eps,
                                                vblob.data_ptr(), sum_w.data_ptr(), wvel.data_ptr(), g.data_ptr(),
                                                dh.data_ptr(), _stream()))
        return None, dh, None, None, None, None, None, None


def visual_from_hidden(visual, hidden, hidden_prev, H, secs, eps=1e-8, visual_grid=None, hidden_grid=None):
    """visual [V,3] (constant), hidden [N,3] (differentiable), hidden_prev [N,3] -> advected visual [V,3].
    `visual_grid`: a HashGrid over `visual` to reuse across iterations (visual is fixed within a frame);
    `hidden_grid`: an up-to-date HashGrid over `hidden`."""
    return _VisualFromHidden.apply(visual, hidden, hidden_prev, float(H), float(secs), float(eps), visual_grid,
                                   hidden_grid)
