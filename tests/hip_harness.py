"""Test helper: run the HIP rasteriser through the C ABI and pull every intermediate out of the
opaque scratch blobs so that it can be compared with the oracle's arrays."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from fluidnexus_amd import _lib


def _t(a, dev, dtype=torch.float32):
    if a is None:
        return None
    return torch.as_tensor(np.ascontiguousarray(a)).to(dev, dtype).contiguous()


def _p(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


def _view(blob, off, n, dtype):
    """n elements of `dtype` at aligned byte offset `off` inside a uint8 blob tensor."""
    base = blob.data_ptr()
    al = (-base) % 256
    es = torch.empty(0, dtype=dtype).element_size()
    return blob[al + off: al + off + n * es].view(dtype)


class HipRun:
    def __init__(self, means3D, opacities, bg, view, proj, campos, W, H, tanx, tany, colors_precomp=None, shs=None,
                 sh_degree=0, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0, channels=3,
                 device="cuda", capacity=None):
        lib = _lib.raster()
        self.lib = lib
        dev = torch.device(device)
        self.dev, self.W, self.H, self.C = dev, W, H, channels
        self.means3D = _t(means3D, dev)
        P = self.P = 0 if self.means3D is None else self.means3D.shape[0]
        self.opac, self.bg = _t(opacities, dev), _t(bg, dev)
        self.view, self.proj, self.campos = _t(view, dev), _t(proj, dev), _t(campos, dev)
        self.colors, self.shs = _t(colors_precomp, dev), _t(shs, dev)
        self.scales, self.rots, self.cov = _t(scales, dev), _t(rotations, dev), _t(cov3D_precomp, dev)
        self.D, self.M = int(sh_degree), (0 if self.shs is None else self.shs.shape[1])
        self.mod, self.tanx, self.tany = float(scale_modifier), float(tanx), float(tany)
        u8 = dict(dtype=torch.uint8, device=dev)
        self.geom = torch.zeros(lib.fnx_geom_bytes(P, W, H), **u8)
        self.img = torch.zeros(lib.fnx_image_bytes(W, H), **u8)
        self.color = torch.zeros(channels, H, W, device=dev)
        self.depth = torch.zeros(1, H, W, device=dev)
        self.radii = torch.zeros(P, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.fnx_forward_stage1(channels, self.geom.data_ptr(), self.img.data_ptr(), P, self.D, self.M, W, H,
                                          _p(self.means3D), _p(self.shs), _p(self.colors), _p(self.opac),
                                          _p(self.scales), self.mod, _p(self.rots), _p(self.cov), _p(self.view),
                                          _p(self.proj), _p(self.campos), self.tanx, self.tany, 0, _p(self.radii), s))
        n = C.c_int(0)
        _lib.check(lib.fnx_read_num_rendered(self.img.data_ptr(), W, H, s, C.byref(n)))
        self.R = int(n.value)
        cap = self.R if capacity is None else capacity
        self.cap = cap
        self.binning = torch.zeros(lib.fnx_binning_bytes(cap), **u8)
        _lib.check(lib.fnx_forward_stage2(channels, self.geom.data_ptr(), self.binning.data_ptr(), cap,
                                          self.img.data_ptr(), P, W, H, _p(self.bg), _p(self.colors), _p(self.radii),
                                          self.color.data_ptr(), self.depth.data_ptr(), s))
        torch.cuda.synchronize()

    def status(self):
        return self.lib.fnx_read_status(self.img.data_ptr(), self.W, self.H, torch.cuda.current_stream().cuda_stream)

    def intermediates(self):
        P, W, H = self.P, self.W, self.H
        T = ((W + 15) // 16) * ((H + 15) // 16)
        g, im, b = _lib.geom_layout(P, W, H), _lib.image_layout(W, H), _lib.binning_layout(self.cap)
        out = dict(
            depths=_view(self.geom, g.depths, P, torch.float32), radii=self.radii,
            means2D=_view(self.geom, g.means2D, 2 * P, torch.float32).view(P, 2),
            cov3D=_view(self.geom, g.cov3D, 6 * P, torch.float32).view(P, 6),
            conic_opacity=_view(self.geom, g.conic_opacity, 4 * P, torch.float32).view(P, 4),
            rgb=_view(self.geom, g.rgb, 3 * P, torch.float32).view(P, 3),
            clamped=_view(self.geom, g.clamped, 3 * P, torch.uint8).view(P, 3),
            tiles_touched=_view(self.geom, g.tiles_touched, P, torch.int32),
            final_T=_view(self.img, im.final_T, H * W, torch.float32).view(H, W),
            n_contrib=_view(self.img, im.n_contrib, H * W, torch.int32).view(H, W),
            ranges=_view(self.img, im.ranges, 2 * T, torch.int32).view(T, 2),
            point_list=_view(self.binning, b.point_list, self.R, torch.int32),
            color=self.color, depth=self.depth)
        return {k: v.cpu().numpy() for k, v in out.items()}

    def backward(self, dL_dcolor, grad_splat_limit=-1, geometry_only=0):
        P, Cn, M, dev = self.P, self.C, self.M, self.dev
        z = lambda *s: torch.zeros(*s, device=dev)
        g = dict(dL_dmeans2D=z(P, 3), dL_dconic=z(P, 2, 2), dL_dopacity=z(P, 1), dL_dcolors=z(P, Cn),
                 dL_dmeans3D=z(P, 3), dL_dcov3D=z(P, 6), dL_dsh=z(P, M, 3), dL_dscales=z(P, 3), dL_drotations=z(P, 4))
        dL = _t(dL_dcolor, dev).reshape(Cn, self.H, self.W).contiguous()
        _lib.check(self.lib.fnx_rasterize_backward_ex(
            Cn, P, self.D, M, self.cap, _p(self.bg), self.W, self.H, _p(self.means3D), _p(self.shs), _p(self.colors),
            _p(self.scales), self.mod, _p(self.rots), _p(self.cov), _p(self.view), _p(self.proj), _p(self.campos),
            self.tanx, self.tany, _p(self.radii), self.geom.data_ptr(), _p(self.binning), self.img.data_ptr(),
            dL.data_ptr(), g["dL_dmeans2D"].data_ptr(), g["dL_dconic"].data_ptr(), g["dL_dopacity"].data_ptr(),
            g["dL_dcolors"].data_ptr(), g["dL_dmeans3D"].data_ptr(), g["dL_dcov3D"].data_ptr(), _p(g["dL_dsh"]),
            g["dL_dscales"].data_ptr(), g["dL_drotations"].data_ptr(), int(grad_splat_limit), int(geometry_only),
            torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in g.items()}


def scene_kwargs(g, cam, W, H, fov=0.8):
    tan = math.tan(fov * 0.5)
    return dict(means3D=g["means3D"], opacities=g["opacities"], view=cam.world_view_transform.cpu().numpy(),
                proj=cam.full_proj_transform.cpu().numpy(), campos=cam.camera_center.cpu().numpy(), W=W, H=H,
                tanx=tan, tany=tan)
