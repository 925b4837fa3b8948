"""Independent fp64 PyTorch-autograd formulation of the rasteriser maths -- TEST INFRASTRUCTURE ONLY.

Purpose: pin the hand-written backward of oracle/raster_oracle.c (a restatement of
ch3/cuda_rasterizer/backward.cu) against automatic differentiation of an independently written,
vectorised forward (restating forward.cu:70-145,148-244,249-373).  The discrete decisions
(tile lists, per-pixel contributor counts) are taken from the C oracle's forward; everything
continuous is recomputed here in float64 and differentiated by autograd.

Conventions reproduced from the reference so that autograd matches its analytic backward:
  * alpha = min(0.99, o*G) back-propagates as if unclamped (backward.cu:516);
  * the gradient of `means2D` is d loss / d pixel * (W/2, H/2) (backward.cu:444-445,524-525);
  * quaternions are used un-normalised (forward.cu:121).
Also usable as BASELINE.json config 0 ("PyTorch CPU autograd rasteriser") for small scenes.
"""
from __future__ import annotations

import numpy as np
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def _sh_rgb(deg, sh, dirs):
    """forward.cu:20-67 on [P,M,3] coefficients."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def render(fwd, means3D, means2D_dummy, opacities, scales=None, rotations=None, cov3D_precomp=None, colors=None,
           shs=None):
    """Differentiable fp64 render using the discrete structure of the C-oracle forward `fwd`.
    All tensor arguments are float64 leaves.  Returns color [C,H,W]."""
    i = fwd["_inputs"]
    W, H, C = fwd["W"], fwd["H"], fwd["C"]
    dd = dict(dtype=torch.float64)
    view = torch.tensor(i["viewmatrix"], **dd).reshape(4, 4)
    proj = torch.tensor(i["projmatrix"], **dd).reshape(4, 4)
    bg = torch.tensor(i["bg"], **dd)
    tanx, tany = float(np.float32(i["tan_fovx"])), float(np.float32(i["tan_fovy"]))
    fx, fy = W / (2.0 * tanx), H / (2.0 * tany)
    P = means3D.shape[0]
    ones = torch.ones(P, 1, **dd)
    hom = torch.cat([means3D, ones], 1)
    p_view = hom @ view  # row-vector convention (auxiliary.h:54-71)
    p_hom = hom @ proj
    p_w = 1.0 / (p_hom[:, 3:4] + 0.0000001)
    p_proj = p_hom[:, :3] * p_w + means2D_dummy
    # cov3D (forward.cu:113-145)
    if cov3D_precomp is None:
        mod = float(i["scale_modifier"])
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        Rm = torch.stack([
            1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
        # glm column-major constructor: the 9 values above fill columns => math matrix = transpose
        Rmath = Rm.transpose(1, 2)
        Smat = torch.diag_embed(mod * scales)
        Mm = Smat @ Rmath
        Sigma = Mm.transpose(1, 2) @ Mm
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                            1).reshape(P, 3, 3)
    # cov2D (forward.cu:70-108)
    t = p_view[:, :3]
    tz = t[:, 2]
    limx, limy = 1.3 * tanx, 1.3 * tany
    txc = torch.clamp(t[:, 0] / tz, -limx, limx) * tz
    tyc = torch.clamp(t[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * txc) / (tz * tz), zero, fy / tz, -(fy * tyc) / (tz * tz), zero, zero, zero],
                    1).reshape(P, 3, 3)  # rows of the math Jacobian
    Wr = view[:3, :3].T  # world->view rotation as a math matrix
    Tm = J @ Wr
    cov2 = Tm @ Sigma @ Tm.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c2 = cov2[:, 1, 1] + 0.3
    det = a * c2 - b * b
    con = torch.stack([c2 / det, -b / det, a / det], 1)
    px = ((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5
    if colors is None:
        campos = torch.tensor(i["campos"], **dd)
        d = means3D - campos
        d = d / d.norm(dim=1, keepdim=True)
        feats = _sh_rgb(fwd["D"], shs, d)
    else:
        feats = colors
    depth_v = p_view[:, 2]

    out = torch.zeros(C, H, W, **dd)
    gx = (W + 15) // 16
    ranges, plist, ncon = fwd["ranges"], fwd["point_list"].astype(np.int64), fwd["n_contrib"]
    for tile in range(ranges.shape[0]):
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        ty, tx = divmod(tile, gx)
        ys = torch.arange(ty * 16, min(ty * 16 + 16, H))
        xs = torch.arange(tx * 16, min(tx * 16 + 16, W))
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        yy, xx = yy.reshape(-1), xx.reshape(-1)
        npx = yy.numel()
        if r1 == r0:
            out[:, yy, xx] = bg[:C, None].expand(C, npx)
            continue
        ids = torch.from_numpy(plist[r0:r1])
        n = ids.numel()
        dx = px[ids][:, None] - xx.to(torch.float64)[None]
        dy = py[ids][:, None] - yy.to(torch.float64)[None]
        cn = con[ids]
        power = -0.5 * (cn[:, 0:1] * dx * dx + cn[:, 2:3] * dy * dy) - cn[:, 1:2] * dx * dy
        G = torch.exp(power)
        oG = opacities[ids].reshape(n, 1) * G
        alpha = oG + (torch.clamp_max(oG, 0.99) - oG).detach()  # cap back-propagates as identity
        last = torch.from_numpy(ncon[yy.numpy(), xx.numpy()].astype(np.int64))
        pos = torch.arange(n)[:, None]
        mask = (power <= 0) & (alpha >= 1.0 / 255.0) & (pos < last[None])
        a_eff = torch.where(mask, alpha, torch.zeros_like(alpha))
        one_m = 1.0 - a_eff
        Tcum = torch.cumprod(one_m, 0)
        Tbefore = torch.cat([torch.ones(1, npx, **dd), Tcum[:-1]], 0)
        wgt = a_eff * Tbefore
        col = feats[ids].T @ wgt  # [C, npx]
        out[:, yy, xx] = col + Tcum[-1][None] * bg[:C, None]
    return out


def gradients(fwd, dL_dcolor):
    """Autograd gradients in the layout of raster_oracle.backward()."""
    i = fwd["_inputs"]
    dd = dict(dtype=torch.float64)

    def leaf(a):
        return None if a is None else torch.tensor(a, **dd).requires_grad_(True)

    means3D = leaf(i["means3D"])
    dummy = torch.zeros(means3D.shape[0], 3, **dd, requires_grad=True)
    opac = leaf(i["opacities"])
    scales, rots, cov = leaf(i["scales"]), leaf(i["rotations"]), leaf(i["cov3D_precomp"])
    colors, shs = leaf(i["colors_precomp"]), leaf(i["shs"])
    img = render(fwd, means3D, dummy, opac, scales, rots, cov, colors, shs)
    loss = (img * torch.tensor(np.asarray(dL_dcolor), **dd).reshape(img.shape)).sum()
    loss.backward()

    def g(t):
        return None if t is None or t.grad is None else t.grad.numpy()

    return dict(color=img.detach().numpy(), dL_dmeans3D=g(means3D), dL_dmeans2D=g(dummy), dL_dopacity=g(opac),
                dL_dscales=g(scales), dL_drotations=g(rots), dL_dcov3D=g(cov), dL_dcolors=g(colors), dL_dsh=g(shs))
