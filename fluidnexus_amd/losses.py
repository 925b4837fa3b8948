"""Fused L1 + SSIM image loss (include/fnx_losses.h) as one autograd node.

`fused_l1_ssim(img, gt)` returns (l1_loss(img, gt), ssim(img, gt)) of FluidDynamics/utils/
loss_utils.py:9,33-64; `fused_l1_dssim_grey(img, gt)` returns (l1, 1 - ssim) of the grey-mean,
3x replicated images exactly as the physical stage forms them (train_physical_particle.py:356-363).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from ._lib import raw_stream as _lib_raw_stream

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
SYMBOLS = ("fnx_losses_abi_version", "fnx_losses_last_error", "fnx_l1_ssim_tiles", "fnx_l1_ssim_forward",
           "fnx_l1_ssim_backward", "fnx_l1_ssim_forward_batch", "fnx_l1_ssim_backward_batch",
           "fnx_image_loss_forward", "fnx_image_loss_backward", "fnx_image_loss_value_and_grad",
           "fnx_level2_activate", "fnx_level2_backward")


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("FNX_LOSSES_LIB") or os.path.join(_HERE, "libfnx_losses.so")  # env: developer variants
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build the HIP extension first (python -m fluidnexus_amd.build). "
                               "fluidnexus_amd has no CPU fallback.")
        L = C.CDLL(path)
        p, i = C.c_void_p, C.c_int
        L.fnx_losses_last_error.restype = C.c_char_p
        L.fnx_l1_ssim_tiles.argtypes = [i, i, i, i]
        L.fnx_l1_ssim_forward.argtypes = [p, p, i, i, i, i, p, p, p]
        L.fnx_l1_ssim_backward.argtypes = [p, p, i, i, i, i, p, p, p, p, p]
        L.fnx_l1_ssim_forward_batch.argtypes = [p, p, i, i, i, i, i, p, p, p]
        L.fnx_l1_ssim_backward_batch.argtypes = [p, p, i, i, i, i, i, p, p, p, p, p]
        f = C.c_float
        L.fnx_image_loss_forward.argtypes = [p, p, i, i, i, i, i, f, f, p, p, p, p, p]
        L.fnx_image_loss_backward.argtypes = [p, p, i, i, i, i, i, f, f, p, p, p, p]
        L.fnx_image_loss_value_and_grad.argtypes = [p, p, i, i, i, i, i, f, f, p, p, p, p, p, p, p]
        L.fnx_level2_activate.argtypes = [p, p, p, p, i, p, p, p, p, p]
        p4, f4 = C.c_void_p * 4, C.c_float * 4
        L.fnx_level2_backward.argtypes = [p4, p4, p4, p4, i, i, f4, f, f, f, f, p]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().fnx_losses_last_error().decode("utf-8", "replace"))


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt, grey):
        L = lib()
        if not img.is_cuda:
            raise RuntimeError("fluidnexus_amd losses: tensors must be on a HIP device (no CPU path)")
        img = img.float().contiguous()
        gt = gt.float().contiguous()
        if img.shape != gt.shape:
            raise RuntimeError(f"image {tuple(img.shape)} and target {tuple(gt.shape)} differ in shape")
        Cn, H, W = img.shape[-3:]
        batched = img.dim() == 4
        N = img.shape[0] if batched else 1
        Ce = 1 if grey else Cn
        nt = L.fnx_l1_ssim_tiles(Cn, H, W, int(grey))
        partials = torch.empty(N, nt, 2, dtype=torch.float32, device=img.device)
        dmaps = torch.empty(N, 3, Ce, H, W, dtype=torch.float32, device=img.device)
        s = _lib_raw_stream()
        _check(L.fnx_l1_ssim_forward_batch(img.data_ptr(), gt.data_ptr(), N, Cn, H, W, int(grey),
                                           partials.data_ptr(), dmaps.data_ptr(), s))
        sums = partials.sum(dim=1) / float(Ce * H * W)  # [N, 2]
        ctx.save_for_backward(img, gt, dmaps)
        ctx.grey = bool(grey)
        if batched:
            return sums[:, 0], sums[:, 1]
        return sums[0, 0], sums[0, 1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        L = lib()
        img, gt, dmaps = ctx.saved_tensors
        Cn, H, W = img.shape[-3:]
        N = img.shape[0] if img.dim() == 4 else 1
        g_l1 = g_l1.float().contiguous()
        g_ssim = g_ssim.float().contiguous()
        out = torch.empty_like(img)
        s = _lib_raw_stream()
        _check(L.fnx_l1_ssim_backward_batch(img.data_ptr(), gt.data_ptr(), N, Cn, H, W, int(ctx.grey),
                                            dmaps.data_ptr(), g_l1.data_ptr(), g_ssim.data_ptr(), out.data_ptr(), s))
        return out, None, None


def fused_l1_ssim(img, gt):
    """(mean |img - gt|, SSIM(img, gt)) for [C,H,W] images; [N,C,H,W] batches give per-image [N] vectors."""
    return _L1SSIM.apply(img, gt, False)


def fused_l1_dssim_grey(img, gt):
    """Physical-stage image terms: grey-mean both [3,H,W] images, then (L1, 1 - SSIM); [N,3,H,W] batches
    give per-image [N] vectors (one launch for all views of a training batch)."""
    l1, s = _L1SSIM.apply(img, gt, True)
    return l1, 1.0 - s


class _ImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt, w_l1, w_dssim, grey):
        L = lib()
        if not img.is_cuda:
            raise RuntimeError("fluidnexus_amd losses: tensors must be on a HIP device (no CPU path)")
        img = img.float().contiguous()
        gt = gt.float().contiguous()
        if img.shape != gt.shape or img.dim() != 4:
            raise RuntimeError(f"image {tuple(img.shape)} / target {tuple(gt.shape)}: expected equal [N,C,H,W] shapes")
        N, Cn, H, W = img.shape
        Ce = 1 if grey else Cn
        nt = L.fnx_l1_ssim_tiles(Cn, H, W, int(grey))
        scratch = torch.empty(N * nt * 2 + N * 2 + 1, dtype=torch.float32, device=img.device)
        partials, per_image, loss = scratch[:N * nt * 2], scratch[N * nt * 2:N * nt * 2 + 2 * N], scratch[-1:]
        dmaps = torch.empty(N, 3, Ce, H, W, dtype=torch.float32, device=img.device)
        _check(L.fnx_image_loss_forward(img.data_ptr(), gt.data_ptr(), N, Cn, H, W, int(grey), w_l1, w_dssim,
                                        partials.data_ptr(), dmaps.data_ptr(), per_image.data_ptr(), loss.data_ptr(),
                                        _lib_raw_stream()))
        ctx.save_for_backward(img, gt, dmaps)
        ctx.consts = (float(w_l1), float(w_dssim), bool(grey))
        per_image = per_image.view(N, 2)
        ctx.mark_non_differentiable(per_image)
        return loss.view(()), per_image

    @staticmethod
    def backward(ctx, g_loss, _g_per_image):
        L = lib()
        img, gt, dmaps = ctx.saved_tensors
        w_l1, w_dssim, grey = ctx.consts
        N, Cn, H, W = img.shape
        g = g_loss.float().contiguous()
        out = torch.empty_like(img)
        _check(L.fnx_image_loss_backward(img.data_ptr(), gt.data_ptr(), N, Cn, H, W, int(grey), w_l1, w_dssim,
                                         dmaps.data_ptr(), g.data_ptr(), out.data_ptr(),
                                         _lib_raw_stream()))
        return out, None, None, None, None


_ONE = {}


def image_loss_value_and_grad(img, gt, lambda_dssim, lambda_image=1.0, grey=True):
    """(loss, per_image [N,2], d loss / d img) of fused_image_loss without an autograd node: forward and backward
    kernels back to back (fnx_image_loss_value_and_grad: the reduction to the scalars rides inside the backward
    launch), seeded with a resident 1.0."""
    L = lib()
    if not img.is_cuda:
        raise RuntimeError("fluidnexus_amd losses: tensors must be on a HIP device (no CPU path)")
    img = img.float().contiguous()
    gt = gt.float().contiguous()
    # grey=True with a one-plane target [N,1,H,W]: the caller formed the target's grey mean once (grey_mean_target)
    pre = bool(grey) and img.dim() == 4 and gt.dim() == 4 and gt.shape[1] == 1 and img.shape[1] == 3 \
        and gt.shape[0] == img.shape[0] and gt.shape[2:] == img.shape[2:]
    if not pre and (img.shape != gt.shape or img.dim() != 4):
        raise RuntimeError(f"image {tuple(img.shape)} / target {tuple(gt.shape)}: expected equal [N,C,H,W] shapes")
    N, Cn, H, W = img.shape
    Ce = 1 if grey else Cn
    grey = 2 if pre else grey
    nt = L.fnx_l1_ssim_tiles(Cn, H, W, int(grey))
    scratch = torch.empty(N * nt * 2 + N * 2 + 1, dtype=torch.float32, device=img.device)
    partials, per_image, loss = scratch[:N * nt * 2], scratch[N * nt * 2:N * nt * 2 + 2 * N], scratch[-1:]
    dmaps = torch.empty(N, 3, Ce, H, W, dtype=torch.float32, device=img.device)
    one = _ONE.get(img.device)
    if one is None:
        one = _ONE[img.device] = torch.ones((), dtype=torch.float32, device=img.device)
    dimg = torch.empty_like(img)
    w_l1, w_dssim = (1.0 - float(lambda_dssim)) * float(lambda_image), float(lambda_dssim) * float(lambda_image)
    _check(L.fnx_image_loss_value_and_grad(img.data_ptr(), gt.data_ptr(), N, Cn, H, W, int(grey), w_l1, w_dssim,
                                           partials.data_ptr(), dmaps.data_ptr(), per_image.data_ptr(), loss.data_ptr(),
                                           one.data_ptr(), dimg.data_ptr(), _lib_raw_stream()))
    return loss.view(()), per_image.view(N, 2), dimg


def grey_mean_target(gt):
    """[N,3,H,W] -> [N,1,H,W]: the grey mean of a target batch exactly as the loss kernels form it per pixel
    (((r + g) + b) * fp32(1/3)), to be computed ONCE per frame and handed to image_loss_value_and_grad(grey=True)."""
    gt = gt.float()
    return (((gt[:, 0] + gt[:, 1]) + gt[:, 2]) * torch.tensor(1.0 / 3.0, dtype=torch.float32, device=gt.device)).unsqueeze(1).contiguous()


def fused_image_loss(img, gt, lambda_dssim, lambda_image=1.0, grey=True):
    """Image term of a whole training batch as one scalar:
    sum_n ((1 - lambda_dssim) * L1_n + lambda_dssim * (1 - SSIM_n)) * lambda_image over the [N,3,H,W] batch
    (train_physical_particle.py:356-366; grey: on the grey-mean images as the physical stage forms them).
    Returns (loss, per_image) with per_image [N,2] = detached (L1_n, SSIM_n) for logging.  Three kernels."""
    return _ImageLoss.apply(img, gt, (1.0 - float(lambda_dssim)) * float(lambda_image),
                            float(lambda_dssim) * float(lambda_image), bool(grey))


_L2_ORDER = ("color", "opacity", "scales", "rotation")
_L2_WIDTH = {"color": 1, "opacity": 1, "scales": 3, "rotation": 4}


def _l2_ptr(t, rows, width, what):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 2 and t.shape[0] >= rows
            and t.shape[1] == width):
        raise RuntimeError(f"level-two {what}: expected a contiguous float32 device tensor [>={rows}, {width}], got "
                           f"{tuple(t.shape)} {t.dtype} on {t.device}")
    return t.data_ptr()


def level2_activate(raw, out):
    """Visual-particle stage: activate the raw attributes of the n fluid Gaussians into the first n rows of the arrays
    the rasteriser reads (gm_dynamics.py getters; pipe_dynamics.py:88-148).  raw / out: dicts keyed colour, opacity,
    scales, rotation; out["color"] has 3 columns (the grey colour repeated)."""
    n = raw["color"].shape[0]
    args = [_l2_ptr(raw[k], n, _L2_WIDTH[k], f"raw {k}") for k in _L2_ORDER]
    args += [_l2_ptr(out[k], n, 3 if k == "color" else _L2_WIDTH[k], f"activated {k}") for k in _L2_ORDER]
    _check(lib().fnx_level2_activate(*args[:4], n, *args[4:], _lib_raw_stream()))


def level2_backward(raw, prev, g, d, lambdas, lambda_reg, reg_threshold, reg_count, scale):
    """Gradients of the raw attributes (d[k], None = not fitted) from the rasteriser's gradients with respect to the
    activated rows (g[k]) plus reg_count x the view-independent terms of train_visual_particle.py:161-194, times scale
    (include/fnx_losses.h)."""
    n = raw["color"].shape[0]
    n_prev = prev["color"].shape[0]
    p4, f4 = C.c_void_p * 4, C.c_float * 4
    lib().fnx_level2_backward.restype = C.c_int
    _check(lib().fnx_level2_backward(
        p4(*[_l2_ptr(raw[k], n, _L2_WIDTH[k], f"raw {k}") for k in _L2_ORDER]),
        p4(*[_l2_ptr(prev[k], n_prev, _L2_WIDTH[k], f"previous {k}") for k in _L2_ORDER]),
        p4(*[_l2_ptr(g[k], n, 3 if k == "color" else _L2_WIDTH[k], f"gradient {k}") for k in _L2_ORDER]),
        p4(*[_l2_ptr(d.get(k), n, _L2_WIDTH[k], f"output {k}") for k in _L2_ORDER]),
        n, n_prev, f4(*[float(lambdas[k]) for k in _L2_ORDER]), float(lambda_reg), float(reg_threshold), float(reg_count),
        float(scale), _lib_raw_stream()))
