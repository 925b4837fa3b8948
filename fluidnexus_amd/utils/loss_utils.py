"""Image / particle losses with the semantics of FluidDynamics/utils/loss_utils.py
(l1_loss :9, l2_loss :13, ssim :33-64, distance_loss :98-121, l2_loss_consistency :140-147).
Pinned by tests/golden/loss_utils.npz.  `fused_image_loss` (HIP) is the hot-loop variant."""
from __future__ import annotations

import math
from functools import lru_cache

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt):
    return (network_output - gt).abs().mean()


def l2_loss(network_output, gt):
    d = network_output - gt
    return (d * d).mean()


def relative_loss(network_output, gt):
    return ((network_output - gt) / (gt + 0.001)).abs().mean()


@lru_cache(maxsize=None)
def _window_1d(window_size: int, sigma: float):
    half = window_size // 2
    g = torch.tensor([math.exp(-((i - half) ** 2) / float(2 * sigma ** 2)) for i in range(window_size)])
    return g / g.sum()


def create_window(window_size, channel):
    """[channel, 1, k, k] normalised Gaussian (sigma 1.5) for a depthwise convolution."""
    g = _window_1d(window_size, 1.5).unsqueeze(1)
    w2 = (g @ g.t()).float()[None, None]
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def _ssim_map(img1, img2, window, pad, channel):
    def blur(x):
        return F.conv2d(x, window, padding=pad, groups=channel)

    mu1, mu2 = blur(img1), blur(img2)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = blur(img1 * img1) - mu1_sq
    s2 = blur(img2 * img2) - mu2_sq
    s12 = blur(img1 * img2) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))


def ssim(img1, img2, window_size=11, size_average=True):
    """loss_utils.py:33-64.  With the per-view automation on (fluidnexus_amd.set_auto), device tensors of the default form (11-tap window, mean over the image, [C,H,W] or [N,C,H,W])
    go through the fused kernel (fluidnexus_amd.losses.fused_l1_ssim: separable windows in LDS, one launch forward and one
    backward instead of ~25 + ~40 -- pinned against this very expression by tests/golden/loss_utils.npz); anything else
    takes the reference's own op sequence below."""
    if img1.is_cuda and img2.is_cuda and window_size == 11 and size_average and img1.dim() == 3 and img1.shape == img2.shape \
            and img1.dtype == torch.float32:
        from .. import auto_enabled
        if auto_enabled():  # part of the per-view seam's automation (fluidnexus_amd.set_auto / FNX_AUTO=1)
            from ..losses import fused_l1_ssim
            return fused_l1_ssim(img1, img2)[1]
    channel = img1.size(-3)
    window = create_window(window_size, channel).to(device=img1.device, dtype=img1.dtype)
    m = _ssim_map(img1, img2, window, window_size // 2, channel)
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def distance_loss(positions, threshold):
    """Penalise particle pairs closer than `threshold` (loss_utils.py:98-121).  Device tensors go through the
    radius-limited fused kernel (fluidnexus_amd.physics.distance_loss: hash grid with cell = threshold, value and
    gradient in one launch sequence, O(N) memory); the dense N x N torch form below is the reference's own
    expression, kept for host tensors (small N only: 300k points would need 360 GB)."""
    if positions.is_cuda:
        from ..physics import distance_loss as _fused
        return _fused(positions, threshold)
    d = torch.cdist(positions, positions, p=2)
    mask = d < threshold
    mask.fill_diagonal_(False)
    return ((threshold - d) * mask.float()).clamp(min=0).pow(2).sum()


def l2_loss_consistency(predictions, prev_predictions, threshold=0.0):
    """MSE between the first len(prev) current particles and the previous frame's (loss_utils.py:140-147)."""
    if prev_predictions is None:
        return torch.zeros(1, device=predictions.device)
    n_prev = prev_predictions.shape[0]
    assert predictions.shape[0] >= n_prev, "Current number of particles must be greater than or equal to the previous"
    return F.mse_loss(predictions[:n_prev], prev_predictions)
