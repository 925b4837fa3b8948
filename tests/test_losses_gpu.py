"""-m gpu: fused L1+SSIM HIP kernel vs golden vectors from the reference's loss_utils.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_utils.npz"))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_fused_l1_ssim_vs_reference(tag):
    from fluidnexus_amd.losses import fused_l1_ssim
    y = torch.tensor(G[f"y_{tag}"]).cuda()
    for which, key in ((0, "l1"), (1, "ssim")):
        x = torch.tensor(G[f"x_{tag}"]).cuda().requires_grad_(True)
        v = fused_l1_ssim(x, y)[which]
        v.backward()
        assert abs(v.item() - G[f"{key}_{tag}"]) < 3e-6, (key, tag, v.item(), G[f"{key}_{tag}"])
        ref = G[f"d{key}_{tag}"]
        err = np.abs(x.grad.cpu().numpy() - ref).max() / np.abs(ref).max()
        assert err < 2e-4, (key, tag, err)


@pytest.mark.parametrize("tag", ["a", "c"])
def test_fused_grey_image_loss_vs_reference(tag):
    """0.8 * L1 + 0.2 * (1 - SSIM) on grey-mean images (train_physical_particle.py:356-363,398-399)."""
    from fluidnexus_amd.losses import fused_l1_dssim_grey
    x = torch.tensor(G[f"x_{tag}"]).cuda().requires_grad_(True)
    y = torch.tensor(G[f"y_{tag}"]).cuda()
    l1, ds = fused_l1_dssim_grey(x, y)
    v = 0.8 * l1 + 0.2 * ds
    v.backward()
    assert abs(v.item() - G[f"grey_{tag}"]) < 3e-6
    ref = G[f"dgrey_{tag}"]
    assert np.abs(x.grad.cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-4


def test_pre_averaged_target_equals_the_per_pixel_mean():
    """grey = 2 (include/fnx_losses.h): the target's grey mean formed once (losses.grey_mean_target) instead of per pixel
    and iteration.  The forward sees the same bits (loss and per-image terms are equal); in the backward the compiler
    contracts `x - sum * (1/3)` into one FMA when it forms the mean itself (the kernels allow contraction, csrc/losses.hip),
    so the gradient agrees to rounding, not bit for bit."""
    from fluidnexus_amd.losses import image_loss_value_and_grad, grey_mean_target
    g = torch.Generator(device="cuda").manual_seed(3)
    img = torch.rand(3, 3, 150, 201, device="cuda", generator=g)
    gt = torch.rand(3, 3, 150, 201, device="cuda", generator=g)
    a = image_loss_value_and_grad(img, gt, 0.2, 1.0)
    b = image_loss_value_and_grad(img, grey_mean_target(gt), 0.2, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    scale = float(a[2].abs().max())
    assert float((a[2] - b[2]).abs().max()) <= 1e-6 * scale
    with pytest.raises(RuntimeError):
        image_loss_value_and_grad(img, grey_mean_target(gt), 0.2, 1.0, grey=False)
