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
