"""fnx_level2_activate / fnx_level2_backward (include/fnx_losses.h) against the torch expressions of the visual-particle
stage they replace (gm_dynamics.py getters, pipe_dynamics.py:88-148, train_visual_particle.py:161-194)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ORDER = ("color", "opacity", "scales", "rotation")
WIDTH = {"color": 1, "opacity": 1, "scales": 3, "rotation": 4}


def _raw(n, seed, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    raw = {k: torch.randn(n, WIDTH[k], generator=g).to(dev) for k in ORDER}
    raw["scales"] = raw["scales"] * 1.5 - 3.0  # ratios on both sides of the regulariser's threshold
    return raw


def _activated(raw):
    return dict(color=raw["color"].repeat(1, 3), opacity=torch.sigmoid(raw["opacity"]), scales=torch.exp(raw["scales"]),
                rotation=F.normalize(raw["rotation"]))


@pytest.mark.parametrize("n,n_prev,extra", [(1000, 700, 50), (257, 257, 0), (1, 0, 3), (5000, 4999, 1)])
def test_activate_and_backward_match_torch(n, n_prev, extra):
    from fluidnexus_amd.losses import level2_activate, level2_backward
    dev = torch.device("cuda:0")
    raw = _raw(n, 3 * n + 1, dev)
    if n > 10:
        raw["rotation"][3] = 0.0  # F.normalize's eps branch
    out = {k: torch.full((n + extra, 3 if k == "color" else WIDTH[k]), 7.0, device=dev) for k in ORDER}
    level2_activate(raw, out)
    ref = _activated(raw)
    for k in ORDER:
        torch.testing.assert_close(out[k][:n], ref[k], rtol=2e-6, atol=1e-7)
        assert (out[k][n:] == 7.0).all()  # the background rows are not touched

    gen = torch.Generator(device="cpu").manual_seed(n)
    prev = {k: (raw[k][:n_prev] + 0.1 * torch.randn(n_prev, WIDTH[k], generator=gen).to(dev)) for k in ORDER}
    g = {k: torch.randn(n + extra, 3 if k == "color" else WIDTH[k], generator=gen).to(dev) for k in ORDER}
    lam = dict(color=0.3, opacity=0.7, scales=1.1, rotation=0.05)
    lam_reg, thr, count, scale = 0.9, 3.0, 4.0, 0.25
    leaves = {k: raw[k].clone().requires_grad_() for k in ORDER}
    act = _activated(leaves)
    loss = sum((act[k] * g[k][:n]).sum() for k in ORDER)
    reg = 0.0
    if n_prev:
        reg = sum(lam[k] * F.mse_loss(leaves[k][:n_prev], prev[k]) for k in ORDER)
    sc = act["scales"]
    reg = reg + lam_reg * torch.clamp_min(sc.max(dim=1).values / sc.min(dim=1).values - thr, 0).mean()
    ((loss + count * reg) * scale).backward()
    d = {k: torch.full_like(raw[k], float("nan")) for k in ORDER}
    level2_backward(raw, prev, g, d, lam, lam_reg, thr, count, scale)
    for k in ORDER:
        torch.testing.assert_close(d[k], leaves[k].grad, rtol=2e-5, atol=2e-7)

    # an attribute that is not fitted is skipped
    d2 = {k: torch.zeros_like(raw[k]) for k in ("color", "scales")}
    level2_backward(raw, prev, g, d2, lam, 0.0, thr, count, scale)
    torch.testing.assert_close(d2["color"], d["color"], rtol=0, atol=0)
    assert not torch.equal(d2["scales"], d["scales"]) or n == 1


def test_arguments_are_checked():
    from fluidnexus_amd.losses import level2_activate
    dev = torch.device("cuda:0")
    raw = _raw(8, 0, dev)
    out = {k: torch.zeros(8, 3 if k == "color" else WIDTH[k], device=dev) for k in ORDER}
    out["scales"] = out["scales"][:4]
    with pytest.raises(RuntimeError, match="activated scales"):
        level2_activate(raw, out)
