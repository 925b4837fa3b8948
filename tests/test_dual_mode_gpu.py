"""Dual mode of the static-split rasteriser (include/fnx_raster.h fnx_raster_dual_t, round 5): the 3-channel render of
fluid + background and the 1-channel render of the fluid alone out of ONE pass over the tile lists.  What it must equal is
the two rasterisers called separately -- ch3 (ch3/cuda_rasterizer/forward.cu:249-373 over all splats) and ch1 (the same
kernel with NUM_CHANNELS = 1 over the fluid splats): images bit for bit in the exact arithmetic, within the stated
tolerance in the fast one; the gradient of a loss on both images equal to the sum of the two separate backward passes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(seed=11, V=3, size=128, occluding=False):
    from fluidnexus_amd import harness as Hn
    gm, cams = Hn.build_smoke_frame(P_fluid=9000, P_background=4000, hidden_dims=(8, 20, 8), n_views=V, size=size,
                                    seed=seed, occluding=occluding)
    from types import SimpleNamespace
    gm.training_setup_current(SimpleNamespace(position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                                              position_lr_max_steps=30000))
    return gm, cams


def _renders(gm, cams, means3D, dual):
    from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
    from fluidnexus_amd.renderer.pipes import render_dynamics_views, render_fluid_views
    _, S3, Z3 = get_render_pipe("render_dynamics")
    _, S1, Z1 = get_render_pipe("render_fluid")
    bg = torch.tensor([0.15, 0.3, 0.45], device="cuda")
    n_fluid = means3D.shape[0] - gm.get_gs_xyz.shape[0]
    if dual:
        pkg = render_dynamics_views(cams, gm, None, bg, GRsetting=S3, GRzer=Z3, pos_type="guess_visual_nn", scale=True,
                                    means3D=means3D, screen_grad=False, dual_bg=bg[:1])
        return pkg["render"], pkg["depth"], pkg["render1"], pkg["depth1"]
    pkg = render_dynamics_views(cams, gm, None, bg, GRsetting=S3, GRzer=Z3, pos_type="guess_visual_nn", scale=True,
                                means3D=means3D, screen_grad=False)
    pkg1 = render_fluid_views(cams, gm, None, bg, GRsetting=S1, GRzer=Z1, pos_type="guess_visual_nn", scale=True,
                              means3D=means3D[:n_fluid], screen_grad=False)
    return pkg["render"], pkg["depth"], pkg1["render"], pkg1["depth"]


@pytest.mark.parametrize("math", ["exact", "fast"])
@pytest.mark.parametrize("occluding", [False, True])
def test_dual_mode_equals_the_two_rasterisers(math, occluding):
    from fluidnexus_amd import rasterizer
    from tests.test_fast_math_gpu import mixed_bound_report
    gm, cams = _scene(occluding=occluding)
    rasterizer.set_blend_math(math)
    try:
        base = gm.render_means_from_nn().detach().clone()
        rng = torch.Generator("cuda").manual_seed(3)
        V, H, W = len(cams), int(cams[0].image_height), int(cams[0].image_width)
        dL3 = torch.randn(V, 3, H, W, device="cuda", generator=rng)
        dL1 = torch.randn(V, 1, H, W, device="cuda", generator=rng)
        out = {}
        for dual in (False, True):
            m = base.clone().requires_grad_(True)
            c3, d3, c1, d1 = _renders(gm, cams, m, dual)
            g, = torch.autograd.grad([c3, c1], [m], grad_outputs=[dL3, dL1])
            torch.cuda.synchronize()
            out[dual] = (c3.detach(), d3.detach(), c1.detach(), d1.detach(), g)
            if dual:  # either image alone (the other's gradient is None -> zeros)
                m2 = base.clone().requires_grad_(True)
                c3b, _, c1b, _ = _renders(gm, cams, m2, True)
                g1, = torch.autograd.grad([c1b], [m2], grad_outputs=[dL1])
                out["only1"] = g1
        a, b = out[False], out[True]
        n_fluid = base.shape[0] - gm.get_gs_xyz.shape[0]
        # the 3-channel image does not notice the second one: bit for bit in both arithmetics
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        if math == "exact":
            assert torch.equal(a[2], b[2]), float((a[2] - b[2]).abs().max())
            assert torch.equal(a[3], b[3])
        else:  # the same fused arithmetic in both kernels; allow the stated tolerance anyway
            assert float((a[2] - b[2]).abs().max()) <= 2e-5
            assert int((a[3] != b[3]).sum()) <= max(2, int(1e-4 * a[3].numel()))
        assert float(b[2].abs().max()) > 0.01 and float(b[4][:n_fluid].abs().max()) > 0
        ok, msg = mixed_bound_report(a[4].cpu().numpy(), b[4].cpu().numpy())
        assert ok, msg
        assert float(b[4][n_fluid:].abs().max()) == 0.0  # the frozen background takes no gradient
        # second image alone against the 1-channel rasteriser's own backward
        m3 = base.clone().requires_grad_(True)
        _, _, c1s, _ = _renders(gm, cams, m3, False)
        g1s, = torch.autograd.grad([c1s], [m3], grad_outputs=[dL1])
        ok, msg = mixed_bound_report(g1s.cpu().numpy(), out["only1"].cpu().numpy())
        assert ok, msg
    finally:
        rasterizer.set_blend_math("exact")


def test_dual_mode_refuses_what_it_does_not_support():
    from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
    from fluidnexus_amd.renderer.pipes import render_dynamics_views
    gm, cams = _scene()
    _, S3, Z3 = get_render_pipe("render_dynamics")
    bg = torch.zeros(3, device="cuda")
    m = gm.render_means_from_nn().detach().clone().requires_grad_(True)
    with pytest.raises(ValueError):  # a screen-space gradient means the full backward
        render_dynamics_views(cams, gm, None, bg, GRsetting=S3, GRzer=Z3, pos_type="guess_visual_nn", scale=True, means3D=m,
                              screen_grad=True, dual_bg=bg[:1])


def test_dual_channel_loop_fused_equals_two_renders():
    """HotLoop(dual_channel=True): the fused iteration (one pass for both images) against the two-render iteration
    (FNX_DUAL_FUSED=0's path): same first Adam moments."""
    from fluidnexus_amd import harness as Hn
    res = {}
    for fused in (True, False):
        gm, cams = Hn.build_ball_frame(P_fluid=9000, P_background=4000, hidden_dims=(8, 20, 8), n_views=3, size=128, seed=3)
        cfg = dict(Hn.SMOKE, distance_threshold_visual=0.004)
        was = Hn._DUAL_FUSED
        Hn._DUAL_FUSED = fused
        try:
            loop = Hn.HotLoop(gm, cams, image_loss="fused", fused_physics=True, defer_visual_backward=True, capturable=True,
                              batched_views=True, fused_step=True, dual_channel=True, cfg=cfg)
            loop.make_targets()
            loop.iteration()
            torch.cuda.synchronize()
        finally:
            Hn._DUAL_FUSED = was
        st = gm.optimizer.state[gm._estimate_xyz_nn]
        res[fused] = st["exp_avg"].detach().clone()
    scale = float(res[False].abs().max())
    assert scale > 0 and float((res[True] - res[False]).abs().max()) <= 1e-3 * scale
