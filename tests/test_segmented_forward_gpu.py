"""The segmented blend forward (opt-in: rasterizer.set_segmented_forward, fnx_raster_opts_t.segment_scratch; a measured
negative of round 5) against the one-list walk of the fast arithmetic: deep tiles cut into segments that are
blended at the same time from hinted transmittances, put together per pixel, pixels whose walk depended on the hint
blended again.  Every decision of the blend is the one-list walk's; what differs is the association of the transmittance
product: colours within the fast mode's tolerance, the positions' gradient within its gradient bound."""
import os
import sys
import types

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


def test_segmented_forward_matches_the_one_list_walk_over_moving_particles():
    from fluidnexus_amd import rasterizer
    from fluidnexus_amd.harness import build_smoke_frame
    from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
    from fluidnexus_amd.renderer.pipes import render_dynamics_views
    gm, cams = build_smoke_frame(n_views=2, size=512)
    gm.training_setup_current(types.SimpleNamespace(position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                                                    position_lr_delay_mult=0.01, position_lr_max_steps=30000))
    _, GRsetting, GRzer = get_render_pipe("render_dynamics")
    bg = torch.zeros(3, device="cuda")
    rasterizer.set_host_sync(False)
    rasterizer.set_blend_math("fast")
    torch.manual_seed(0)
    dimg = None
    try:
        def run(seg):
            nonlocal dimg
            rasterizer.set_segmented_forward(seg)
            gm.invalidate_caches()
            pkg = render_dynamics_views(cams, gm, None, bg, GRsetting=GRsetting, GRzer=GRzer, pos_type="guess_visual_nn",
                                        scale=True, screen_grad=False)
            if dimg is None:
                dimg = torch.randn_like(pkg["render"]) * 1e-3
            gm.optimizer.zero_grad()
            gm._estimate_xyz_nn.grad = None
            pkg["render"].backward(dimg)
            return pkg["render"].detach().clone(), pkg["depth"].detach().clone(), gm._estimate_xyz_nn.grad.detach().clone()

        cut = 0
        for call in range(5):  # call 0 has no hints (every segment starts from 1); the particles move between the calls
            a, b = run(False), run(True)
            torch.cuda.synchronize()
            rasterizer.check_status()
            d = (a[0] - b[0]).abs()
            assert int((d > 2e-5).sum()) <= max(2, int(1e-5 * d.numel())) and float(d.max()) <= 2e-3, (call, float(d.max()))
            g = (a[2] - b[2]).abs()
            assert float(g.max()) <= 1e-3 * float(a[2].abs().max()), (call, float(g.max()), float(a[2].abs().max()))
            # the median depth comes from the hinted walk where walk and truth agree on the segment T crosses 1/2 in
            assert int(((a[1] - b[1]).abs() > 0).sum()) <= int(2e-3 * a[1].numel()), call
            counters = rasterizer.segmented_forward_counters()
            cut = max(cut, max(c[2] for c in counters))
            gm.optimizer.step()
        assert cut > 10, f"the scene must have tiles that are cut ({cut})"
    finally:
        rasterizer.set_segmented_forward(False)
        rasterizer.set_blend_math("exact")
        rasterizer.set_host_sync(True)
