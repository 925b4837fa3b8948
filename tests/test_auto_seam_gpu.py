"""Library-side automation of the per-view plug-in seam (fluidnexus_amd.set_auto / FNX_AUTO=1, round 5): the reference's
per-view op sequence -- render_dynamics(camera, ...) per view, utils.loss_utils terms, loss.backward() per view,
cache_gradient_current, set_batch_gradient_current, torch.optim.Adam (train_physical_particle.py:329-432) -- must give the
same images and the same batch gradient with the automation on as with it off; only the launch sequence differs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _loop(auto):
    import fluidnexus_amd
    from fluidnexus_amd import harness as Hn, rasterizer
    from fluidnexus_amd.renderer import pipes
    fluidnexus_amd.set_auto(auto)
    pipes.set_static_split(True)
    rasterizer.set_host_sync(True)
    gm, cams = Hn.build_smoke_frame(P_fluid=8000, P_background=3000, hidden_dims=(8, 16, 8), n_views=3, size=128, seed=4)
    cfg = dict(Hn.SMOKE, distance_threshold_visual=0.004)
    loop = Hn.HotLoop(gm, cams, image_loss="torch", fused_physics=False, defer_visual_backward=False, cfg=cfg)
    loop.make_targets()
    return gm, cams, loop


def test_automated_seam_equals_the_plain_one():
    import fluidnexus_amd
    from fluidnexus_amd import rasterizer
    from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
    was = torch.backends.cudnn.enabled
    torch.backends.cudnn.enabled = False
    res = {}
    try:
        for auto in (False, True):
            gm, cams, loop = _loop(auto)
            rd, S_, Z_ = get_render_pipe("render_dynamics")
            with torch.no_grad():
                pkg = rd(cams[1], gm, None, loop.background, GRsetting=S_, GRzer=Z_, pos_type="guess_visual_nn", scale=True)
            keys = sorted(pkg.keys())
            shapes = {k: tuple(pkg[k].shape) for k in ("render", "radii", "depth", "viewspace_points", "render_xyz", "means3D")}
            loop.iteration()
            torch.cuda.synchronize()
            rasterizer.check_status()
            st = gm.optimizer.state[gm._estimate_xyz_nn]
            res[auto] = (keys, shapes, pkg["render"].clone(), pkg["radii"].clone(), pkg["depth"].clone(), st["exp_avg"].clone())
    finally:
        fluidnexus_amd.set_auto(False)
        rasterizer.set_host_sync(True)
        torch.backends.cudnn.enabled = was
    a, b = res[False], res[True]
    assert a[0] == b[0] and a[1] == b[1]
    # the rendered positions come from two summation orders of the same interpolation (per particle / cell by cell with the
    # deferred backward): 1 ulp apart on a few particles, hence pixels to ~1e-7; the integer state agrees
    assert torch.equal(a[3], b[3])
    assert float((a[2] - b[2]).abs().max()) <= 2e-6 and int((a[4] != b[4]).sum()) <= 2
    scale = float(a[5].abs().max())
    assert scale > 0 and float((a[5] - b[5]).abs().max()) <= 1e-3 * scale


def test_automation_leaves_other_stages_alone():
    """Attribute leaves (the visual-particle stage) or a missing background: the plain path, screen-space gradient and all."""
    import fluidnexus_amd
    from fluidnexus_amd.renderer import pipes
    gm, cams, loop = _loop(True)
    try:
        assert pipes._auto_applies(gm, "guess_visual_nn")
        gm._visual_opacity.requires_grad_(True)
        assert not pipes._auto_applies(gm, "guess_visual_nn")
        assert not pipes._auto_applies(gm, "hidden")
    finally:
        fluidnexus_amd.set_auto(False)
