"""-m gpu: the loops behind bench.py's configurations at reduced size, each checked against the reference's own op
sequence written with the per-view pipes and autograd:

  config 2  FirstFrameLoop(render_fluid, 1 channel): positions of the visual particles are the leaf
            (entries_scalar_real/train_physical_particle.py:97-165)
  first frame of the dynamics scenes  FirstFrameLoop(render_dynamics): grey-mean image term, static background
            (entries_fluid_nexus/train_physical_particle.py:103-163)
  config 5  HotLoop(dual_channel): every view through the 3-channel (fluid + background) and the 1-channel (fluid)
            rasteriser, both image terms + distance + physics terms on the same hidden-particle leaf

"Checked" = after one optimiser step Adam's first moment is (1 - beta1) x the batch gradient: it must equal the
gradient autograd gives for the sum of the per-view losses built from render_* per view, utils.loss_utils and the
model's differentiable getters."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _first_moment(gm, param):
    return gm.optimizer.state[param]["exp_avg"].detach().clone()


def _close(a, b, rtol):
    scale = b.abs().max().item()
    assert scale > 0
    return (a - b).abs().max().item() <= rtol * scale


@pytest.mark.parametrize("pipe", ["render_fluid", "render_dynamics"])
def test_first_frame_loop_matches_per_view_autograd(pipe):
    from fluidnexus_amd import harness as Hn
    from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
    from fluidnexus_amd.utils.loss_utils import distance_loss, l1_loss, ssim
    V, size = 3, 128
    if pipe == "render_fluid":
        gm, cams = Hn.build_scalar_real_frame(P=8000, n_views=V, size=size, seed=2)
    else:
        gm, cams = Hn.build_smoke_frame(P_fluid=8000, P_background=3000, hidden_dims=(6, 12, 6), n_views=V, size=size, seed=2)
        gm._visual_xyz = gm._visual_xyz / gm.scale_factor  # first-frame stage: world units (detach_visual_and_scale comes later)
    cfg = dict(Hn.SCALAR_REAL, distance_threshold_visual=0.004)
    loop = Hn.FirstFrameLoop(gm, cams, rd_pipe=pipe, cfg=cfg, capturable=False)
    loop.make_targets()
    x0 = gm._visual_xyz.detach().clone()

    # reference op sequence: per view render -> (grey-mean) L1 + D-SSIM + distance loss, summed, autograd
    render_func, GRsetting, GRzer = get_render_pipe(pipe)
    x = x0.clone().requires_grad_(True)
    keep = gm._visual_xyz
    gm._visual_xyz = x
    total = 0.0
    for cam in cams:
        pkg = render_func(cam, gm, None, loop.background, GRsetting=GRsetting, GRzer=GRzer, pos_type="visual")
        image, gt = pkg["render"], cam.original_image
        if pipe == "render_dynamics":
            gt = torch.cat([torch.mean(gt, dim=0, keepdim=True)] * 3, dim=0)
            image = torch.cat([torch.mean(image, dim=0, keepdim=True)] * 3, dim=0)
        total = total + (1.0 - cfg["lambda_dssim"]) * l1_loss(image, gt) + cfg["lambda_dssim"] * (1.0 - ssim(image, gt))
        total = total + cfg["lambda_first_distance"] * distance_loss(gm.get_visual_xyz, cfg["distance_threshold_visual"])
    total.backward()
    want = x.grad / V
    gm._visual_xyz = keep
    assert float(want.abs().max()) > 0

    loop.iteration()
    torch.cuda.synchronize()
    got = _first_moment(gm, gm._visual_xyz) / (1.0 - 0.9)
    assert _close(got, want, 5e-4)
    # the step moved the particles, and a few more iterations reduce the image error
    assert float((gm._visual_xyz.detach() - x0).abs().max()) > 0


def test_first_frame_loop_graph_equals_eager():
    from fluidnexus_amd import harness as Hn
    from fluidnexus_amd import rasterizer
    res = {}
    rasterizer.set_host_sync(False)
    try:
        for mode in ("eager", "graph"):
            gm, cams = Hn.build_scalar_real_frame(P=6000, n_views=2, size=96, seed=5)
            loop = Hn.FirstFrameLoop(gm, cams, cfg=dict(Hn.SCALAR_REAL, distance_threshold_visual=0.004))
            loop.make_targets()
            loop.iteration()  # seeds the binning capacity
            rasterizer.check_status()
            if mode == "graph":
                loop.capture(warmup=1, iterations=2)  # 1 + 1 eager, then 2 per replay
                loop.iteration()
                loop.iteration()
                assert loop.iterations_per_call == 2
            else:
                for _ in range(5):
                    loop.iteration()
            torch.cuda.synchronize()
            rasterizer.check_status()
            res[mode] = (gm._visual_xyz.detach().clone(), float(gm.optimizer.state[gm._visual_xyz]["step"]))
    finally:
        rasterizer.set_host_sync(True)
    assert res["eager"][1] == res["graph"][1] == 6.0
    # identical launch sequences; the blend backward's atomics may reorder fp32 sums
    assert (res["eager"][0] - res["graph"][0]).abs().max().item() < 2e-5


def test_dual_channel_hot_loop_matches_per_view_autograd():
    """Config 5's iteration on a small ball scene against the sum of the per-view losses of both rasterisers."""
    from fluidnexus_amd import harness as Hn
    from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
    from fluidnexus_amd.utils.loss_utils import distance_loss, l1_loss, l2_loss, ssim
    V, size = 3, 128
    gm, cams = Hn.build_ball_frame(P_fluid=9000, P_background=4000, hidden_dims=(8, 20, 8), n_views=V, size=size, seed=3)
    cfg = dict(Hn.SMOKE, distance_threshold_visual=0.004)
    loop = Hn.HotLoop(gm, cams, image_loss="fused", fused_physics=True, defer_visual_backward=True, capturable=True,
                      batched_views=True, fused_step=True, dual_channel=True, cfg=cfg)
    loop.make_targets()
    assert cams[0].original_image.shape[0] == 3 and cams[0].original_image_ch1.shape[0] == 1
    x0 = gm._estimate_xyz_nn.detach().clone()

    rd3, S3, Z3 = get_render_pipe("render_dynamics")
    rd1, S1, Z1 = get_render_pipe("render_fluid")
    keep, share, defer = gm._estimate_xyz_nn, gm.share_visual_output, gm.defer_visual_backward
    x = torch.nn.Parameter(x0.clone())
    gm._estimate_xyz_nn, gm.share_visual_output, gm.defer_visual_backward = x, False, False
    gm.invalidate_caches()
    c = cfg
    total = 0.0
    for cam in cams:
        p3 = rd3(cam, gm, None, loop.background, GRsetting=S3, GRzer=Z3, pos_type="guess_visual_nn", scale=True)
        gt = torch.cat([torch.mean(cam.original_image, dim=0, keepdim=True)] * 3, dim=0)
        im = torch.cat([torch.mean(p3["render"], dim=0, keepdim=True)] * 3, dim=0)
        total = total + ((1 - c["lambda_dssim"]) * l1_loss(im, gt) + c["lambda_dssim"] * (1 - ssim(im, gt))) * c["lambda_image"]
        p1 = rd1(cam, gm, None, loop.background, GRsetting=S1, GRzer=Z1, pos_type="guess_visual_nn", scale=True)
        total = total + ((1 - c["lambda_dssim"]) * l1_loss(p1["render"], cam.original_image_ch1)
                         + c["lambda_dssim"] * (1 - ssim(p1["render"], cam.original_image_ch1))) * c["lambda_image"]
        total = total + c["lambda_current_distance"] * distance_loss(p3["render_xyz"], c["distance_threshold_visual"])
        total = total + c["lambda_exyz"] * l2_loss(x * gm.scale_factor, gm._estimate_xyz)
        pr = gm.get_gas_constraints_from_exyz_nn()
        total = total + c["lambda_gas_constraints"] * l2_loss(pr, torch.ones_like(pr))
        pn = gm.get_gas_constraints_from_vel_nn_guess()
        total = total + c["lambda_next_gas_constraints"] * l2_loss(pn, torch.ones_like(pn))
    total.backward()
    want = x.grad / V
    gm._estimate_xyz_nn, gm.share_visual_output, gm.defer_visual_backward = keep, share, defer
    gm.invalidate_caches()

    loop.iteration()
    torch.cuda.synchronize()
    got = _first_moment(gm, gm._estimate_xyz_nn) / (1.0 - 0.9)
    assert _close(got, want, 1e-3)
    loop.log_scalars = True
    for i in range(3):
        loop.iteration()
    assert np.isfinite(loop.last["total"]) and np.isfinite(float(gm._estimate_xyz_nn.detach().abs().max()))


def test_level_two_batched_matches_per_view_loop():
    """The view-batched visual-particle stage (one render / loss / backward for the batch, static background binned
    once, regularisers counted once per view) against the per-view loop of train_visual_particle.py:133-222: same
    first moments after one step for all four attribute groups."""
    from fluidnexus_amd import harness as Hn
    res = {}
    for mode in ("per_view", "batched", "fused", "fused_device_adam"):
        gm, cams = Hn.build_smoke_frame(P_fluid=12000, P_background=4000, hidden_dims=(6, 10, 6), n_views=3, size=128, seed=4)
        cfg = dict(Hn.SMOKE_L2, lambda_reg_scaling=0.05, scaling_reg_ratio_threshold=1.2)
        loop = Hn.HotLoopLevelTwo(gm, cams, cfg=cfg, batched_views=mode != "per_view", fused_attributes=mode.startswith("fused"),
                                  capturable=mode == "fused_device_adam")
        loop.make_targets()
        for n in gm._L2:  # away from the previous frame's values, so that the consistency terms have a gradient
            with torch.no_grad():
                p = getattr(gm, f"_visual_{n}")
                p.add_(0.01 * torch.randn(p.shape, device=p.device, generator=torch.Generator(p.device).manual_seed(7)))
        loop.iteration()
        torch.cuda.synchronize()
        res[mode] = {n: _first_moment(gm, getattr(gm, f"_visual_{n}")) for n in gm._L2}
    for n in res["per_view"]:
        assert _close(res["batched"][n], res["per_view"][n], 1e-3), n
        # the attribute kernels (fnx_level2_activate / fnx_level2_backward) against the same chain in torch ops
        assert _close(res["fused"][n], res["batched"][n], 1e-4), n
        # ... and with fnx_adam_step on the torch optimiser's state instead of torch's fused Adam
        assert _close(res["fused_device_adam"][n], res["fused"][n], 1e-6), n


def test_level_two_graph_equals_eager():
    from fluidnexus_amd import harness as Hn
    from fluidnexus_amd import rasterizer
    res = {}
    rasterizer.set_host_sync(False)
    try:
        for mode in ("eager", "graph"):
            gm, cams = Hn.build_smoke_frame(P_fluid=10000, P_background=3000, hidden_dims=(6, 10, 6), n_views=2, size=96, seed=6)
            loop = Hn.HotLoopLevelTwo(gm, cams, batched_views=True, capturable=True)
            loop.make_targets()
            loop.iteration()
            rasterizer.check_status()
            if mode == "graph":
                loop.capture(warmup=1, iterations=2)
                loop.iteration()
                loop.iteration()
            else:
                for _ in range(5):
                    loop.iteration()
            torch.cuda.synchronize()
            rasterizer.check_status()
            res[mode] = {n: getattr(gm, f"_visual_{n}").detach().clone() for n in gm._L2}
            assert float(gm.optimizer.state[gm._visual_color]["step"]) == 6.0
    finally:
        rasterizer.set_host_sync(True)
    for n in res["eager"]:
        moved = (res["eager"][n] - loop.prev[n]).abs().max().item()
        assert moved > 0
        # Adam (eps = 1e-15) turns a gradient that is pure summation noise (e.g. the radial component of a quaternion)
        # into steps of +-lr: a handful of such elements may differ between two runs of the same loop
        off = ((res["eager"][n] - res["graph"][n]).abs() > 0.02 * moved + 1e-6).sum().item()
        assert off <= max(2, res["eager"][n].numel() // 2000), (n, off)


def test_config1_reference_cpu_case_end_to_end(oracle):
    """BASELINE configs[0] / SURVEY 8(d) config 1 -- the one case the reference can run on a CPU: 10 000 random
    Gaussians, one 256 x 256 view (FoV 0.8 rad, camera at distance 2 on +z), ch3, loss = L1 + 0.2 (1 - SSIM) against a
    seeded U(0,1) target.  Through the plug-in API (`get_render_pipe` -> GaussianRasterizer, autograd) and the fused
    image loss: the image is bit-identical to the CPU oracle's, the loss equals utils.loss_utils' (the reference's
    functions, golden-pinned in test_golden_utils.py) on that image, and the gradients of all five inputs equal the
    oracle's backward seeded with that loss's dL/dimage."""
    from fluidnexus_amd import synthetic as S
    from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
    from fluidnexus_amd.losses import fused_l1_ssim
    from fluidnexus_amd.utils.loss_utils import l1_loss, ssim
    from tests.hip_harness import scene_kwargs
    P, W, H = 10_000, 256, 256
    dev = torch.device("cuda:0")
    g = S.random_gaussians(P, seed=0)  # xyz U([-0.5, 0.5]^3), log-scale U(-5.5, -3.5), quat N(0,1) normalised, ...
    cam = S.front_camera(W, H, device="cpu")
    kw = scene_kwargs(g, cam, W, H, 0.8)
    bg = np.zeros(3, np.float32)
    target = np.random.RandomState(0).uniform(0, 1, size=(3, H, W)).astype(np.float32)

    f = oracle.forward(kw["means3D"], kw["opacities"], bg, kw["view"], kw["proj"], kw["campos"], W, H, kw["tanx"], kw["tany"],
                       colors_precomp=g["colors"], scales=g["scales"], rotations=g["rotations"])
    img_ref = torch.tensor(f["color"], requires_grad=True)
    loss_ref = 0.8 * l1_loss(img_ref, torch.tensor(target)) + 0.2 * (1.0 - ssim(img_ref, torch.tensor(target)))
    loss_ref.backward()
    grads_ref = oracle.backward(f, img_ref.grad.numpy())

    _, Settings, Rasterizer = get_render_pipe("render_dynamics")
    t = lambda a: torch.tensor(a, device=dev)  # noqa: E731
    leaves = {k: t(g[k]).requires_grad_() for k in ("means3D", "opacities", "scales", "rotations", "colors")}
    screen = torch.zeros(P, 3, device=dev, requires_grad=True)
    rs = Settings(image_height=H, image_width=W, tan_fov_x=kw["tanx"], tan_fov_y=kw["tany"], bg=t(bg), scale_modifier=1.0,
                  view_matrix=t(kw["view"]), proj_matrix=t(kw["proj"]), sh_degree=0, campos=t(kw["campos"]), prefiltered=False)
    image, radii, depth = Rasterizer(raster_settings=rs)(
        means3D=leaves["means3D"], means2D=screen, opacities=leaves["opacities"], colors_precomp=leaves["colors"],
        scales=leaves["scales"], rotations=leaves["rotations"])
    assert torch.equal(image.detach().cpu().view(torch.int32), torch.tensor(f["color"]).view(torch.int32))
    assert torch.equal(radii.cpu(), torch.tensor(f["radii"]))
    l1_value, ssim_value = fused_l1_ssim(image, t(target))
    loss = 0.8 * l1_value + 0.2 * (1.0 - ssim_value)
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    loss.backward()
    names = dict(means3D="dL_dmeans3D", opacities="dL_dopacity", scales="dL_dscales", rotations="dL_drotations",
                 colors="dL_dcolors")
    for k, rk in names.items():
        ref = torch.tensor(grads_ref[rk]).reshape(leaves[k].shape)
        assert _close(leaves[k].grad.cpu(), ref, 5e-4), k
    assert _close(screen.grad.cpu(), torch.tensor(grads_ref["dL_dmeans2D"]), 5e-4)


def test_level_two_one_rank_process_group_matches_single_process():
    """The visual-particle stage with a process group (views sharded, ONE all-reduce of the flat attribute gradient,
    then the per-group Adam kernels) on a 1-rank RCCL group against the same loop without communication."""
    import os
    import torch.distributed as dist
    from fluidnexus_amd import harness as Hn
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/fnx_rccl_test_%h_%p.log")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    res = {}
    try:
        for use_dist in (False, True):
            gm, cams = Hn.build_smoke_frame(P_fluid=8000, P_background=3000, hidden_dims=(6, 10, 6), n_views=3, size=96, seed=9)
            loop = Hn.HotLoopLevelTwo(gm, cams, batched_views=True, capturable=True, force_all_reduce=use_dist)
            loop.make_targets()
            for n in gm._L2:  # anisotropic scales (else the rotation gradient is identically zero), consistency terms active
                with torch.no_grad():
                    p = getattr(gm, f"_visual_{n}")
                    p.add_(0.01 * torch.randn(p.shape, device=p.device, generator=torch.Generator(p.device).manual_seed(11)))
            loop.iteration()
            torch.cuda.synchronize()
            grads = torch.cat([getattr(gm, f"_visual_{n}").grad.flatten() for n in gm._L2])
            assert loop._flat_grad is not None and torch.equal(grads, loop._flat_grad)  # one buffer behind the four .grad
            res[use_dist] = {n: _first_moment(gm, getattr(gm, f"_visual_{n}")) for n in gm._L2}
    finally:
        if created:
            dist.destroy_process_group()
    for n in res[False]:
        assert _close(res[True][n], res[False][n], 1e-4), n
