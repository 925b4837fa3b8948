"""-m gpu: the view-batched entry points (fnx_*_views, GaussianRasterizerViews) against the single-view
ones.  Forward: every view's slice must be BIT-IDENTICAL to a single-view call with that camera.
Backward: per-view screen-space gradients and the sums over the views agree with the per-view calls
within fp32 atomic-order tolerance (the blend backward accumulates with atomics in both paths)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fluidnexus_amd import synthetic as S  # noqa: E402


def _settings(cams, W, H, bg, fovs, sh_degree=0):
    from fluidnexus_amd.rasterizer import GaussianRasterizationSettings
    return [GaussianRasterizationSettings(image_height=H, image_width=W, tan_fov_x=math.tan(f * 0.5),
                                          tan_fov_y=math.tan(f * 0.5), bg=bg, scale_modifier=1.0,
                                          view_matrix=c.world_view_transform, proj_matrix=c.full_proj_transform,
                                          sh_degree=sh_degree, campos=c.camera_center, prefiltered=False)
            for c, f in zip(cams, fovs)]


def _leaves(g, names, dev):
    out = {}
    for n in names:
        out[n] = torch.tensor(g[n], dtype=torch.float32, device=dev, requires_grad=True)
    return out


def _close(a, b, rtol=2e-4):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = b.abs().max().item() + 1e-20
    return (a - b).abs().max().item() / scale < rtol


@pytest.mark.parametrize("channels,mode", [(3, "precomp"), (3, "sh"), (1, "precomp"), (3, "limit")])
def test_views_match_single_view_calls(channels, mode):
    from fluidnexus_amd.rasterizer import GaussianRasterizer, GaussianRasterizerViews
    dev = torch.device("cuda")
    P, W, H, V = 6000, 144, 112, 4
    g = S.random_gaussians(P, seed=11, box=0.45, log_scale=(-4.8, -2.6), channels=channels,
                           center=(0.34, 0.3, -0.225))
    fovs = [0.8, 0.7, 0.9, 0.8]
    cams = []
    for k, c in enumerate(S.arc_cameras(V, W, H, device="cuda", fov=0.8)):
        cams.append(c)
    bg = torch.tensor([0.2, 0.5, 0.1], device=dev)
    rng = np.random.RandomState(3)
    names = ["means3D", "opacities", "scales", "rotations"]
    sh_degree = 0
    if mode == "sh":
        sh_degree = 3
        g["shs"] = rng.normal(scale=0.4, size=(P, 16, 3)).astype(np.float32)
        names.append("shs")
    else:
        names.append("colors")
    settings = _settings(cams, W, H, bg, fovs, sh_degree)
    dL = torch.tensor(rng.normal(size=(V, channels, H, W)).astype(np.float32), device=dev)
    limit = 4000 if mode == "limit" else None

    def kwargs(L):
        kw = dict(means3D=L["means3D"], opacities=L["opacities"].reshape(P, 1), scales=L["scales"],
                  rotations=L["rotations"])
        if mode == "sh":
            kw.update(shs=L["shs"])
        else:
            kw.update(colors_precomp=L["colors"])
        return kw

    # single-view reference calls
    A = _leaves(g, names, dev)
    if mode == "limit":  # appearance is constant: geometry-only gradients
        for n in ("opacities", "colors"):
            A[n].requires_grad_(False)
    imgs, radii, depths, screens = [], [], [], []
    for v in range(V):
        r = GaussianRasterizer(settings[v], channels=channels)
        r.grad_splat_limit = limit
        screen = torch.zeros(P, 3, device=dev, requires_grad=True)
        im, ra, de = r(means2D=screen, **kwargs(A))
        imgs.append(im), radii.append(ra), depths.append(de), screens.append(screen)
    loss = sum((im * dL[v]).sum() for v, im in enumerate(imgs))
    loss.backward()

    # one batched call
    B = _leaves(g, names, dev)
    if mode == "limit":
        for n in ("opacities", "colors"):
            B[n].requires_grad_(False)
    rv = GaussianRasterizerViews(settings, channels=channels)
    rv.grad_splat_limit = limit
    screen_b = torch.zeros(V, P, 3, device=dev, requires_grad=True)
    im_b, ra_b, de_b = rv(means2D=screen_b, **kwargs(B))
    (im_b * dL).sum().backward()
    torch.cuda.synchronize()

    assert im_b.shape == (V, channels, H, W) and ra_b.shape == (V, P) and de_b.shape == (V, 1, H, W)
    for v in range(V):
        assert torch.equal(ra_b[v], radii[v]), f"radii of view {v}"
        assert torch.equal(im_b[v].view(torch.int32), imgs[v].view(torch.int32)), f"colour of view {v} not bit-identical"
        assert torch.equal(de_b[v].view(torch.int32), depths[v].view(torch.int32)), f"depth of view {v} not bit-identical"
        assert (ra_b[v] > 0).sum().item() > 100
        assert _close(screen_b.grad[v], screens[v].grad), f"dL_dmean2D of view {v}"
    for n in names:
        if A[n].grad is None:
            assert B[n].grad is None or float(B[n].grad.abs().max()) == 0.0
            continue
        assert float(A[n].grad.abs().max()) > 0.0, n
        assert _close(B[n].grad, A[n].grad), f"summed gradient of {n}"
    if limit is not None:
        assert float(B["means3D"].grad[limit:].abs().max()) == 0.0


def test_views_scratch_slices_are_single_view_blobs():
    """C ABI level: the binning/geometry slices of a batched forward hold exactly the single-view arrays."""
    from fluidnexus_amd import _lib
    from tests.hip_harness import HipRun, scene_kwargs, _t, _p, _view
    import ctypes as C
    dev = torch.device("cuda")
    P, W, H, V = 3000, 96, 80, 3
    g = S.random_gaussians(P, seed=2, box=0.45, log_scale=(-4.8, -2.6), center=(0.34, 0.3, -0.225))
    cams = S.arc_cameras(V, W, H, device="cpu")
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    singles = [HipRun(bg=bg, colors_precomp=g["colors"], scales=g["scales"], rotations=g["rotations"],
                      **scene_kwargs(g, c, W, H)) for c in cams]
    lib = _lib.raster()
    tan = math.tan(0.4)
    tx = (C.c_float * V)(*([tan] * V))
    view = _t(np.stack([c.world_view_transform.numpy().reshape(16) for c in cams]), dev)
    proj = _t(np.stack([c.full_proj_transform.numpy().reshape(16) for c in cams]), dev)
    campos = _t(np.stack([c.camera_center.numpy() for c in cams]), dev)
    m, o, col, sc, ro = (_t(g[k], dev) for k in ("means3D", "opacities", "colors", "scales", "rotations"))
    gb, ib = lib.fnx_geom_bytes(P, W, H), lib.fnx_image_bytes(W, H)
    geom = torch.zeros(V * gb, dtype=torch.uint8, device=dev)
    img = torch.zeros(V * ib, dtype=torch.uint8, device=dev)
    radii = torch.zeros(V, P, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.fnx_forward_stage1_views(3, V, geom.data_ptr(), img.data_ptr(), P, 0, 0, W, H, _p(m), None, _p(col),
                                            _p(o), _p(sc), 1.0, _p(ro), None, _p(view), _p(proj), _p(campos), tx, tx, 0,
                                            radii.data_ptr(), s))
    counts = []
    n = C.c_int(0)
    for v in range(V):
        _lib.check(lib.fnx_read_num_rendered(img.data_ptr() + v * ib, W, H, s, C.byref(n)))
        counts.append(int(n.value))
    assert counts == [h.R for h in singles]
    cap = max(counts) + 17
    bb = lib.fnx_binning_bytes(cap)
    binning = torch.zeros(V * bb, dtype=torch.uint8, device=dev)
    color = torch.zeros(V, 3, H, W, device=dev)
    depth = torch.zeros(V, 1, H, W, device=dev)
    _lib.check(lib.fnx_forward_stage2_views(3, V, geom.data_ptr(), binning.data_ptr(), cap, img.data_ptr(), P, W, H,
                                            _p(_t(bg, dev)), radii.data_ptr(), color.data_ptr(), depth.data_ptr(), s))
    torch.cuda.synchronize()
    T = ((W + 15) // 16) * ((H + 15) // 16)
    im_l, b_l = _lib.image_layout(W, H), _lib.binning_layout(cap)
    for v, h in enumerate(singles):
        it = h.intermediates()
        pl = _view(binning[v * bb:(v + 1) * bb], b_l.point_list, counts[v], torch.int32).cpu().numpy()
        assert (pl == it["point_list"]).all(), f"point_list of view {v}"
        rg = _view(img[v * ib:(v + 1) * ib], im_l.ranges, 2 * T, torch.int32).view(T, 2).cpu().numpy()
        assert (rg == it["ranges"]).all()
        nc = _view(img[v * ib:(v + 1) * ib], im_l.n_contrib, H * W, torch.int32).view(H, W).cpu().numpy()
        assert (nc == it["n_contrib"]).all()
        assert (color[v].cpu().numpy().view(np.uint32) == it["color"].view(np.uint32)).all()
    # too many views / missing radii are argument errors, not crashes
    assert lib.fnx_forward_stage1_views(3, 17, geom.data_ptr(), img.data_ptr(), P, 0, 0, W, H, _p(m), None, _p(col),
                                        _p(o), _p(sc), 1.0, _p(ro), None, _p(view), _p(proj), _p(campos), tx, tx, 0,
                                        radii.data_ptr(), s) == _lib.FNX_ERR_INVALID_ARG


def test_fused_loss_batch_matches_per_image():
    from fluidnexus_amd.losses import fused_l1_dssim_grey, fused_l1_ssim
    dev = torch.device("cuda")
    gen = torch.Generator(device="cpu").manual_seed(0)
    N, H, W = 3, 70, 90
    x = torch.rand(N, 3, H, W, generator=gen).to(dev)
    y = torch.rand(N, 3, H, W, generator=gen).to(dev)
    for fn in (fused_l1_ssim, fused_l1_dssim_grey):
        xa = x.clone().requires_grad_(True)
        a1, a2 = fn(xa, y)
        w1 = torch.tensor([0.3, 1.1, -0.4], device=dev)
        w2 = torch.tensor([0.9, -0.2, 0.5], device=dev)
        ((a1 * w1).sum() + (a2 * w2).sum()).backward()
        assert a1.shape == (N,) and a2.shape == (N,)
        for n in range(N):
            xb = x[n].clone().requires_grad_(True)
            b1, b2 = fn(xb, y[n])
            (b1 * w1[n] + b2 * w2[n]).backward()
            assert torch.allclose(a1[n], b1, rtol=1e-6, atol=0) and torch.allclose(a2[n], b2, rtol=1e-6, atol=0)
            assert torch.allclose(xa.grad[n], xb.grad, rtol=1e-6, atol=1e-12)


def test_fused_image_loss_scalar_matches_terms():
    """fused_image_loss (one scalar for the batch) == the weighted sum of the per-image terms, value and gradient."""
    from fluidnexus_amd.losses import fused_image_loss, fused_l1_dssim_grey, fused_l1_ssim
    dev = torch.device("cuda")
    gen = torch.Generator(device="cpu").manual_seed(1)
    N, H, W = 4, 96, 80
    x = torch.rand(N, 3, H, W, generator=gen).to(dev)
    y = torch.rand(N, 3, H, W, generator=gen).to(dev)
    lam, lam_img = 0.2, 1.5
    for grey in (True, False):
        xa = x.clone().requires_grad_(True)
        loss, per = fused_image_loss(xa, y, lam, lam_img, grey=grey)
        (loss * 0.7).backward()
        xb = x.clone().requires_grad_(True)
        if grey:
            l1, dssim = fused_l1_dssim_grey(xb, y)
        else:
            l1, ss = fused_l1_ssim(xb, y)
            dssim = 1.0 - ss
        ref = (((1.0 - lam) * l1 + lam * dssim) * lam_img).sum()
        (ref * 0.7).backward()
        assert loss.shape == () and per.shape == (N, 2) and not per.requires_grad
        assert torch.allclose(loss, ref, rtol=2e-6, atol=0)
        assert torch.allclose(per[:, 0], l1, rtol=2e-6, atol=0) and torch.allclose(1.0 - per[:, 1], dssim, rtol=1e-5, atol=1e-7)
        assert (xa.grad - xb.grad).abs().max().item() <= 1e-5 * xb.grad.abs().max().item()


def test_views_sync_free_capacity_overflow_is_reported():
    """Sync-free mode sizes the binning buffer from a high-water mark; when a later batch needs more, nothing is
    rendered for the overflowing view and check_status() raises (the status words are written by the forward's
    last kernel into the persistent ring), after which the mark has grown and the same call succeeds."""
    from fluidnexus_amd import _lib, rasterizer
    from fluidnexus_amd.rasterizer import GaussianRasterizerViews, ViewBatch
    dev = torch.device("cuda")
    W = H = 64
    g = S.random_gaussians(3000, seed=2, box=0.5, log_scale=(-4.5, -3.0))
    cams = S.arc_cameras(2, W, H, target=(0.0, 0.0, 0.0), distance=2.0, height=0.1, device=dev)
    bg = torch.zeros(3, device=dev)
    vb = ViewBatch(_settings(cams, W, H, bg, [0.8, 0.8]))
    t = {k: torch.tensor(v, device=dev) for k, v in g.items()}

    def run():
        return GaussianRasterizerViews(vb)(means3D=t["means3D"], means2D=torch.zeros(2, 3000, 3, device=dev),
                                           shs=None, colors_precomp=t["colors"], opacities=t["opacities"],
                                           scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    try:
        rasterizer.set_host_sync(False)
        rasterizer._capacity_hwm.clear()
        ref = run()[0].clone()          # first call sizes the buffer with a host read
        rasterizer.check_status()
        key = (t["means3D"].device.index, W, H, 3, 3000)
        assert rasterizer._capacity_hwm[key] > 1024
        rasterizer._capacity_hwm[key] = 64   # pretend the mark came from a much lighter batch
        run()
        with pytest.raises(_lib.FnxError):
            rasterizer.check_status()
        assert rasterizer._capacity_hwm[key] > 1024  # grown from the reported instance count
        again = run()[0]
        rasterizer.check_status()
        assert torch.equal(again, ref)
    finally:
        rasterizer.set_host_sync(True)


def test_capacity_overflow_inside_a_graph_replay_is_reported():
    """A forward recorded into a hipGraph keeps its binning capacity; when the replayed scene has grown beyond it (the
    splats are inflated in place between replays) the replay renders nothing for the overflowing views and the next
    check_status() raises, although no Python ran during the replay: the captured forward's status rows are persistent
    and re-read on every check."""
    from fluidnexus_amd import _lib, rasterizer
    from fluidnexus_amd.rasterizer import GaussianRasterizerViews, ViewBatch
    dev = torch.device("cuda")
    W = H = 64
    g = S.random_gaussians(3000, seed=2, box=0.5, log_scale=(-5.5, -4.5))
    cams = S.arc_cameras(2, W, H, target=(0.0, 0.0, 0.0), distance=2.0, height=0.1, device=dev)
    bg = torch.zeros(3, device=dev)
    vb = ViewBatch(_settings(cams, W, H, bg, [0.8, 0.8]))
    t = {k: torch.tensor(v, device=dev) for k, v in g.items()}
    screen = torch.zeros(2, 3000, 3, device=dev)

    def run():
        return GaussianRasterizerViews(vb)(means3D=t["means3D"], means2D=screen, shs=None, colors_precomp=t["colors"],
                                           opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"],
                                           cov3D_precomp=None)[0]
    stream = torch.cuda.Stream()
    try:
        rasterizer.set_host_sync(False)
        rasterizer._capacity_hwm.clear()
        rasterizer.release_captured_status()
        with torch.no_grad():
            run()                      # sizes the buffer (small splats)
            rasterizer.check_status()
            graph = torch.cuda.CUDAGraph()
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                run()
                torch.cuda.synchronize()
                with torch.cuda.graph(graph, stream=stream):
                    out = run()
            graph.replay()
            torch.cuda.synchronize()
            rasterizer.check_status()  # fits: no error, and the check does not consume the captured rows
            small, keep = out.clone(), t["scales"].clone()
            t["scales"].add_(2.0)      # e^2 times larger splats: several times the instances
            graph.replay()
            torch.cuda.synchronize()
            with pytest.raises(_lib.FnxError):
                rasterizer.check_status()
            t["scales"].copy_(keep)
            graph.replay()
            torch.cuda.synchronize()
            rasterizer.check_status()
            assert torch.equal(out, small)
    finally:
        rasterizer.release_captured_status()
        rasterizer.set_host_sync(True)


def test_image_loss_value_and_grad_matches_autograd_node():
    """The hot loop's two-launch image term (scalar reduction inside the backward launch) against the autograd
    node fused_image_loss: loss, per-image terms and the gradient with respect to the rendered batch."""
    from fluidnexus_amd.losses import fused_image_loss, image_loss_value_and_grad
    dev = torch.device("cuda")
    gen = torch.Generator(device="cpu").manual_seed(3)
    img = torch.rand(5, 3, 70, 90, generator=gen).to(dev)
    gt = torch.rand(5, 3, 70, 90, generator=gen).to(dev)
    x = img.clone().requires_grad_(True)
    loss_a, per_a = fused_image_loss(x, gt, 0.2, 0.8)
    loss_a.backward()
    loss_b, per_b, g_b = image_loss_value_and_grad(img, gt, 0.2, 0.8)
    torch.cuda.synchronize()
    assert abs(loss_a.item() - loss_b.item()) <= 2e-6 * abs(loss_a.item())
    assert (per_a - per_b).abs().max().item() <= 2e-6
    assert torch.equal(x.grad, g_b)


def test_lean_geometry_gives_the_same_render_and_gradients():
    """fnx_set_lean_geometry(1): a view batch that skips the unread GeometryState copies and keeps ONE world covariance
    array (view 0's) equals the full one bit for bit in colour / depth / radii, and in the gradients up to the order of
    the atomics -- including splats that view 0 does not see but another view does."""
    import math
    import torch
    from fluidnexus_amd import rasterizer, synthetic as S
    from fluidnexus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizerViews
    W = H = 96
    g = S.to_torch(S.random_gaussians(6000, seed=8, box=1.5, log_scale=(-4.0, -2.5)))  # box 1.5: splats behind some cameras
    cams = S.ring_cameras(4, W, H, distance=1.2)
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    tan = math.tan(0.4)
    rs = [GaussianRasterizationSettings(H, W, tan, tan, bg, 1.0, c.world_view_transform, c.full_proj_transform, 0,
                                        c.camera_center, False) for c in cams]
    P = g["means3D"].shape[0]
    dL = torch.randn(4, 3, H, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    res = {}
    for lean in (False, True):
        rasterizer.set_lean_geometry(lean)
        try:
            leaves = {k: v.clone().requires_grad_(True) for k, v in g.items()}
            out = GaussianRasterizerViews(rs)(means3D=leaves["means3D"], means2D=torch.zeros(4, P, 3, device="cuda", requires_grad=True),
                                              opacities=leaves["opacities"], colors_precomp=leaves["colors"],
                                              scales=leaves["scales"], rotations=leaves["rotations"])
            grads = torch.autograd.grad([out[0]], list(leaves.values()), grad_outputs=[dL])
            res[lean] = (out, grads)
        finally:
            rasterizer.set_lean_geometry(False)
    (o0, g0), (o1, g1) = res[False], res[True]
    vis = o0[1] > 0
    assert bool((vis[1:].any(0) & ~vis[0]).any()), "the scene must hold splats view 0 does not see"
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    for a, b in zip(g0, g1):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max())


def test_sort_narrow_mode_is_exact_when_it_applies_and_reports_when_it_does_not():
    """fnx_set_sort_narrow(1): without the fourth depth-sort pass a scene whose depth keys span < 2^27 ulps renders bit
    for bit as before and reports its span; a scene spanning more raises FNX_ERR_SORT_SPAN at check_status()."""
    import math
    import torch
    from fluidnexus_amd import _lib, rasterizer, synthetic as S
    from fluidnexus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizerViews
    W = H = 96
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    tan = math.tan(0.4)

    def render(g, cams):
        rs = [GaussianRasterizationSettings(H, W, tan, tan, bg, 1.0, c.world_view_transform, c.full_proj_transform, 0,
                                            c.camera_center, False) for c in cams]
        P = g["means3D"].shape[0]
        out = GaussianRasterizerViews(rs)(means3D=g["means3D"], means2D=torch.zeros(len(cams), P, 3, device="cuda"),
                                          opacities=g["opacities"], colors_precomp=g["colors"], scales=g["scales"],
                                          rotations=g["rotations"])
        torch.cuda.synchronize()
        return out

    rasterizer.set_host_sync(False, 2_000_000)
    try:
        near = S.to_torch(S.plume_gaussians(5000, seed=1, channels=3))  # depths 1.5 .. 1.7: ~2^21 ulps
        cams = S.arc_cameras(2, W, H)
        rasterizer.max_sort_span_bits = 0
        ref = render(near, cams)
        rasterizer.check_status()
        assert 0 < rasterizer.max_sort_span_bits <= 25
        rasterizer.set_sort_narrow(True)
        got = render(near, cams)
        rasterizer.check_status()
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
        gw = S.random_gaussians(3000, seed=2, box=0.3, log_scale=(-4.0, -2.5))
        gw["means3D"][:50, 2] = -4.0e6  # depths 1.7 .. 2.3 and 4e6: 21 binades apart, > 2^27 ulps of span
        gw["scales"][:50] = 1.0e5
        wide = S.to_torch(gw)
        render(wide, [S.front_camera(W, H)])
        with pytest.raises(_lib.FnxError) as e:
            rasterizer.check_status()
        assert e.value.code == _lib.FNX_ERR_SORT_SPAN
        rasterizer.set_sort_narrow(False)
        render(wide, [S.front_camera(W, H)])
        rasterizer.check_status()
        assert rasterizer.max_sort_span_bits >= 27
    finally:
        rasterizer.set_sort_narrow(False)
        rasterizer.set_host_sync(True)
