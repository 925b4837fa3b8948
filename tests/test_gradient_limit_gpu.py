"""-m gpu: the forward's gradient limit (include/fnx_raster.h fnx_request_gradient_limit).  Told that only splats with id <
L will be differentiated, the forward records per pixel the list position of the last such splat at or in front of the
pixel's last contributor and lays down backward work items only up to it; the backward walks no further.  What lies
behind an entry enters its gradient only through the final colour and transmittance the forward stores, so nothing may
change: gradients equal up to the order of the global atomics (the same addends), every forward output bit for bit,
fewer work items."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fluidnexus_amd import synthetic as S  # noqa: E402


def _blob(t, off, n, dtype):
    al = (-t.data_ptr()) % 256
    es = torch.empty(0, dtype=dtype).element_size()
    return t[al + off: al + off + n * es].view(dtype)


def _scene(channels, P_dyn, P_static):
    a = S.plume_gaussians(P_dyn, seed=4, radius=0.05, y_range=(0.15, 0.45), channels=channels)
    b = S.backdrop_gaussians(P_static, seed=5, channels=channels)  # behind the plume, seen through it
    g = {k: np.concatenate([a[k], b[k]], 0) for k in a}
    # a few "static" splats INSIDE the plume as well: the walk must go on behind them
    g["means3D"][P_dyn:P_dyn + 200] = a["means3D"][:200] + 0.002
    return g


@pytest.mark.parametrize("channels,math_mode,screen", [(3, "exact", False), (3, "fast", False), (1, "fast", True), (3, "fast", True)])
def test_limited_forward_changes_no_gradient_and_no_output(channels, math_mode, screen, monkeypatch):
    from fluidnexus_amd import _lib, rasterizer
    from fluidnexus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizerViews, ViewBatch
    dev = torch.device("cuda")
    W, H, V = 160, 128, 2
    P_dyn, P_static = 30000, 4000
    g = _scene(channels, P_dyn, P_static)
    P = P_dyn + P_static
    cams = S.arc_cameras(V, W, H, device="cuda")
    bg = torch.tensor([0.2, 0.5, 0.1], device=dev)
    tan = math.tan(0.4)
    settings = [GaussianRasterizationSettings(H, W, tan, tan, bg, 1.0, c.world_view_transform, c.full_proj_transform, 0,
                                              c.camera_center, False) for c in cams]
    dL = torch.tensor(np.random.RandomState(9).normal(size=(V, channels, H, W)).astype(np.float32), device=dev)
    lib = _lib.raster()
    IL = _lib.image_layout(W, H)
    ib = lib.fnx_image_bytes(W, H)
    rasterizer.set_blend_math(math_mode)
    real_options = rasterizer._call_options
    res = {}
    try:
        for name in ("limited", "full"):
            if name == "full":  # the same calls with the forward's limit suppressed: the backward walks to the last contributor
                monkeypatch.setattr(rasterizer, "_call_options",
                                    lambda *a, **k: real_options(*a, **dict(k, grad_splat_limit=None)))
            vb = ViewBatch(settings)
            L = {n: torch.tensor(g[n], device=dev, requires_grad=True) for n in g}
            rv = GaussianRasterizerViews(vb, channels=channels)
            rv.grad_splat_limit = P_dyn
            m2d = torch.zeros(V, P, 3, device=dev, requires_grad=screen)
            im, ra, de = rv(means3D=L["means3D"], means2D=m2d, opacities=L["opacities"], colors_precomp=L["colors"],
                            scales=L["scales"], rotations=L["rotations"])
            img = im.grad_fn.saved_tensors[-1]
            (im * dL).sum().backward()
            torch.cuda.synchronize()
            rasterizer.check_status()
            views = []
            for v in range(V):
                iv = img[v * ib:(v + 1) * ib]
                nc = _blob(iv, IL.n_contrib, 2 * H * W, torch.int32).clone().view(2, H * W)
                views.append(dict(header=_blob(iv, IL.header, 16, torch.int32).tolist(), n_contrib=nc[0], limit=nc[1],
                                  final_T=_blob(iv, IL.final_T, H * W, torch.int32).clone()))
            res[name] = dict(im=im.detach().clone(), de=de.clone(), ra=ra.clone(), views=views,
                             grads={n: (L[n].grad.detach().clone() if L[n].grad is not None else None) for n in L},
                             m2d=m2d.grad.detach().clone() if screen else None)
    finally:
        monkeypatch.setattr(rasterizer, "_call_options", real_options)
        rasterizer.set_blend_math("exact")
    a, b = res["limited"], res["full"]
    assert torch.equal(a["im"].view(torch.int32), b["im"].view(torch.int32))
    assert torch.equal(a["de"].view(torch.int32), b["de"].view(torch.int32)) and torch.equal(a["ra"], b["ra"])
    items = [0, 0]
    for va, vf in zip(a["views"], b["views"]):
        assert torch.equal(va["n_contrib"], vf["n_contrib"]) and torch.equal(va["final_T"], vf["final_T"])
        assert torch.equal(vf["limit"], vf["n_contrib"])  # no limit: the walking limit IS the last contributor
        assert bool((va["limit"] <= va["n_contrib"]).all()) and int((va["limit"] < va["n_contrib"]).sum()) > 1000
        assert va["header"][9] == P_dyn and vf["header"][9] == -1  # HDR_DYN_LIMIT (0xFFFFFFFF without a request)
        items[0] += va["header"][4]
        items[1] += vf["header"][4]
    assert 0 < items[0] < items[1], items  # fewer backward work items
    for n, ga in a["grads"].items():
        gb = b["grads"][n]
        assert (ga is None) == (gb is None), n
        if ga is not None:
            # the same addends per splat; the tiles' sums meet in global atomics whose order is not fixed, so two runs of
            # the SAME configuration differ in the last bits too: fp32 summation-order tolerance, per element
            d, ref = (ga[:P_dyn] - gb[:P_dyn]).abs(), gb[:P_dyn].abs()
            assert bool((d <= 2e-5 * ref + 2e-6 * ref.max()).all()), (n, float(d.max()), float(ref.max()))
            assert float(ga[P_dyn:].abs().max()) == 0.0 and float(gb[P_dyn:].abs().max()) == 0.0, n
    assert float(a["grads"]["means3D"][:P_dyn].abs().max()) > 0
    if screen:
        d, ref = (a["m2d"] - b["m2d"]).abs(), b["m2d"].abs()
        assert bool((d <= 2e-5 * ref + 2e-6 * ref.max()).all())
    print(f"[gradient limit ch{channels} {math_mode}] backward work items {items[1]} -> {items[0]}")


def test_backward_beyond_the_forwards_limit_is_refused(oracle):
    """Through the C ABI: a backward call that differentiates splats its forward's limit excluded produces no gradients and
    leaves FNX_ERR_INVALID_ARG in the view's status word; within the limit it gives the oracle's gradients."""
    from fluidnexus_amd import _lib
    from tests.hip_harness import HipRun, scene_kwargs
    P, W, H, limit = 800, 64, 64, 300
    g = S.random_gaussians(P, seed=41, log_scale=(-4.5, -2.5))
    cam = S.front_camera(W, H, device="cpu")
    bg = np.array([0.2, 0.3, 0.4], np.float32)
    kw = scene_kwargs(g, cam, W, H, 0.8)
    extra = dict(colors_precomp=g["colors"], scales=g["scales"], rotations=g["rotations"])
    f = oracle.forward(kw["means3D"], kw["opacities"], bg, kw["view"], kw["proj"], kw["campos"], W, H, kw["tanx"], kw["tany"],
                       channels=3, **extra)
    dL = np.random.RandomState(1).normal(size=(3, H, W)).astype(np.float32)
    _lib.check(_lib.raster().fnx_request_gradient_limit(limit))  # one-shot: consumed by the stage 2 inside HipRun
    h = HipRun(bg=bg, channels=3, **kw, **extra)
    bad = h.backward(dL)  # no limit = all P splats: more than the forward prepared
    assert all(not v.any() for v in bad.values())
    assert h.status() == _lib.FNX_ERR_INVALID_ARG
    _lib.check(_lib.raster().fnx_request_gradient_limit(limit))
    h = HipRun(bg=bg, channels=3, **kw, **extra)
    assert h.status() == _lib.FNX_OK
    got, ref = h.backward(dL, grad_splat_limit=limit), oracle.backward(f, dL)
    for n in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dcolors"):
        r, q = ref[n].reshape(P, -1), got[n].reshape(P, -1)
        assert np.abs(q[:limit] - r[:limit]).max() <= 2e-4 * np.abs(r).max(), n
        assert (q[limit:] == 0).all(), n
    h2 = HipRun(bg=bg, channels=3, **kw, **extra)  # the request was one-shot: this forward has no limit again
    _ = h2.backward(dL)
    assert h2.status() == _lib.FNX_OK
