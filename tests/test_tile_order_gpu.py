"""-m gpu: depth hints (include/fnx_raster.h).  The forward records how deep every tile's list was walked; the next
forward of the same view batch hands the tiles that went deep -- lists that do not saturate: thousands of contributing
entries per pixel -- to the first workgroups of the blend launch at raised wave priority.  A scheduling hint only:
every output the reference defines, and the hand-over to the backward, must be bit-identical with and without it."""
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


@pytest.mark.parametrize("channels,split", [(1, False), (3, True), (3, False)])
def test_deep_first_tile_order_changes_nothing(channels, split):
    from fluidnexus_amd import _lib, rasterizer
    from fluidnexus_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizerViews, StaticBin, ViewBatch)
    dev = torch.device("cuda")
    W, H, V = 160, 128, 2
    T = ((W + 15) // 16) * ((H + 15) // 16)
    P_dyn, P_static = 40000, (3000 if split else 0)
    # a dense, low-opacity plume: thousands of contributing entries per pixel in the tiles it covers
    a = S.plume_gaussians(P_dyn, seed=4, radius=0.05, y_range=(0.15, 0.45), channels=channels)
    if P_static:
        b = S.backdrop_gaussians(P_static, seed=5, channels=channels)
        g = {k: np.concatenate([a[k], b[k]], 0) for k in a}
    else:
        g = a
    P = P_dyn + P_static
    cams = S.arc_cameras(V, W, H, device="cuda")
    bg = torch.tensor([0.2, 0.5, 0.1], device=dev)
    tan = math.tan(0.4)
    settings = [GaussianRasterizationSettings(H, W, tan, tan, bg, 1.0, c.world_view_transform, c.full_proj_transform, 0,
                                              c.camera_center, False) for c in cams]
    rng = np.random.RandomState(9)
    dL = torch.tensor(rng.normal(size=(V, channels, H, W)).astype(np.float32), device=dev)
    lib = _lib.raster()
    IL = _lib.image_layout(W, H)
    ib = lib.fnx_image_bytes(W, H)
    rasterizer.set_deep_variant(True, 256)
    try:
        vb = ViewBatch(settings)
        L = {n: torch.tensor(g[n], device=dev, requires_grad=(n == "means3D")) for n in g}
        sb = None
        if split:
            sb = StaticBin(vb, L["means3D"][P_dyn:].detach(), L["opacities"][P_dyn:], P_dyn, colors_precomp=L["colors"][P_dyn:],
                           scales=L["scales"][P_dyn:], rotations=L["rotations"][P_dyn:], channels=channels)
        outs = []
        for call in range(3):
            rv = GaussianRasterizerViews(vb, channels=channels)
            rv.grad_splat_limit = P_dyn
            rv.static_bin = sb
            L["means3D"].grad = None
            screen = torch.zeros(V, P, 3, device=dev, requires_grad=True)
            im, ra, de = rv(means3D=L["means3D"], means2D=screen, opacities=L["opacities"], colors_precomp=L["colors"],
                            scales=L["scales"], rotations=L["rotations"])
            img = im.grad_fn.saved_tensors[-1]
            (im * dL).sum().backward()
            torch.cuda.synchronize()
            per_view = []
            for v in range(V):
                iv = img[v * ib:(v + 1) * ib]
                per_view.append(dict(header=_blob(iv, IL.header, 8, torch.int32).tolist(),
                                     final_T=_blob(iv, IL.final_T, H * W, torch.int32).clone(),
                                     n_contrib=_blob(iv, IL.n_contrib, H * W, torch.int32).clone(),
                                     acc=_blob(iv, IL.acc_final, channels * H * W, torch.int32).clone()))
            outs.append(dict(im=im.detach().clone(), de=de.clone(), ra=ra.clone(), views=per_view,
                             grad=L["means3D"].grad.detach().clone(), hint=vb.depth_hint(channels).clone()))
    finally:
        rasterizer.set_deep_variant(True, 1024)
    first, second, third = outs
    # header word 5 = deep-tile count | bit length of the depth-key span << 24 (fnx_state.h HDR_DEEP_COUNT)
    assert all(v["header"][5] & 0xFFFFFF == 0 for v in first["views"])  # no history: the plain XCD-aware tile order
    deep = [v["header"][5] & 0xFFFFFF for v in second["views"]]
    assert min(deep) > 0 and max(deep) < T, deep           # history: the plume tiles are scheduled first
    assert int(first["hint"].max()) > 1000                 # ... because they went this deep
    for other in (second, third):
        assert torch.equal(first["im"].view(torch.int32), other["im"].view(torch.int32)), "colour not bit-identical"
        assert torch.equal(first["de"].view(torch.int32), other["de"].view(torch.int32))
        assert torch.equal(first["ra"], other["ra"]) and torch.equal(first["hint"], other["hint"])
        for va, vo in zip(first["views"], other["views"]):
            for k in ("final_T", "n_contrib", "acc"):
                assert torch.equal(va[k], vo[k]), k
            assert va["header"][4] == vo["header"][4]       # same number of backward work items
        scale = first["grad"].abs().max().item()
        assert scale > 0 and (first["grad"] - other["grad"]).abs().max().item() < 2e-4 * scale
