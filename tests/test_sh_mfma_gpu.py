"""The SH colour kernel on the matrix cores (the fast arithmetic's) (csrc/sh_mfma.h; FNX_LAB_SH_MFMA=0 / 1 pins the choice) against the production
kernel: same clamp flags except at the clamp's edge, colours to fp32 rounding -- i.e. the operand layout of
v_mfma_f32_4x4x1_16B_f32 (block = Gaussian, row = view, column = channel) is what the kernel assumes."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("degree,V", [(3, 5), (2, 3), (1, 8), (0, 2)])
def test_mfma_sh_colours_match_the_scalar_kernel(degree, V):
    from fluidnexus_amd import synthetic as S
    from fluidnexus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizerViews
    import math
    P, W, H = 5003, 96, 96
    g = S.random_gaussians(P, seed=degree, log_scale=(-4.5, -2.5), channels=3)
    cams = S.arc_cameras(V, W, H, device="cuda")
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    rng = np.random.RandomState(7)
    M = (degree + 1) ** 2
    shs = np.zeros((P, M, 3), np.float32)
    shs[:, 0] = rng.uniform(-1.5, 1.5, size=(P, 3))
    shs[:, 1:] = rng.normal(size=(P, M - 1, 3)) * 0.4
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    shs_t = torch.from_numpy(shs).cuda()
    sets = [GaussianRasterizationSettings(H, W, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), bg, 1.0, c.world_view_transform,
                                          c.full_proj_transform, degree, c.camera_center, False) for c in cams]
    out = {}
    for lab in ("0", "1"):
        os.environ["FNX_LAB_SH_MFMA"] = lab
        try:
            rv = GaussianRasterizerViews(sets, channels=3)
            with torch.no_grad():
                im, radii, _ = rv(means3D=t["means3D"], means2D=torch.zeros(V, P, 3, device="cuda"), opacities=t["opacities"],
                                  shs=shs_t, scales=t["scales"], rotations=t["rotations"])
            torch.cuda.synchronize()
            out[lab] = (im.clone(), radii.clone())
        finally:
            os.environ.pop("FNX_LAB_SH_MFMA", None)
    assert torch.equal(out["0"][1], out["1"][1])
    assert float(out["0"][0].abs().max()) > 0.05
    assert float((out["0"][0] - out["1"][0]).abs().max()) <= 5e-6


def test_fast_arithmetic_takes_the_matrix_core_kernel_and_stays_in_tolerance():
    """blend_math = fast routes a view batch's SH colours through the MFMA kernel (no environment override); the image stays
    within the fast mode's stated pixel tolerance of the exact mode's."""
    from fluidnexus_amd import rasterizer, synthetic as S
    from fluidnexus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizerViews
    import math
    P, W, H, V, degree = 4001, 96, 96, 3, 3
    g = S.random_gaussians(P, seed=2, log_scale=(-4.5, -2.5), channels=3)
    cams = S.arc_cameras(V, W, H, device="cuda")
    bg = torch.zeros(3, device="cuda")
    rng = np.random.RandomState(9)
    shs = np.zeros((P, 16, 3), np.float32)
    shs[:, 0] = rng.uniform(-1.0, 1.5, size=(P, 3))
    shs[:, 1:] = rng.normal(size=(P, 15, 3)) * 0.3
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    shs_t = torch.from_numpy(shs).cuda()
    sets = [GaussianRasterizationSettings(H, W, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), bg, 1.0, c.world_view_transform,
                                          c.full_proj_transform, degree, c.camera_center, False) for c in cams]
    os.environ.pop("FNX_LAB_SH_MFMA", None)
    ims = {}
    try:
        for mode in ("exact", "fast"):
            rasterizer.set_blend_math(mode)
            with torch.no_grad():
                ims[mode] = GaussianRasterizerViews(sets, channels=3)(
                    means3D=t["means3D"], means2D=torch.zeros(V, P, 3, device="cuda"), opacities=t["opacities"], shs=shs_t,
                    scales=t["scales"], rotations=t["rotations"])[0].clone()
    finally:
        rasterizer.set_blend_math("exact")
    d = (ims["fast"] - ims["exact"]).abs()
    assert int((d > 2e-5).sum()) <= max(2, int(1e-4 * d.numel())) and float(d.max()) <= 2e-3
