"""-m gpu: the two forms of the blend backward (include/fnx_raster.h fnx_set_backward_form) -- a lane per PIXEL (rows,
rounds 1-5) and a lane per LIST ENTRY (lanes, the default since round 6: csrc/raster_backward_lanes.h) -- against the
oracle and against each other, for every gradient mode, both channel counts and both blend arithmetics, on a plume with
several batches per tile (the carries of the lanes form cross chunk and batch boundaries) and with a gradient limit.
Reference: ch3 cuda_rasterizer/backward.cu:384-536."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from fluidnexus_amd import synthetic as S  # noqa: E402


def _ratio(ref, got, rel=1e-3, abs_of_max=2e-5):
    ref = ref.astype(np.float64)
    got = got.reshape(ref.shape).astype(np.float64)
    if ref.size == 0:
        return 0.0
    bound = rel * np.abs(ref) + abs_of_max * np.abs(ref).max()
    return float((np.abs(got - ref) / np.maximum(bound, 1e-300)).max())


_KEYS = {0: ("dL_dmeans3D", "dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dscales", "dL_drotations"),
         1: ("dL_dmeans3D", "dL_dmeans2D", "dL_dconic", "dL_dscales", "dL_drotations"),
         2: ("dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dscales", "dL_drotations"),
         3: ("dL_dmeans3D",)}


@pytest.mark.parametrize("channels", [3, 1])
@pytest.mark.parametrize("math_mode", ["exact", "fast"])
def test_lanes_and_rows_forms_against_the_oracle(oracle, channels, math_mode):
    from fluidnexus_amd import rasterizer
    from tests.hip_harness import HipRun, scene_kwargs
    W = H = 176  # 11 x 11 tiles; the plume's lists run to several 256-entry batches
    g = S.smoke_scene(14_000, 6_000, seed=3, channels=3) if channels == 3 else S.plume_gaussians(14_000, seed=3, channels=1)
    P = g["means3D"].shape[0]
    cam = S.arc_cameras(5, W, H, device="cpu")[2]
    bg = np.array([0.1, 0.3, 0.2], np.float32)
    kw = scene_kwargs(g, cam, W, H, 0.8)
    extra = dict(colors_precomp=g["colors"], scales=g["scales"], rotations=g["rotations"])
    f = oracle.forward(kw["means3D"], kw["opacities"], bg, kw["view"], kw["proj"], kw["campos"], W, H, kw["tanx"], kw["tany"],
                       channels=channels, **extra)
    assert int(f["n_contrib"].max()) > 512
    dL = np.random.RandomState(9).normal(size=(channels, H, W)).astype(np.float32)
    ref = oracle.backward(f, dL)
    rasterizer.set_blend_math(math_mode)
    try:
        h = HipRun(bg=bg, channels=channels, **kw, **extra)
        got = {}
        for form in ("rows", "lanes"):
            rasterizer.set_backward_form(form)
            assert rasterizer.get_backward_form() == form
            for mode in (0, 1, 2, 3):
                got[form, mode, -1] = h.backward(dL, geometry_only=mode)
            got[form, 3, P // 3] = h.backward(dL, grad_splat_limit=P // 3, geometry_only=3)
            got[form, 0, P // 3] = h.backward(dL, grad_splat_limit=P // 3, geometry_only=0)
    finally:
        rasterizer.set_backward_form("lanes")
        rasterizer.set_blend_math("exact")
    worst = {}
    for (form, mode, lim), gr in got.items():
        for k in _KEYS[mode]:
            r = ref[k].reshape(P, -1)
            x = gr[k].reshape(P, -1)
            n = P if lim < 0 else lim
            assert (x[n:] == 0).all(), (form, mode, lim, k)
            bound = 1e-3 * np.abs(r[:n].astype(np.float64)) + 2e-5 * np.abs(r).max()
            q = float((np.abs(x[:n].astype(np.float64) - r[:n]) / np.maximum(bound, 1e-300)).max()) if r.size else 0.0
            worst[form] = max(worst.get(form, 0.0), q)
            # fast arithmetic: an alpha that a rounding moves across 1 / 255 is a discontinuity of the reference's own
            # function; on this scene no element is seen beyond twice the bound
            assert q <= (1.0 if math_mode == "exact" else 2.0), (form, mode, lim, k, q)
    print(f"[backward forms ch{channels} {math_mode}] worst err / bound against the oracle: rows {worst['rows']:.3f}, lanes {worst['lanes']:.3f}")
    # the two forms against each other: same decisions, sums associated differently
    for mode in (0, 1, 2, 3):
        for k in _KEYS[mode]:
            assert _ratio(got["rows", mode, -1][k], got["lanes", mode, -1][k], rel=2e-4, abs_of_max=5e-6) <= 1.0, (mode, k)
