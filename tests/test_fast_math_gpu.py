"""-m gpu: the blend kernels' FAST arithmetic (fnx_set_blend_math(1)) against the CPU oracle.

The exact mode repeats the oracle bit for bit (test_raster_parity_gpu.py).  The fast mode keeps every list, every
index and every test of ch3 forward.cu:319-345 but evaluates the power with fused multiply-adds on pre-scaled
coefficients and the exponential with v_exp_f32, so its pixels carry a STATED tolerance (include/fnx_raster.h):

  * everything computed before the blend (radii, tiles_touched, ranges, point_list) stays bit-exact;
  * |pixel - oracle| <= 2e-5, except pixels where a rounding moves one alpha across 1/255 or one T across 1e-4:
    those are counted and may be at most max(2, 1e-4 H W) per image, and even they stay within 2e-3;
  * n_contrib / median depth differ on a counted number of pixels only (the same threshold flips);
  * gradients: per element |err| <= 1e-3 |ref| + 2e-5 max|ref| (the worst element is reported on failure).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from fluidnexus_amd import synthetic as S  # noqa: E402

PIX_TOL = 2e-5
FLIP_TOL = 2e-3


@pytest.fixture(autouse=True)
def _fast_mode():
    from fluidnexus_amd import rasterizer
    rasterizer.set_blend_math("fast")
    yield
    rasterizer.set_blend_math("exact")


def _pair(oracle, g, cam, W, H, bg, channels=3, fov=0.8):
    from tests.hip_harness import HipRun, scene_kwargs
    kw = scene_kwargs(g, cam, W, H, fov)
    extra = dict(colors_precomp=g["colors"], scales=g["scales"], rotations=g["rotations"])
    f = oracle.forward(kw["means3D"], kw["opacities"], bg, kw["view"], kw["proj"], kw["campos"], W, H, kw["tanx"],
                       kw["tany"], channels=channels, **extra)
    h = HipRun(bg=bg, channels=channels, **kw, **extra)
    return f, h


def mixed_bound_report(ref, got, rel=1e-3, abs_of_max=2e-5):
    """(ok, message): per-element |err| <= rel |ref| + abs_of_max max|ref|; the message names the worst element."""
    ref = ref.astype(np.float64)
    got = got.reshape(ref.shape).astype(np.float64)
    if ref.size == 0:
        return True, "empty"
    err = np.abs(got - ref)
    bound = rel * np.abs(ref) + abs_of_max * np.abs(ref).max()
    ratio = err / np.maximum(bound, 1e-300)
    i = np.unravel_index(np.argmax(ratio), ref.shape)
    msg = f"worst element {i}: ref {ref[i]:.6e} got {got[i]:.6e} err {err[i]:.3e} bound {bound[i]:.3e} (max|ref| {np.abs(ref).max():.3e})"
    return bool(ratio[i] <= 1.0), msg


def _check_forward(f, h, name):
    it = h.intermediates()
    assert h.R == f["num_rendered"]
    for k in ("radii", "tiles_touched", "ranges", "point_list"):  # binning: bit-exact in either mode
        a, b = it[k].astype(np.int64), f[k].astype(np.int64)
        assert a.shape == b.shape and (a == b).all(), f"{k}: {np.sum(a != b)} mismatches"
    HW = f["final_T"].size
    allowed = max(2, int(1e-4 * HW))
    d = np.abs(it["color"].astype(np.float64) - f["color"]).max(0)
    n_bad = int((d > PIX_TOL).sum())
    dT = np.abs(it["final_T"].astype(np.float64) - f["final_T"])
    n_nc = int((it["n_contrib"] != f["n_contrib"]).sum())
    n_dep = int((it["depth"].reshape(-1) != f["depth"].reshape(-1)).sum())
    print(f"[fast {name}] pixels {HW}: max|d colour| {d.max():.2e} mean {d.mean():.2e}, > {PIX_TOL:g}: {n_bad} (allowed {allowed}); "
          f"max|d T| {dT.max():.2e}; n_contrib differs on {n_nc}; depth differs on {n_dep}")
    assert n_bad <= allowed, f"{n_bad} pixels off by more than {PIX_TOL}"
    assert d.max() <= FLIP_TOL
    assert int((dT > PIX_TOL).sum()) <= allowed and dT.max() <= FLIP_TOL
    assert n_nc <= max(4, int(1e-3 * HW)), f"n_contrib differs on {n_nc} pixels"
    assert n_dep <= max(4, int(1e-3 * HW)), f"median depth differs on {n_dep} pixels"
    return it


def _check_grads(go, gh, name, keys=None):
    bad = []
    for k, ref in go.items():
        if ref.size == 0 or (keys and k not in keys):
            continue
        ok, msg = mixed_bound_report(ref, gh[k])
        print(f"[fast {name}] {k}: {msg}")
        if not ok:
            bad.append(f"{k}: {msg}")
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("P,W,H,seed", [(64, 32, 32, 0), (1000, 64, 48, 1), (3000, 100, 70, 3)])
def test_random_scenes_ch3(oracle, P, W, H, seed):
    g = S.random_gaussians(P, seed=seed, box=0.5, log_scale=(-5.0, -2.5))
    cam = S.front_camera(W, H, device="cpu")
    bg = np.array([0.1, 0.4, 0.9], np.float32)
    f, h = _pair(oracle, g, cam, W, H, bg)
    _check_forward(f, h, f"random P={P}")
    dL = np.random.RandomState(seed).normal(size=(3, H, W)).astype(np.float32)
    _check_grads(oracle.backward(f, dL), h.backward(dL), f"random P={P}")


def test_config1_shape(oracle):
    """BASELINE configs[0]: 10k random Gaussians, one 256x256 camera."""
    P, W, H = 10_000, 256, 256
    g = S.random_gaussians(P, seed=0, box=0.5, log_scale=(-5.5, -3.5))
    cam = S.front_camera(W, H, device="cpu")
    bg = np.zeros(3, np.float32)
    f, h = _pair(oracle, g, cam, W, H, bg)
    _check_forward(f, h, "config 1")
    dL = np.random.RandomState(1).normal(size=(3, H, W)).astype(np.float32)
    _check_grads(oracle.backward(f, dL), h.backward(dL), "config 1")


@pytest.mark.parametrize("channels", [1, 3])
def test_config2_and_3_style_plumes(oracle, channels):
    """Reduced-size frames of configs 2 (ScalarReal plume, 1 channel) and 3 (plume in front of a backdrop, 3 channels)
    from an arc camera: deep semi-transparent lists, several batches per tile."""
    W = H = 192
    if channels == 3:
        g = S.smoke_scene(16_000, 8_000, seed=2, channels=3)
    else:
        g = S.plume_gaussians(16_000, seed=2, channels=1)
    cam = S.arc_cameras(5, W, H, device="cpu")[1]
    bg = np.array([0.0, 0.0, 0.0], np.float32)
    f, h = _pair(oracle, g, cam, W, H, bg, channels=channels)
    it = _check_forward(f, h, f"plume ch{channels}")
    assert int(it["n_contrib"].max()) > 500
    dL = np.random.RandomState(4).normal(size=(channels, H, W)).astype(np.float32)
    go = oracle.backward(f, dL)
    _check_grads(go, h.backward(dL), f"plume ch{channels}")
    # positions only / fixed positions / geometry only modes of the fast backward
    ref = go["dL_dmeans3D"]
    ok, msg = mixed_bound_report(ref, h.backward(dL, geometry_only=3)["dL_dmeans3D"])
    assert ok, "mode 3: " + msg
    g1 = h.backward(dL, geometry_only=1)
    _check_grads(go, g1, f"plume ch{channels} mode 1", keys=("dL_dmeans3D", "dL_dmeans2D", "dL_dscales", "dL_drotations"))
    g2 = h.backward(dL, geometry_only=2)
    _check_grads(go, g2, f"plume ch{channels} mode 2", keys=("dL_dopacity", "dL_dcolors", "dL_dscales", "dL_drotations"))


@pytest.mark.parametrize("channels", [1, 3])
def test_ball_style_scene(oracle, channels):
    """BASELINE config 5 at reduced size, one camera of the ring (both rasterisers run per view there)."""
    W = H = 160
    g = S.smoke_scene(14_000, 6_000, seed=9, channels=3, ring=True) if channels == 3 else S.plume_gaussians(14_000, seed=9, channels=1)
    cam = S.ring_cameras(8, W, H, device="cpu")[3]
    bg = np.array([0.05, 0.1, 0.2], np.float32)
    f, h = _pair(oracle, g, cam, W, H, bg, channels=channels)
    _check_forward(f, h, f"ball ch{channels}")
    dL = np.random.RandomState(4).normal(size=(channels, H, W)).astype(np.float32)
    _check_grads(oracle.backward(f, dL), h.backward(dL), f"ball ch{channels}")


def test_fast_forward_is_deterministic_and_views_match_single_calls():
    """The fast forward is as deterministic as the exact one, and a view batch equals its single-view calls bit for bit."""
    import math
    import torch
    from fluidnexus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, GaussianRasterizerViews
    W = H = 128
    g = S.to_torch(S.smoke_scene(8000, 4000, seed=3, channels=3))
    cams = S.arc_cameras(3, W, H)
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    tan = math.tan(0.4)
    rs = [GaussianRasterizationSettings(H, W, tan, tan, bg, 1.0, c.world_view_transform, c.full_proj_transform, 0,
                                        c.camera_center, False) for c in cams]
    args = dict(means3D=g["means3D"], opacities=g["opacities"], colors_precomp=g["colors"], scales=g["scales"],
                rotations=g["rotations"])
    P = g["means3D"].shape[0]
    singles = [GaussianRasterizer(r)(means2D=torch.zeros(P, 3, device="cuda"), **args)[0] for r in rs]
    again = [GaussianRasterizer(r)(means2D=torch.zeros(P, 3, device="cuda"), **args)[0] for r in rs]
    batch = GaussianRasterizerViews(rs)(means2D=torch.zeros(3, P, 3, device="cuda"), **args)[0]
    for v in range(3):
        assert torch.equal(singles[v], again[v])
        assert torch.equal(singles[v], batch[v])


def _views_render(g, cams, W, H, bg_np, channels, deep_kernel, min_depth=256, backward_seed=None, static_split=False):
    """Two forwards of one view batch (the second one sees the depth hints of the first: deep tiles exist), fast mode."""
    import math
    import torch
    from fluidnexus_amd import _lib, rasterizer
    from fluidnexus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizerViews, StaticBin
    rasterizer.set_deep_kernel(2 if deep_kernel else 0)
    rasterizer.set_deep_variant(True, min_depth)
    try:
        t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
        bg = torch.from_numpy(bg_np).cuda()
        tan = math.tan(0.4)
        rs = [GaussianRasterizationSettings(H, W, tan, tan, bg, 1.0, c.world_view_transform.cuda(), c.full_proj_transform.cuda(),
                                            0, c.camera_center.cuda(), False) for c in cams]
        rz = GaussianRasterizerViews(rs, channels=channels)
        P, V = t["means3D"].shape[0], len(cams)
        if static_split:
            n_dyn = static_split
            rz.static_bin = StaticBin(rz.view_batch, t["means3D"][n_dyn:], t["opacities"][n_dyn:], colors_precomp=t["colors"][n_dyn:],
                                      scales=t["scales"][n_dyn:], rotations=t["rotations"][n_dyn:], channels=channels,
                                      id_offset=n_dyn)
            rz.grad_splat_limit = n_dyn
        leaves = {k: v.clone().requires_grad_(True) for k, v in t.items()}
        out = None
        for _ in range(2):
            m2d = torch.zeros(V, P, 3, device="cuda", requires_grad=True)
            out = rz(means3D=leaves["means3D"], means2D=m2d, opacities=leaves["opacities"], colors_precomp=leaves["colors"],
                     scales=leaves["scales"], rotations=leaves["rotations"])
        hint = rz.view_batch.depth_hint(channels)
        grads = None
        if backward_seed is not None:
            dL = torch.from_numpy(np.random.RandomState(backward_seed).normal(size=tuple(out[0].shape)).astype(np.float32)).cuda()
            gl = torch.autograd.grad([out[0]], [leaves["means3D"], leaves["opacities"], leaves["colors"], leaves["scales"],
                                                leaves["rotations"]], grad_outputs=[dL])
            grads = [x.detach().cpu().numpy() for x in gl]
        torch.cuda.synchronize()
        rasterizer.check_status()
        return out[0].detach().cpu().numpy(), out[2].detach().cpu().numpy(), hint.cpu().numpy(), grads
    finally:
        rasterizer.set_deep_kernel(5)  # the library's default
        rasterizer.set_deep_variant(True, 1024)


@pytest.mark.parametrize("channels,split", [(3, False), (1, False), (3, True)])
def test_deep_tile_kernel_matches_per_tile_kernel_and_oracle(oracle, channels, split):
    """blend_forward_deep_kernel (super-batches of 4 x 256 entries, transmittance pre-pass) against the per-tile kernel
    and the oracle on a plume in front of a wall: colours within the fast mode's tolerance, the tiles' depths equal,
    gradients (the backward reads the deep kernel's hand-over records) within the mixed bound."""
    from tests.hip_harness import scene_kwargs
    W = H = 160
    n_dyn = 14_000
    g = S.smoke_scene(n_dyn, 6_000, seed=9, channels=3, ring=True) if channels == 3 else S.plume_gaussians(n_dyn, seed=9, channels=1)
    cams = S.ring_cameras(8, W, H, device="cpu")[3:5]
    bg = np.array([0.05, 0.1, 0.2], np.float32)
    sp = n_dyn if split else False
    col_d, dep_d, hint_d, gr_d = _views_render(g, cams, W, H, bg, channels, True, backward_seed=4, static_split=sp)
    col_r, dep_r, hint_r, gr_r = _views_render(g, cams, W, H, bg, channels, False, backward_seed=4, static_split=sp)
    assert int((hint_r >= 256).sum()) > 10, "the scene must have deep tiles"
    assert (hint_d == hint_r).all(), f"last-contributor depth of {int((hint_d != hint_r).sum())} tiles differs"
    d = np.abs(col_d.astype(np.float64) - col_r).max(1)
    print(f"[deep ch{channels} split={split}] deep tiles {int((hint_r >= 256).sum())}; max|deep - per-tile| {d.max():.2e}, "
          f"pixels > {PIX_TOL:g}: {int((d > PIX_TOL).sum())}; depth differs on {int((dep_d != dep_r).sum())}")
    assert int((d > PIX_TOL).sum()) <= max(2, int(1e-4 * d.size)) and d.max() <= FLIP_TOL
    assert int((dep_d != dep_r).sum()) <= max(4, int(1e-3 * dep_r.size))
    for v, cam in enumerate(cams):  # and against the oracle
        kw = scene_kwargs(g, cam, W, H, 0.8)
        f = oracle.forward(kw["means3D"], kw["opacities"], bg, kw["view"], kw["proj"], kw["campos"], W, H, kw["tanx"],
                           kw["tany"], channels=channels, colors_precomp=g["colors"], scales=g["scales"],
                           rotations=g["rotations"])
        do = np.abs(col_d[v].astype(np.float64) - f["color"]).max(0)
        assert int((do > PIX_TOL).sum()) <= max(2, int(1e-4 * do.size)) and do.max() <= FLIP_TOL, f"view {v}: {do.max():.2e}"
    names = ("dL_dmeans3D", "dL_dopacity", "dL_dcolors", "dL_dscales", "dL_drotations")
    bad = []
    for n, a, b in zip(names, gr_d, gr_r):
        lim = n_dyn if split else a.shape[0]
        ok, msg = mixed_bound_report(b[:lim], a[:lim])
        print(f"[deep ch{channels} split={split}] {n}: {msg}")
        if not ok:
            bad.append(n + ": " + msg)
    assert not bad, "\n".join(bad)
