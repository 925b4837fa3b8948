"""-m gpu: HIP rasteriser (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bar (DESIGN.md "Parity"): every forward quantity BIT-EXACT (integers: radii, tiles_touched,
ranges, point_list, n_contrib; floats: means2D, depths, conic, cov3D, colour, depth, final_T);
gradients within fp32 summation-order tolerance of the oracle's double-accumulated sums."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from fluidnexus_amd import synthetic as S  # noqa: E402


def _run_pair(oracle, g, cam, W, H, bg, channels=3, shs=None, sh_degree=0, cov=False, fov=0.8):
    from tests.hip_harness import HipRun, scene_kwargs
    kw = scene_kwargs(g, cam, W, H, fov)
    extra = {}
    if shs is not None:
        extra.update(shs=shs, sh_degree=sh_degree)
    else:
        extra.update(colors_precomp=g["colors"])
    if cov is not False:
        extra.update(cov3D_precomp=cov)
    else:
        extra.update(scales=g["scales"], rotations=g["rotations"])
    f = oracle.forward(kw["means3D"], kw["opacities"], bg, kw["view"], kw["proj"], kw["campos"], W, H, kw["tanx"],
                       kw["tany"], channels=channels, **extra)
    h = HipRun(bg=bg, channels=channels, **kw, **extra)
    return f, h


def _assert_forward_exact(f, h):
    it = h.intermediates()
    assert h.R == f["num_rendered"]
    for k in ("radii", "tiles_touched", "ranges", "point_list", "n_contrib"):
        a, b = it[k].astype(np.int64), f[k].astype(np.int64)
        assert a.shape == b.shape and (a == b).all(), f"{k}: {np.sum(a != b)} mismatches"
    vis = f["radii"] > 0
    for k in ("means2D", "depths", "conic_opacity", "cov3D"):
        a, b = it[k][vis], f[k][vis]
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), f"{k} not bit-exact: max |d| {np.abs(a - b).max()}"
    for k in ("color", "depth", "final_T"):
        a, b = it[k], f[k]
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), f"{k} not bit-exact: max |d| {np.abs(a - b).max()}"
    return it


def _assert_grads_close(go, gh, rtol=2e-4, rel=1e-3, abs_of_max=2e-5):
    """Two bounds per gradient array: the largest error against the largest entry (rtol), and PER ELEMENT
    |err| <= rel |ref| + abs_of_max max|ref| -- an error of rtol * max on an entry a thousand times smaller than the
    maximum would pass the first alone.  The worst element is named on failure."""
    for k, ref in go.items():
        if ref.size == 0:
            continue
        ref64 = ref.astype(np.float64)
        got = gh[k].reshape(ref.shape).astype(np.float64)
        scale = np.abs(ref64).max() + 1e-20
        err = np.abs(got - ref64)
        assert err.max() / scale < rtol, f"{k}: rel-to-max error {err.max() / scale:.3e}"
        bound = rel * np.abs(ref64) + abs_of_max * scale
        i = np.unravel_index(np.argmax(err / bound), ref.shape)
        assert err[i] <= bound[i], (f"{k}: worst element {i}: ref {ref64[i]:.6e} got {got[i]:.6e} err {err[i]:.3e} > "
                                    f"bound {bound[i]:.3e} (max|ref| {scale:.3e})")


@pytest.mark.parametrize("P,W,H,seed", [(64, 32, 32, 0), (1000, 64, 48, 1), (10000, 256, 256, 2), (3000, 100, 70, 3)])
def test_forward_backward_ch3_precomp(oracle, P, W, H, seed):
    g = S.random_gaussians(P, seed=seed, box=0.5, log_scale=(-5.0, -2.5))
    cam = S.front_camera(W, H, device="cpu")
    bg = np.array([0.1, 0.4, 0.9], np.float32)
    f, h = _run_pair(oracle, g, cam, W, H, bg)
    _assert_forward_exact(f, h)
    dL = np.random.RandomState(seed).normal(size=(3, H, W)).astype(np.float32)
    _assert_grads_close(oracle.backward(f, dL), h.backward(dL))


def test_ch1_grey(oracle):
    P, W, H = 5000, 128, 96
    g = S.random_gaussians(P, seed=5, channels=1, log_scale=(-4.5, -2.5))
    cam = S.front_camera(W, H, device="cpu")
    bg = np.array([0.3, 0.0, 0.0], np.float32)  # ch1 reads bg[0] only (forward.cu:370 with C = 1)
    f, h = _run_pair(oracle, g, cam, W, H, bg, channels=1)
    _assert_forward_exact(f, h)
    dL = np.random.RandomState(7).normal(size=(1, H, W)).astype(np.float32)
    _assert_grads_close(oracle.backward(f, dL), h.backward(dL))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colours(oracle, deg):
    P, W, H = 2000, 64, 64
    g = S.random_gaussians(P, seed=11 + deg, log_scale=(-4.5, -2.5))
    shs = (np.random.RandomState(deg).normal(size=(P, 16, 3)) * 0.4).astype(np.float32)
    cam = S.front_camera(W, H, device="cpu")
    bg = np.zeros(3, np.float32)
    f, h = _run_pair(oracle, g, cam, W, H, bg, shs=shs, sh_degree=deg)
    it = _assert_forward_exact(f, h)
    vis = f["radii"] > 0
    assert (it["rgb"][vis].view(np.uint32) == f["rgb"][vis].view(np.uint32)).all()
    assert (it["clamped"][vis] == f["clamped"][vis]).all()
    dL = np.random.RandomState(3).normal(size=(3, H, W)).astype(np.float32)
    _assert_grads_close(oracle.backward(f, dL), h.backward(dL))


def test_cov3d_precomp(oracle):
    P, W, H = 1500, 64, 64
    g = S.random_gaussians(P, seed=21, log_scale=(-4.5, -2.5))
    cam = S.front_camera(W, H, device="cpu")
    bg = np.ones(3, np.float32)
    f0, _ = _run_pair(oracle, g, cam, W, H, bg)
    f, h = _run_pair(oracle, g, cam, W, H, bg, cov=f0["cov3D"].copy())
    _assert_forward_exact(f, h)
    dL = np.random.RandomState(4).normal(size=(3, H, W)).astype(np.float32)
    ref = oracle.backward(f, dL)
    _assert_grads_close(ref, h.backward(dL))
    # positions only (geometry_only = 3) with precomputed covariances: the flush reads them at stride 0
    got = h.backward(dL, geometry_only=3)["dL_dmeans3D"]
    assert np.abs(got - ref["dL_dmeans3D"]).max() <= 2e-4 * np.abs(ref["dL_dmeans3D"]).max()


def test_backward_extension_limit_and_geometry_only(oracle):
    """fnx_rasterize_backward_ex: gradients only for splats below the limit, optionally geometry only --
    the kept entries equal the full backward's, the rest stay zero."""
    P, W, H = 4000, 96, 96
    g = S.random_gaussians(P, seed=31, log_scale=(-4.5, -2.5))
    cam = S.front_camera(W, H, device="cpu")
    f, h = _run_pair(oracle, g, cam, W, H, np.array([0.2, 0.3, 0.4], np.float32))
    dL = np.random.RandomState(5).normal(size=(3, H, W)).astype(np.float32)
    ref = oracle.backward(f, dL)
    lim = 1500
    part = h.backward(dL, grad_splat_limit=lim, geometry_only=1)
    for k in ("dL_dmeans2D", "dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dcov3D", "dL_dconic"):
        r = ref[k].reshape(P, -1)
        got = part[k].reshape(P, -1)
        assert np.abs(got[:lim] - r[:lim]).max() <= 2e-4 * np.abs(r).max(), k
        assert (got[lim:] == 0).all(), k
    assert (part["dL_dopacity"] == 0).all() and (part["dL_dcolors"] == 0).all()
    part2 = h.backward(dL, grad_splat_limit=lim, geometry_only=0)
    for k in ("dL_dopacity", "dL_dcolors"):
        r, got = ref[k].reshape(P, -1), part2[k].reshape(P, -1)
        assert np.abs(got[:lim] - r[:lim]).max() <= 2e-4 * np.abs(r).max(), k
        assert (got[lim:] == 0).all(), k
    # geometry_only = 2 ("fixed positions"): appearance and shape gradients as the full backward's, the sums for the
    # 2D means are not formed
    part3 = h.backward(dL, grad_splat_limit=lim, geometry_only=2)
    for k in ("dL_dopacity", "dL_dcolors", "dL_dconic", "dL_dscales", "dL_drotations", "dL_dcov3D"):
        r, got = ref[k].reshape(P, -1), part3[k].reshape(P, -1)
        assert np.abs(got[:lim] - r[:lim]).max() <= 2e-4 * np.abs(r).max(), k
        assert (got[lim:] == 0).all(), k
    assert (part3["dL_dmeans2D"] == 0).all()


def test_fixed_positions_mode_ch1(oracle):
    """geometry_only = 2 on the 1-channel rasteriser (all splats): opacity / colour / shape gradients against the oracle."""
    P, W, H = 3000, 80, 64
    g = S.random_gaussians(P, seed=32, channels=1, log_scale=(-4.5, -2.5))
    cam = S.front_camera(W, H, device="cpu")
    f, h = _run_pair(oracle, g, cam, W, H, np.array([0.1, 0.0, 0.0], np.float32), channels=1)
    dL = np.random.RandomState(6).normal(size=(1, H, W)).astype(np.float32)
    ref = oracle.backward(f, dL)
    got = h.backward(dL, grad_splat_limit=-1, geometry_only=2)
    for k in ("dL_dopacity", "dL_dcolors", "dL_dconic", "dL_dscales", "dL_drotations"):
        assert np.abs(got[k].reshape(P, -1) - ref[k].reshape(P, -1)).max() <= 2e-4 * np.abs(ref[k]).max(), k
    assert (got["dL_dmeans2D"] == 0).all()


def test_edge_cases(oracle):
    """behind-camera and off-screen splats, opaque stacks that saturate T, a tile list longer
    than one 256-entry staging round, an LDS-overflowing tile (global-memory sort fallback)."""
    W, H = 48, 48
    rng = np.random.RandomState(0)
    n_stack = 6000  # > 4096 pairs in the central tiles -> global sort path; >256 -> multi-round
    xyz = np.concatenate([
        rng.uniform(-0.02, 0.02, size=(n_stack, 3)),            # dense stack at the centre
        rng.uniform(-0.5, 0.5, size=(200, 3)) + [0, 0, 3.0],    # behind the camera (z_view <= 0.2)
        rng.uniform(-0.5, 0.5, size=(200, 3)) + [5.0, 0, 0],    # far off-screen
    ]).astype(np.float32)
    P = xyz.shape[0]
    g = dict(means3D=xyz, scales=np.exp(rng.uniform(-4.5, -3.5, size=(P, 3))).astype(np.float32),
             rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1)),
             opacities=rng.uniform(0.0, 1.0, size=(P, 1)).astype(np.float32),
             colors=rng.uniform(0, 1, size=(P, 3)).astype(np.float32))
    g["opacities"][:50] = 1.0  # hits the 0.99 cap
    g["opacities"][50:60] = 0.0
    cam = S.front_camera(W, H, device="cpu")
    bg = np.array([0.5, 0.5, 0.5], np.float32)
    f, h = _run_pair(oracle, g, cam, W, H, bg)
    assert (f["ranges"][:, 1] - f["ranges"][:, 0]).max() > 4096
    assert f["final_T"].min() < 2e-4  # saturating stack reached the T < 1e-4 stop
    _assert_forward_exact(f, h)
    dL = rng.normal(size=(3, H, W)).astype(np.float32)
    _assert_grads_close(oracle.backward(f, dL), h.backward(dL), rtol=5e-4)


@pytest.mark.parametrize("far,wide", [(3.0, False), (9000.0, False), (40000.0, True), (1.0e7, True)])
def test_depth_span(oracle, far, wide):
    """The depth sort works on keys relative to the nearest visible splat and runs its fourth radix pass only
    when they span >= 2^27 ulps: both regimes, with tied depths (id order must decide)."""
    W, H = 64, 64
    rng = np.random.RandomState(7)
    P = 6000
    dist = np.exp(rng.uniform(math.log(0.25), math.log(far), size=P))
    dist[:200] = dist[200:400]  # exact ties
    dist[-1], dist[-2] = 0.25, far
    lateral = rng.uniform(-0.25, 0.25, size=(P, 2)) * dist[:, None]
    xyz = np.stack([lateral[:, 0], lateral[:, 1], 2.0 - dist], axis=1).astype(np.float32)  # camera at z = +2, looking down -z
    xyz[:200, :2] = xyz[200:400, :2]
    g = dict(means3D=xyz, scales=(np.exp(rng.uniform(-4.5, -3.5, size=(P, 3))) * dist[:, None]).astype(np.float32),
             rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1)),
             opacities=rng.uniform(0.05, 0.6, size=(P, 1)).astype(np.float32),
             colors=rng.uniform(0, 1, size=(P, 3)).astype(np.float32))
    cam = S.front_camera(W, H, device="cpu")
    bg = np.zeros(3, np.float32)
    f, h = _run_pair(oracle, g, cam, W, H, bg)
    keys = f["depths"][f["radii"] > 0].view(np.uint32).astype(np.int64)
    assert ((keys.max() - keys.min()) >> 27 != 0) == wide
    _assert_forward_exact(f, h)
    dL = rng.normal(size=(3, H, W)).astype(np.float32)
    _assert_grads_close(oracle.backward(f, dL), h.backward(dL), rtol=5e-4)


def test_large_image_tile_windows(oracle):
    """More than 2048 tiles: the emission walks the tiles in windows (light rank blocks) and in bands of tile
    rows (heavy ones); a few screen-filling splats make the first rank block heavy."""
    P, W, H = 5000, 1040, 800
    g = S.random_gaussians(P, seed=11, box=0.7, log_scale=(-5.0, -3.0))
    g["scales"][:40] *= 12.0
    g["means3D"][:40, 2] += 0.4  # large and near the camera
    cam = S.front_camera(W, H, device="cpu")
    bg = np.array([0.0, 0.2, 0.1], np.float32)
    f, h = _run_pair(oracle, g, cam, W, H, bg)
    assert ((W + 15) // 16) * ((H + 15) // 16) > 2048 and f["num_rendered"] > 50000
    _assert_forward_exact(f, h)
    dL = np.random.RandomState(11).normal(size=(3, H, W)).astype(np.float32)
    _assert_grads_close(oracle.backward(f, dL), h.backward(dL), rtol=5e-4)


def test_empty_and_capacity(oracle):
    from tests.hip_harness import HipRun, scene_kwargs
    from fluidnexus_amd import _lib
    W, H = 32, 32
    cam = S.front_camera(W, H, device="cpu")
    g = S.random_gaussians(500, seed=9, log_scale=(-4.0, -2.5))
    kw = scene_kwargs(g, cam, W, H)
    bg = np.zeros(3, np.float32)
    # too-small binning capacity: nothing rendered, status says so
    h = HipRun(bg=bg, colors_precomp=g["colors"], scales=g["scales"], rotations=g["rotations"], capacity=16, **kw)
    assert h.R > 16
    assert h.status() == _lib.FNX_ERR_CAPACITY
    # all splats culled: image == background, R == 0
    g2 = dict(g)
    g2["means3D"] = g["means3D"] + np.array([0, 0, 10.0], np.float32)
    kw2 = scene_kwargs(g2, cam, W, H)
    bg2 = np.array([0.25, 0.5, 0.75], np.float32)
    h2 = HipRun(bg=bg2, colors_precomp=g["colors"], scales=g["scales"], rotations=g["rotations"], **kw2)
    assert h2.R == 0 and h2.status() == 0
    col = h2.color.cpu().numpy()
    assert (col == bg2[:, None, None]).all()
    assert (h2.depth.cpu().numpy() == 15.0).all()


def test_mark_visible(oracle):
    import torch
    from diff_gaussian_rasterization_ch3 import GaussianRasterizationSettings, GaussianRasterizer
    cam = S.front_camera(64, 64, device="cuda")
    g = S.random_gaussians(4000, seed=2, box=2.5)
    rs = GaussianRasterizationSettings(64, 64, 0.4, 0.4, torch.zeros(3, device="cuda"), 1.0, cam.world_view_transform,
                                       cam.full_proj_transform, 0, cam.camera_center, False)
    vis = GaussianRasterizer(rs).mark_visible(torch.from_numpy(g["means3D"]).cuda()).cpu().numpy()
    ref = oracle.mark_visible(g["means3D"], cam.world_view_transform.cpu().numpy(), cam.full_proj_transform.cpu().numpy())
    assert (vis == ref).all() and 0 < vis.sum() < vis.size


@pytest.mark.parametrize("channels", [3, 1])
def test_callback_forward_and_plain_backward_entry_points(oracle, channels):
    """The two entry points that mirror CudaRasterizer::Rasterizer::forward / ::backward one to one
    (rasterizer.h:30-83): fnx_rasterize_forward with the three resize callbacks (it synchronises once to size the
    binning buffer, like the reference's cudaMemcpy of num_rendered) and fnx_rasterize_backward with R = the forward's
    return value -- the functions a maintainer would bind in place of the reference's."""
    import ctypes as C

    import torch
    from fluidnexus_amd import _lib
    from tests.hip_harness import _p, _t, scene_kwargs
    lib = _lib.raster()
    dev = torch.device("cuda")
    P, W, H = 5000, 112, 80
    g = S.random_gaussians(P, seed=6, log_scale=(-4.5, -2.5), channels=channels)
    cam = S.front_camera(W, H, device="cpu")
    kw = scene_kwargs(g, cam, W, H, 0.8)
    bg = np.array([0.3, 0.1, 0.6], np.float32)
    f = oracle.forward(kw["means3D"], kw["opacities"], bg, kw["view"], kw["proj"], kw["campos"], W, H, kw["tanx"],
                       kw["tany"], channels=channels, colors_precomp=g["colors"], scales=g["scales"], rotations=g["rotations"])
    t = {k: _t(v, dev) for k, v in dict(means3D=kw["means3D"], opac=kw["opacities"], bg=bg, view=kw["view"], proj=kw["proj"],
                                         campos=kw["campos"], colors=g["colors"], scales=g["scales"], rots=g["rotations"]).items()}
    held = {}  # the buffers the callbacks hand out (the reference's resizeFunctional keeps torch tensors alive the same way)

    def make_alloc(name):
        def alloc(nbytes, _user):
            held[name] = torch.zeros(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
            return held[name].data_ptr()
        return _lib.ALLOC_FN(alloc)

    cbs = [make_alloc(n) for n in ("geom", "binning", "image")]
    color = torch.zeros(channels, H, W, device=dev)
    depth = torch.zeros(1, H, W, device=dev)
    radii = torch.zeros(P, dtype=torch.int32, device=dev)
    R = C.c_int(-1)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.fnx_rasterize_forward(
        channels, cbs[0], None, cbs[1], None, cbs[2], None, P, 0, 0, _p(t["bg"]), W, H, _p(t["means3D"]), None, _p(t["colors"]),
        _p(t["opac"]), _p(t["scales"]), 1.0, _p(t["rots"]), None, _p(t["view"]), _p(t["proj"]), _p(t["campos"]), kw["tanx"],
        kw["tany"], 0, color.data_ptr(), depth.data_ptr(), radii.data_ptr(), s, C.byref(R)))
    torch.cuda.synchronize()
    assert R.value == f["num_rendered"] and set(held) == {"geom", "binning", "image"}
    assert (radii.cpu().numpy() == f["radii"]).all()
    assert (color.cpu().numpy().view(np.uint32) == f["color"].view(np.uint32)).all(), "colour not bit-exact"
    assert (depth.cpu().numpy().view(np.uint32) == f["depth"].view(np.uint32)).all()
    dL = np.random.RandomState(1).normal(size=(channels, H, W)).astype(np.float32)
    go = oracle.backward(f, dL)
    z = lambda *shape: torch.zeros(*shape, device=dev)  # noqa: E731
    gr = dict(dL_dmeans2D=z(P, 3), dL_dconic=z(P, 2, 2), dL_dopacity=z(P, 1), dL_dcolors=z(P, channels), dL_dmeans3D=z(P, 3),
              dL_dcov3D=z(P, 6), dL_dscales=z(P, 3), dL_drotations=z(P, 4))
    dLt = _t(dL, dev)
    _lib.check(lib.fnx_rasterize_backward(
        channels, P, 0, 0, R.value, _p(t["bg"]), W, H, _p(t["means3D"]), None, _p(t["colors"]), _p(t["scales"]), 1.0,
        _p(t["rots"]), None, _p(t["view"]), _p(t["proj"]), _p(t["campos"]), kw["tanx"], kw["tany"], radii.data_ptr(),
        held["geom"].data_ptr(), held["binning"].data_ptr(), held["image"].data_ptr(), dLt.data_ptr(),
        gr["dL_dmeans2D"].data_ptr(), gr["dL_dconic"].data_ptr(), gr["dL_dopacity"].data_ptr(), gr["dL_dcolors"].data_ptr(),
        gr["dL_dmeans3D"].data_ptr(), gr["dL_dcov3D"].data_ptr(), None, gr["dL_dscales"].data_ptr(),
        gr["dL_drotations"].data_ptr(), s))
    torch.cuda.synchronize()
    _assert_grads_close({k: v for k, v in go.items() if k in gr}, {k: v.cpu().numpy() for k, v in gr.items()})


@pytest.mark.parametrize("channels", [3, 1])
def test_ball_style_scene_both_rasterisers(oracle, channels):
    """BASELINE config 5 at reduced size, one camera of the 8-camera ring: the 3-channel rasteriser sees fluid +
    background wall, the 1-channel one the fluid alone (both run per view in that configuration).  A deep,
    semi-transparent plume in front of a wall: many batches per tile, block lists of very different lengths."""
    W = H = 160
    if channels == 3:
        g = S.smoke_scene(14_000, 6_000, seed=9, channels=3, ring=True)
    else:
        g = S.plume_gaussians(14_000, seed=9, channels=1)
    cam = S.ring_cameras(8, W, H, device="cpu")[3]
    bg = np.array([0.05, 0.1, 0.2], np.float32)
    f, h = _run_pair(oracle, g, cam, W, H, bg, channels=channels)
    it = _assert_forward_exact(f, h)
    assert int(it["n_contrib"].max()) > 600  # the plume is deep: several 256-entry batches per tile
    dL = np.random.RandomState(4).normal(size=(channels, H, W)).astype(np.float32)
    _assert_grads_close(oracle.backward(f, dL), h.backward(dL))


def test_non_finite_colour_is_a_stated_deviation(oracle):
    """Inputs are assumed finite: the blends multiply the colour of an entry a pixel does NOT take by
    alpha = 0 instead of skipping it, so a NaN colour also reaches the other pixels of the 4x4 blocks the splat's
    footprint touches, where the reference leaves the pixel alone.  This pins the extent of the deviation: every
    pixel the reference would poison is poisoned, every other NaN lies within a block of one of those, and all
    remaining pixels are bit-identical."""
    P, W, H = 400, 64, 64
    g = S.random_gaussians(P, seed=12, box=0.4, log_scale=(-4.2, -3.2))
    g["colors"][7] = np.nan
    cam = S.front_camera(W, H, device="cpu")
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    f, h = _run_pair(oracle, g, cam, W, H, bg)
    ref, got = f["color"], h.intermediates()["color"]
    nan_ref, nan_got = np.isnan(ref).any(0), np.isnan(got).any(0)
    assert nan_ref.sum() > 0 and (nan_got | ~nan_ref).all()          # superset of the reference's poisoned pixels
    ys, xs = np.nonzero(nan_got & ~nan_ref)
    ry, rx = np.nonzero(nan_ref)
    for y, x in zip(ys, xs):                                          # extras: same 4x4 block neighbourhood only
        assert (np.maximum(np.abs(ry - y), np.abs(rx - x)) <= 4).any(), (y, x)
    ok = ~nan_got
    assert (got[:, ok].view(np.uint32) == ref[:, ok].view(np.uint32)).all()


@pytest.mark.parametrize("channels", [3, 1])
def test_positions_only_mode(oracle, channels):
    """geometry_only = 3: the blend backward's flush runs the per-(splat, view) geometry backward itself and adds into
    dL/dmeans3D -- against the oracle's full backward, with and without a gradient limit."""
    P, W, H = 4000, 96, 80
    g = S.random_gaussians(P, seed=41, channels=channels, log_scale=(-4.5, -2.5))
    cam = S.front_camera(W, H, device="cpu")
    f, h = _run_pair(oracle, g, cam, W, H, np.array([0.2, 0.3, 0.4], np.float32), channels=channels)
    dL = np.random.RandomState(8).normal(size=(channels, H, W)).astype(np.float32)
    ref = oracle.backward(f, dL)["dL_dmeans3D"].reshape(P, 3)
    for lim in (-1, 1500):
        got = h.backward(dL, grad_splat_limit=lim, geometry_only=3)["dL_dmeans3D"].reshape(P, 3)
        n = P if lim < 0 else lim
        assert np.abs(got[:n] - ref[:n]).max() <= 2e-4 * np.abs(ref).max(), lim
        assert (got[n:] == 0).all()


def test_backward_refuses_a_capacity_that_is_not_the_forwards(oracle):
    """The binning blob's layout depends on the capacity stage 2 ran with: a backward call that names
    another one must not read the hand-over records at wrong offsets -- it produces no gradients and leaves
    FNX_ERR_CAPACITY in the view's status word; the right capacity still works afterwards."""
    from fluidnexus_amd import _lib
    P, W, H = 800, 64, 64
    g = S.random_gaussians(P, seed=41, log_scale=(-4.5, -2.5))
    cam = S.front_camera(W, H, device="cpu")
    f, h = _run_pair(oracle, g, cam, W, H, np.array([0.2, 0.3, 0.4], np.float32))
    dL = np.random.RandomState(1).normal(size=(3, H, W)).astype(np.float32)
    good_cap = h.cap
    h.cap = good_cap + 256
    bad = h.backward(dL)
    assert all(not v.any() for v in bad.values()), "a mismatching capacity must produce no gradients"
    assert h.status() == _lib.FNX_ERR_CAPACITY
    # a fresh forward clears the status; the matching capacity gives the oracle's gradients
    f, h = _run_pair(oracle, g, cam, W, H, np.array([0.2, 0.3, 0.4], np.float32))
    assert h.cap == good_cap and h.status() == _lib.FNX_OK
    _assert_grads_close(oracle.backward(f, dL), h.backward(dL))


def test_forward_rejects_lists_longer_than_the_work_item_packing_allows():
    """Backward work items hold the batch index in 18 bits: stage 2 refuses a capacity above 2^26 instances."""
    import torch
    from fluidnexus_amd import _lib
    lib = _lib.raster()
    z = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    rc = lib.fnx_forward_stage2(3, z.data_ptr(), z.data_ptr(), (1 << 26) + 1, z.data_ptr(), 10, 64, 64, z.data_ptr(),
                                z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), None)
    assert rc == 5 and b"2^26" in lib.fnx_last_error()  # FNX_ERR_UNSUPPORTED
