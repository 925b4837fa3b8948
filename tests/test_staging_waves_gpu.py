"""-m gpu: the blend forward with staging waves (csrc/raster_forward_ws.h, deep_kernel = 3: the tiles that went deep in the
previous forward, beside the per-tile kernel; = 4: every tile) against the per-tile kernel.

Per pixel the staging-wave kernel runs the per-tile kernel's arithmetic in the per-tile kernel's order -- it only moves
the staging of a batch to other waves -- so colours, depths, final T, last contributors and the tiles' depth hints are
BIT-EQUAL in both arithmetics (and through the per-tile kernel's own tests therefore equal to the oracle's in the exact
one); gradients agree within the backward's mixed bound (the backward reads the hand-over records, masks and merged ids
this kernel wrote; its work items arrive in another order, so its sums associate differently)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from fluidnexus_amd import synthetic as S  # noqa: E402
from tests.test_fast_math_gpu import mixed_bound_report  # noqa: E402


def _render(g, cams, W, H, bg_np, channels, deep_mode, math_mode, min_depth=256, static_split=False, seed=4, rounds=3):
    import torch
    from fluidnexus_amd import rasterizer
    from fluidnexus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizerViews, StaticBin
    rasterizer.set_blend_math(math_mode)
    rasterizer.set_deep_kernel(deep_mode)
    rasterizer.set_deep_variant(True, min_depth)
    rasterizer.keep_last_blobs(True)
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
        for _ in range(rounds):  # the later forwards see the depth hints of the earlier ones: deep tiles exist
            m2d = torch.zeros(V, P, 3, device="cuda", requires_grad=True)
            out = rz(means3D=leaves["means3D"], means2D=m2d, opacities=leaves["opacities"], colors_precomp=leaves["colors"],
                     scales=leaves["scales"], rotations=leaves["rotations"])
        hint = rz.view_batch.depth_hint(channels)
        dL = torch.from_numpy(np.random.RandomState(seed).normal(size=tuple(out[0].shape)).astype(np.float32)).cuda()
        gl = torch.autograd.grad([out[0]], [leaves["means3D"], leaves["opacities"], leaves["colors"], leaves["scales"],
                                            leaves["rotations"]], grad_outputs=[dL])
        grads = [x.detach().cpu().numpy() for x in gl]
        torch.cuda.synchronize()
        rasterizer.check_status()
        return dict(color=out[0].detach().cpu().numpy(), depth=out[2].detach().cpu().numpy(), hint=hint.cpu().numpy(),
                    walked=rasterizer.walked_entries(), grads=grads)
    finally:
        rasterizer.set_deep_kernel(5)  # the library's default
        rasterizer.set_deep_variant(True, 1024)
        rasterizer.set_blend_math("exact")
        rasterizer.keep_last_blobs(False)


@pytest.mark.parametrize("math_mode", ["exact", "fast"])
@pytest.mark.parametrize("channels,split", [(3, False), (1, False), (3, True)])
@pytest.mark.parametrize("deep_mode", [3, 4])
def test_staging_wave_forward_is_bit_equal_to_the_per_tile_kernel(channels, split, deep_mode, math_mode):
    W = H = 160
    n_dyn = 14_000
    g = S.smoke_scene(n_dyn, 6_000, seed=9, channels=3, ring=True) if channels == 3 else S.plume_gaussians(n_dyn, seed=9, channels=1)
    cams = S.ring_cameras(8, W, H, device="cpu")[3:6]
    bg = np.array([0.05, 0.1, 0.2], np.float32)
    sp = n_dyn if split else False
    a = _render(g, cams, W, H, bg, channels, deep_mode, math_mode, static_split=sp)
    r = _render(g, cams, W, H, bg, channels, 0, math_mode, static_split=sp)
    assert int((r["hint"] >= 256).sum()) > 10, "the scene must have deep tiles"
    for k in ("color", "depth", "hint"):
        x, y = a[k], r[k]
        if x.dtype == np.float32:
            x, y = x.view(np.uint32), y.view(np.uint32)
        assert (x == y).all(), f"{k}: {int((x != y).sum())} of {x.size} words differ from the per-tile kernel's"
    assert a["walked"] == r["walked"] or a["walked"] is None, f"entry counters differ: {a['walked']} vs {r['walked']}"
    names = ("dL_dmeans3D", "dL_dopacity", "dL_dcolors", "dL_dscales", "dL_drotations")
    bad = []
    for n, x, y in zip(names, a["grads"], r["grads"]):
        lim = n_dyn if split else x.shape[0]
        ok, msg = mixed_bound_report(y[:lim], x[:lim])
        if not ok:
            bad.append(n + ": " + msg)
    assert not bad, "\n".join(bad)


def test_staging_wave_forward_on_empty_and_single_batch_tiles():
    """A sparse scene: most tiles are empty or hold a single short batch; every tile goes through the staging-wave kernel."""
    W, H = 200, 120  # ragged: the last tile column / row is cut by the image border
    g = S.random_gaussians(600, seed=3, log_scale=(-4.5, -2.5), channels=3)
    cams = [S.front_camera(W, H, device="cpu")]
    bg = np.array([0.3, 0.2, 0.1], np.float32)
    for math_mode in ("exact", "fast"):
        a = _render(g, cams, W, H, bg, 3, 4, math_mode, rounds=2)
        r = _render(g, cams, W, H, bg, 3, 0, math_mode, rounds=2)
        for k in ("color", "depth"):
            assert (a[k].view(np.uint32) == r[k].view(np.uint32)).all(), k


@pytest.mark.parametrize("math_mode", ["exact", "fast"])
@pytest.mark.parametrize("deep_mode,V", [(4, 2), (3, 3), (5, 1)])
def test_staging_wave_forward_dual_mode(deep_mode, V, math_mode):
    """DUAL mode (the 3-channel image and the 1-channel image of the per-call splats out of one pass, config 5): both images,
    both depths bit-equal to the per-tile kernel's; the positions' gradient within the backward's bound."""
    import torch
    from fluidnexus_amd import rasterizer
    from tests.test_dual_mode_gpu import _renders, _scene
    gm, cams = _scene(V=V, size=160)
    rasterizer.set_blend_math(math_mode)
    rasterizer.set_deep_variant(True, 256)
    try:
        base = gm.render_means_from_nn().detach().clone()
        rng = torch.Generator("cuda").manual_seed(3)
        H, W = int(cams[0].image_height), int(cams[0].image_width)
        dL3 = torch.randn(V, 3, H, W, device="cuda", generator=rng)
        dL1 = torch.randn(V, 1, H, W, device="cuda", generator=rng)
        out = {}
        for mode in (0, deep_mode):
            rasterizer.set_deep_kernel(mode)
            for _ in range(3):  # the later forwards see the depth hints of the earlier ones
                m = base.clone().requires_grad_(True)
                c3, d3, c1, d1 = _renders(gm, cams, m, True)
            g, = torch.autograd.grad([c3, c1], [m], grad_outputs=[dL3, dL1])
            torch.cuda.synchronize()
            rasterizer.check_status()
            out[mode] = (c3.detach(), d3.detach(), c1.detach(), d1.detach(), g)
        a, b = out[0], out[deep_mode]
        for k, name in enumerate(("render", "depth", "render1", "depth1")):
            assert torch.equal(a[k], b[k]), f"{name}: {int((a[k] != b[k]).sum())} values differ from the per-tile kernel's"
        assert float(b[2].abs().max()) > 0.01
        ok, msg = mixed_bound_report(a[4].cpu().numpy(), b[4].cpu().numpy())
        assert ok, msg
    finally:
        rasterizer.set_deep_kernel(5)
        rasterizer.set_deep_variant(True, 1024)
        rasterizer.set_blend_math("exact")


def test_tile_order_reaching_down_to_shallow_tiles_changes_nothing():
    """tile_scan_kernel orders the tiles that went at least `deep_threshold` deep in the previous forward deepest first;
    above 128 of them per view it sorts them with its bitonic network (1 024 keys, one per thread).  A threshold of 32 on a
    512 x 512 plume puts several hundred tiles per view through that path: the images do not notice the order."""
    import torch
    from fluidnexus_amd import rasterizer
    W = H = 512
    n_dyn = 60_000
    g = S.smoke_scene(n_dyn, 20_000, seed=5, channels=3)
    cams = S.arc_cameras(5, W, H, device="cpu")[:4]
    bg = np.array([0.0, 0.0, 0.0], np.float32)
    ref = _render(g, cams, W, H, bg, 3, 0, "fast", min_depth=1 << 20, static_split=n_dyn)  # nothing flagged: the plain tile order
    got = _render(g, cams, W, H, bg, 3, 0, "fast", min_depth=32, static_split=n_dyn)
    n_flagged = int((got["hint"] >= 32).sum())
    assert n_flagged > 4 * 128, f"only {n_flagged} tiles reach the threshold: the bitonic path did not run"
    for k in ("color", "depth", "hint"):
        x, y = got[k], ref[k]
        if x.dtype == np.float32:
            x, y = x.view(np.uint32), y.view(np.uint32)
        assert (x == y).all(), k


def test_more_tiles_than_the_sorted_prefix_holds():
    """An image of 3 072 tiles with a threshold low enough to flag more than 1 024 of them: tile_scan_kernel sorts the first
    1 024 flagged tiles (its bitonic network's width) and keeps the arrival order of the rest; the staging-wave kernel takes
    every tile (one view).  Bit-equal to the per-tile kernel without any flagged tile."""
    W, H = 1024, 768
    n_dyn = 60_000
    g = S.smoke_scene(n_dyn, 20_000, seed=6, channels=3)
    cams = S.arc_cameras(5, W, H, device="cpu")[2:3]
    bg = np.array([0.1, 0.0, 0.2], np.float32)
    ref = _render(g, cams, W, H, bg, 3, 0, "exact", min_depth=1 << 20)
    got = _render(g, cams, W, H, bg, 3, 4, "exact", min_depth=8)
    assert int((got["hint"] >= 8).sum()) > 1024, f"only {int((got['hint'] >= 8).sum())} tiles reach the threshold"
    for k in ("color", "depth", "hint"):
        x, y = got[k], ref[k]
        if x.dtype == np.float32:
            x, y = x.view(np.uint32), y.view(np.uint32)
        assert (x == y).all(), k


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FNX_RANDOM_CASES", "8"))))
def test_staging_wave_forward_random_scenes(seed):
    """Seeded random scenes: image sizes with partial tiles, one to three views, thin (deep lists, nothing saturates) or
    opaque splats, 1 or 3 channels, with and without the static split, either arithmetic, deep tiles only or every tile --
    bit-equal to the per-tile kernel every time."""
    rng = np.random.RandomState(4200 + seed)
    channels = int(rng.choice([1, 3]))
    W, H = int(rng.randint(40, 260)), int(rng.randint(40, 220))
    V = int(rng.randint(1, 4))
    n_dyn = int(rng.choice([300, 4000, 15000]))
    n_stat = int(rng.choice([0, 200, 5000])) if channels == 3 else 0
    math_mode = str(rng.choice(["exact", "fast"]))
    deep_mode = int(rng.choice([3, 4]))
    g = S.smoke_scene(n_dyn, max(n_stat, 1), seed=700 + seed, channels=3) if channels == 3 else S.plume_gaussians(n_dyn, seed=700 + seed, channels=1)
    if rng.rand() < 0.5:  # thin: lists of thousands of entries
        g["opacities"] = rng.uniform(0.004, 0.05, size=g["opacities"].shape).astype(np.float32)
    cams = S.arc_cameras(5, W, H, device="cpu")[:V]
    bg = rng.uniform(0, 0.3, size=3).astype(np.float32)
    sp = n_dyn if (channels == 3 and n_stat > 0) else False
    thr = int(rng.choice([16, 256, 1024]))
    a = _render(g, cams, W, H, bg, channels, deep_mode, math_mode, min_depth=thr, static_split=sp, rounds=3)
    r = _render(g, cams, W, H, bg, channels, 0, math_mode, min_depth=thr, static_split=sp, rounds=3)
    for k in ("color", "depth", "hint"):
        x, y = a[k], r[k]
        if x.dtype == np.float32:
            x, y = x.view(np.uint32), y.view(np.uint32)
        assert (x == y).all(), f"seed {seed} ({W}x{H}, V={V}, ch{channels}, split={bool(sp)}, {math_mode}, mode {deep_mode}): {k}"
    names = ("dL_dmeans3D", "dL_dopacity", "dL_dcolors", "dL_dscales", "dL_drotations")
    for n, x, y in zip(names, a["grads"], r["grads"]):
        lim = n_dyn if sp else x.shape[0]
        ok, msg = mixed_bound_report(y[:lim], x[:lim])
        assert ok, f"seed {seed} {n}: {msg}"
