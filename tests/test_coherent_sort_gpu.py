"""-m gpu: the temporal-coherence depth sort (fnx_raster_opts_t.sort_mode = FNX_SORT_COHERENT) and the per-call options.

The repaired order must be the radix sort's order bit for bit -- (depth bits, id) is a total order, so point_list,
ranges and pixels are compared for equality with a from-scratch call on the same inputs -- whatever happened since the
previous call: small drift (the case it is built for: no fallback), splats entering / leaving the views, a completely
new arrangement, an unseeded state (each repaired by the in-launch full sort and counted).
Reference being replaced: ch3/cuda_rasterizer/rasterizer_impl.cu:259-296 (duplicateWithKeys + cub radix sort)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fluidnexus_amd import synthetic as S  # noqa: E402


class _Views:
    """V cameras over one splat set through the C ABI's *_opts entry points; blobs are fresh per call (as the autograd
    function allocates them), only `state` persists."""

    def __init__(self, cams, W, H, channels=3, fov=0.8):
        from fluidnexus_amd import _lib
        from tests.hip_harness import _t
        self.lib, self._lib = _lib.raster(), _lib
        self.dev = torch.device("cuda")
        self.V, self.W, self.H, self.Cn = len(cams), W, H, channels
        tan = math.tan(fov * 0.5)
        self.tx = (C.c_float * self.V)(*([tan] * self.V))
        self.view = _t(np.stack([c.world_view_transform.numpy().reshape(16) for c in cams]), self.dev)
        self.proj = _t(np.stack([c.full_proj_transform.numpy().reshape(16) for c in cams]), self.dev)
        self.campos = _t(np.stack([c.camera_center.numpy() for c in cams]), self.dev)
        self.bg = torch.tensor([0.1, 0.2, 0.3][:channels], device=self.dev)
        self.state = {}

    def sort_state(self, P):
        if P not in self.state:
            self.state[P] = torch.zeros(self.V * self.lib.fnx_sort_state_bytes(P), dtype=torch.uint8, device=self.dev)
        return self.state[P]

    def counters(self, P):
        out = []
        for v in range(self.V):
            pair = (C.c_uint32 * 3)()
            self._lib.check(self.lib.fnx_sort_state_read(self.state[P].data_ptr(), P, v,
                                                         torch.cuda.current_stream().cuda_stream, pair))
            out.append((int(pair[0]), int(pair[1])))
        return out

    def outliers(self, P):
        out = []
        for v in range(self.V):
            n = C.c_uint32(0)
            self._lib.check(self.lib.fnx_sort_state_outliers(self.state[P].data_ptr(), P, v,
                                                             torch.cuda.current_stream().cuda_stream, C.byref(n)))
            out.append(int(n.value))
        return out

    def render(self, g, sort_mode, with_state=True, blend_math=0):
        from tests.hip_harness import _t, _p, _view
        lib, _lib, dev, V, W, H, Cn = self.lib, self._lib, self.dev, self.V, self.W, self.H, self.Cn
        m, o, col, sc, ro = (_t(g[k], dev) for k in ("means3D", "opacities", "colors", "scales", "rotations"))
        P = m.shape[0]
        st = self.sort_state(P) if with_state else None
        opts = _lib.make_opts(blend_math=blend_math, sort_mode=sort_mode, sort_state=None if st is None else st.data_ptr())
        gb, ib = lib.fnx_geom_bytes(P, W, H), lib.fnx_image_bytes(W, H)
        geom = torch.empty(V * gb, dtype=torch.uint8, device=dev)
        img = torch.empty(V * ib, dtype=torch.uint8, device=dev)
        radii = torch.zeros(V, P, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.fnx_forward_stage1_views_split_opts(
            Cn, V, geom.data_ptr(), img.data_ptr(), P, 0, 0, W, H, _p(m), None, _p(col), _p(o), _p(sc), 1.0, _p(ro), None,
            _p(self.view), _p(self.proj), _p(self.campos), self.tx, self.tx, 0, radii.data_ptr(), None, 0, 0, None,
            C.byref(opts), s))
        n, counts = C.c_int(0), []
        for v in range(V):
            _lib.check(lib.fnx_read_num_rendered(img.data_ptr() + v * ib, W, H, s, C.byref(n)))
            counts.append(int(n.value))
        cap = max(counts) + 5
        bb = lib.fnx_binning_bytes(cap)
        binning = torch.empty(V * bb, dtype=torch.uint8, device=dev)
        color = torch.zeros(V, Cn, H, W, device=dev)
        depth = torch.zeros(V, 1, H, W, device=dev)
        _lib.check(lib.fnx_forward_stage2_views_split_opts(
            Cn, V, geom.data_ptr(), binning.data_ptr(), cap, img.data_ptr(), P, W, H, _p(self.bg), color.data_ptr(),
            depth.data_ptr(), None, None, 0, 0, 0, None, C.byref(opts), s))
        torch.cuda.synchronize()
        for v in range(V):
            assert lib.fnx_read_status(img.data_ptr() + v * ib, W, H, s) == 0
        T = ((W + 15) // 16) * ((H + 15) // 16)
        im_l, b_l = _lib.image_layout(W, H), _lib.binning_layout(cap)
        out = dict(counts=counts, color=color.cpu().numpy(), radii=radii.cpu().numpy(), point_list=[], ranges=[])
        for v in range(V):
            out["point_list"].append(_view(binning[v * bb:(v + 1) * bb], b_l.point_list, counts[v], torch.int32).cpu().numpy())
            out["ranges"].append(_view(img[v * ib:(v + 1) * ib], im_l.ranges, 2 * T, torch.int32).cpu().numpy())
        return out


def _same(a, b, what=""):
    assert a["counts"] == b["counts"], what
    assert (a["radii"] == b["radii"]).all(), what
    for v in range(len(a["counts"])):
        assert (a["ranges"][v] == b["ranges"][v]).all(), f"{what}: ranges of view {v}"
        assert (a["point_list"][v] == b["point_list"][v]).all(), f"{what}: point_list of view {v}"
    assert (a["color"].view(np.uint32) == b["color"].view(np.uint32)).all(), what


def _scene(P, seed):
    return S.random_gaussians(P, seed=seed, box=0.45, log_scale=(-4.8, -2.9), center=(0.34, 0.3, -0.225))


@pytest.mark.parametrize("P", [1, 700, 1024, 1025, 2048, 2049, 5000, 20_000])
def test_coherent_equals_radix_while_the_splats_drift(P):
    from fluidnexus_amd import _lib
    W, H = 112, 96
    cams = S.arc_cameras(3, W, H, device="cpu")
    rv = _Views(cams, W, H)
    g = _scene(P, seed=P)
    rng = np.random.RandomState(7)
    ref = rv.render(g, _lib.FNX_SORT_FULL, with_state=False)
    got = rv.render(g, _lib.FNX_SORT_FULL)  # radix passes that leave the state seeded
    _same(ref, got, "seeding call")
    steps = 8
    for it in range(steps):
        g["means3D"] = (g["means3D"] + rng.normal(size=g["means3D"].shape).astype(np.float32) * 2e-4).astype(np.float32)
        ref = rv.render(g, _lib.FNX_SORT_FULL, with_state=False)
        got = rv.render(g, _lib.FNX_SORT_COHERENT)
        _same(ref, got, f"step {it}")
    # an optimiser-sized drift never needed the in-launch full sort
    assert rv.counters(P) == [(steps, 0)] * 3


@pytest.mark.parametrize("P,K", [(300, 2), (700, 2), (2049, 5), (4097, 30), (5000, 3), (20_000, 1), (20_000, 40), (20_000, 200), (60_000, 120)])
def test_far_travellers_are_merged_in_as_outliers_without_a_full_sort(P, K):
    """While the cloud drifts, K splats per call jump anywhere in it (fringe particles of a later frame do: their
    interpolated velocity is noise) -- among them one to the very front and one to the very back of every view's order.
    They travel in the outlier list (include/fnx_raster.h fnx_sort_state_outliers); the order stays the radix sort's, bit
    for bit, and no call needs the in-launch full sort."""
    from fluidnexus_amd import _lib
    W, H = 112, 96
    cams = S.arc_cameras(3, W, H, device="cpu")
    rv = _Views(cams, W, H)
    g = _scene(P, seed=P + K)
    rng = np.random.RandomState(K)
    rv.render(g, _lib.FNX_SORT_FULL)       # seeds the state (radix order: no samples yet)
    rv.render(g, _lib.FNX_SORT_COHERENT)   # first repair call: writes the samples the outlier test needs
    lo, hi = g["means3D"].min(0), g["means3D"].max(0)
    steps = 6
    for it in range(steps):
        g["means3D"] = (g["means3D"] + rng.normal(size=g["means3D"].shape).astype(np.float32) * 2e-4).astype(np.float32)
        sel = rng.choice(P, size=K, replace=False)
        g["means3D"][sel] = rng.uniform(lo, hi, size=(K, 3)).astype(np.float32)
        c0 = cams[0].camera_center.numpy()
        towards = (g["means3D"].mean(0) - c0) / np.linalg.norm(g["means3D"].mean(0) - c0)
        g["means3D"][sel[0]] = (c0 + towards * 0.3).astype(np.float32)                   # nearest of view 0
        if K > 1:
            g["means3D"][sel[1]] = (c0 + towards * (10.0 + it)).astype(np.float32)       # farthest of every view
        ref = rv.render(g, _lib.FNX_SORT_FULL, with_state=False)
        got = rv.render(g, _lib.FNX_SORT_COHERENT)
        _same(ref, got, f"step {it}")
    assert rv.counters(P) == [(steps + 1, 0)] * 3, rv.counters(P)
    # (a jump can land within reach of the old rank in one of the views: the count is a lower bound, not K per call)
    # (... and a view of fewer than ~1 000 splats is one repair window: nothing there ever counts as out of reach)
    if P >= 4000:
        assert all(n >= steps * max(1, K // 2) - 2 for n in rv.outliers(P)), rv.outliers(P)


def test_more_far_travellers_than_the_outlier_list_holds_cost_a_full_sort_and_nothing_else():
    from fluidnexus_amd import _lib
    W, H, P, K = 112, 96, 20_000, 1500
    cams = S.arc_cameras(2, W, H, device="cpu")
    rv = _Views(cams, W, H)
    g = _scene(P, seed=77)
    rng = np.random.RandomState(5)
    rv.render(g, _lib.FNX_SORT_FULL)
    rv.render(g, _lib.FNX_SORT_COHERENT)
    lo, hi = g["means3D"].min(0), g["means3D"].max(0)
    for it in range(3):
        sel = rng.choice(P, size=K if it < 2 else 5, replace=False)
        g["means3D"][sel] = rng.uniform(lo, hi, size=(sel.size, 3)).astype(np.float32)
        ref = rv.render(g, _lib.FNX_SORT_FULL, with_state=False)
        got = rv.render(g, _lib.FNX_SORT_COHERENT)
        _same(ref, got, f"step {it}")
    assert rv.counters(P) == [(4, 2)] * 2, rv.counters(P)   # the two crowded calls fell back, the third did not


def test_coherent_survives_visibility_changes_without_a_full_sort():
    """Splats that leave the view (off the image, same depth) and come back keep their place in the order: culled splats
    are sorted at their own depth (csrc/raster_binning.hip relative_key), so a visibility change is no displacement."""
    from fluidnexus_amd import _lib
    W, H, P = 112, 96, 12_000
    rv = _Views([S.front_camera(W, H, device="cpu")], W, H)  # looks down -z: moving a splat along x keeps its depth
    g = _scene(P, seed=3)
    rv.render(g, _lib.FNX_SORT_FULL)
    home = g["means3D"].copy()
    rng = np.random.RandomState(1)
    for it in range(6):
        g["means3D"] = (home + rng.normal(size=home.shape).astype(np.float32) * 1e-4).astype(np.float32)
        sel = np.arange(it * 300, it * 300 + 900)
        if it % 2 == 0:
            g["means3D"][sel, 0] += 10.0  # far off the image: culled (empty tile rectangle)
        ref = rv.render(g, _lib.FNX_SORT_FULL, with_state=False)
        got = rv.render(g, _lib.FNX_SORT_COHERENT)
        _same(ref, got, f"step {it}")
        if it % 2 == 0:
            assert (ref["radii"][0][sel] == 0).all()
    assert all(f == 0 for _, f in rv.counters(P)), rv.counters(P)


def test_coherent_heals_itself_after_a_new_arrangement_and_from_an_unseeded_state():
    from fluidnexus_amd import _lib
    W, H, P = 112, 96, 9_000
    cams = S.arc_cameras(2, W, H, device="cpu")
    rv = _Views(cams, W, H)
    g = _scene(P, seed=5)
    # zero-filled state, coherent mode on the very first call: sorted from scratch inside the launch
    ref = rv.render(g, _lib.FNX_SORT_FULL, with_state=False)
    got = rv.render(g, _lib.FNX_SORT_COHERENT)
    _same(ref, got, "unseeded")
    assert rv.counters(P) == [(1, 1)] * 2
    got = rv.render(g, _lib.FNX_SORT_COHERENT)
    _same(ref, got, "same positions again")
    assert rv.counters(P) == [(2, 1)] * 2
    # a new frame: same count, every splat somewhere else
    g2 = _scene(P, seed=6)
    ref = rv.render(g2, _lib.FNX_SORT_FULL, with_state=False)
    got = rv.render(g2, _lib.FNX_SORT_COHERENT)
    _same(ref, got, "new arrangement")
    assert rv.counters(P) == [(3, 2)] * 2
    # depth ties (equal keys: order by id) and a wide depth span (no 27-bit limit in this mode)
    g2["means3D"][:300] = g2["means3D"][300:600]
    g2["means3D"][600:620, 2] = -3.0e5
    g2["scales"][600:620] = 3.0e3
    ref = rv.render(g2, _lib.FNX_SORT_FULL, with_state=False)
    got = rv.render(g2, _lib.FNX_SORT_COHERENT)
    _same(ref, got, "ties + wide span")


def test_opts_struct_is_checked_and_one_shot_requests_do_not_outlive_a_failed_call():
    """a request armed before a call that returns early must not reach a later, unrelated call."""
    from fluidnexus_amd import _lib
    lib = _lib.raster()
    bad = _lib.make_opts()
    bad.size = 12
    s = torch.cuda.current_stream().cuda_stream
    tx = (C.c_float * 1)(1.0)
    img = torch.zeros(lib.fnx_image_bytes(64, 64), dtype=torch.uint8, device="cuda")
    rc = lib.fnx_forward_stage1_views_split_opts(3, 1, None, img.data_ptr(), 0, 0, 0, 64, 64, None, None, None, None, None,
                                                 1.0, None, None, None, None, None, tx, tx, 0, None, None, 0, 0, None,
                                                 C.byref(bad), s)
    assert rc == _lib.FNX_ERR_INVALID_ARG
    coh = _lib.make_opts(sort_mode=_lib.FNX_SORT_COHERENT)  # coherent mode without a state
    rc = lib.fnx_forward_stage1_views_split_opts(3, 1, None, img.data_ptr(), 0, 0, 0, 64, 64, None, None, None, None, None,
                                                 1.0, None, None, None, None, None, tx, tx, 0, None, None, 0, 0, None,
                                                 C.byref(coh), s)
    assert rc == _lib.FNX_ERR_INVALID_ARG
    # a zero-fill request followed by a stage 1 that fails its argument checks ...
    victim = torch.full((64, 3), 7.0, device="cuda")
    _lib.check(lib.fnx_request_zero3(victim.data_ptr()))
    rc = lib.fnx_forward_stage1_views_split(5, 1, None, img.data_ptr(), 0, 0, 0, 64, 64, None, None, None, None, None,
                                            1.0, None, None, None, None, None, tx, tx, 0, None, None, 0, 0, None, s)
    assert rc == _lib.FNX_ERR_INVALID_ARG
    # ... is gone: the next forward (64 splats) leaves the tensor alone
    from fluidnexus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizerViews
    g = S.to_torch(_scene(64, seed=1))
    cam = S.front_camera(64, 64)
    rs = [GaussianRasterizationSettings(64, 64, 0.4, 0.4, torch.zeros(3, device="cuda"), 1.0, cam.world_view_transform,
                                        cam.full_proj_transform, 0, cam.camera_center, False)]
    GaussianRasterizerViews(rs)(means3D=g["means3D"], means2D=torch.zeros(1, 64, 3, device="cuda"),
                                opacities=g["opacities"], colors_precomp=g["colors"], scales=g["scales"],
                                rotations=g["rotations"])
    torch.cuda.synchronize()
    assert (victim == 7.0).all()


def test_two_rasteriser_instances_with_different_options_interleaved_on_two_streams():
    """options are per call.  Instance A (exact arithmetic, full sort) and instance B (fast
    arithmetic, lean geometry, coherent sort) run forward A / forward B / backward A / backward B on two streams; each
    must equal its own run in isolation."""
    from fluidnexus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizerViews
    W = H = 96
    P, V = 6000, 2
    dev = torch.device("cuda")
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    cams = S.arc_cameras(V, W, H)
    tan = math.tan(0.4)
    rs = [GaussianRasterizationSettings(H, W, tan, tan, bg, 1.0, c.world_view_transform, c.full_proj_transform, 0,
                                        c.camera_center, False) for c in cams]
    g = S.to_torch(_scene(P, seed=11))
    seed = torch.randn(V, 3, H, W, device=dev)
    optA = dict(blend_math=0, lean_geometry=0, coherent_sort=0)
    optB = dict(blend_math=1, lean_geometry=1, coherent_sort=1)

    def leaves():
        return {k: g[k].clone().requires_grad_() for k in ("means3D", "opacities", "colors", "scales", "rotations")}

    def fwd(rz, lv):
        return rz(means3D=lv["means3D"], means2D=torch.zeros(V, P, 3, device=dev), opacities=lv["opacities"],
                  colors_precomp=lv["colors"], scales=lv["scales"], rotations=lv["rotations"])[0]

    def alone(opt):
        rz, lv = GaussianRasterizerViews(rs, options=opt), leaves()
        img = fwd(rz, lv)
        grads = torch.autograd.grad(img, list(lv.values()), grad_outputs=seed)
        torch.cuda.synchronize()
        return img.detach().clone(), [x.clone() for x in grads]

    refA, refB = alone(optA), alone(optB)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    rzA, rzB, lA, lB = GaussianRasterizerViews(rs, options=optA), GaussianRasterizerViews(rs, options=optB), leaves(), leaves()
    torch.cuda.synchronize()
    with torch.cuda.stream(sA):
        imgA = fwd(rzA, lA)
    with torch.cuda.stream(sB):
        imgB = fwd(rzB, lB)
    with torch.cuda.stream(sA):
        gA = torch.autograd.grad(imgA, list(lA.values()), grad_outputs=seed)
    with torch.cuda.stream(sB):
        gB = torch.autograd.grad(imgB, list(lB.values()), grad_outputs=seed)
    torch.cuda.synchronize()
    assert torch.equal(imgA, refA[0]) and torch.equal(imgB, refB[0])
    assert not torch.equal(imgA, imgB)  # the two arithmetic modes do differ in the last bits somewhere
    for got, ref in ((gA, refA[1]), (gB, refB[1])):
        for a, b in zip(got, ref):  # equal up to the order of the backward's atomics
            scale = b.abs().max().item() + 1e-20
            assert (a - b).abs().max().item() / scale < 2e-4


def test_full_size_plume_with_far_travellers_stays_bit_equal_to_the_radix_order():
    """BASELINE config 3's per-call splats at full size (200 k plume Gaussians, 5 views at 512 x 512): optimiser-sized
    drift plus 40 splats per call that jump anywhere in the plume; point_list, ranges and pixels equal the radix path's
    on every call and no call takes the in-launch full sort."""
    from fluidnexus_amd import _lib
    W = H = 512
    P = 200_000
    cams = S.arc_cameras(5, W, H, device="cpu")
    rv = _Views(cams, W, H)
    g = S.plume_gaussians(P, seed=0, channels=3)
    rng = np.random.RandomState(2)
    rv.render(g, _lib.FNX_SORT_FULL)
    rv.render(g, _lib.FNX_SORT_COHERENT)
    lo, hi = np.percentile(g["means3D"], 2, axis=0), np.percentile(g["means3D"], 98, axis=0)
    steps = 3
    for it in range(steps):
        g["means3D"] = (g["means3D"] + rng.normal(size=g["means3D"].shape).astype(np.float32) * 1e-5).astype(np.float32)
        sel = rng.choice(P, size=40, replace=False)
        g["means3D"][sel] = rng.uniform(lo, hi, size=(40, 3)).astype(np.float32)
        ref = rv.render(g, _lib.FNX_SORT_FULL, with_state=False)
        got = rv.render(g, _lib.FNX_SORT_COHERENT)
        _same(ref, got, f"step {it}")
        assert min(ref["counts"]) > 400_000
    assert rv.counters(P) == [(steps + 1, 0)] * 5, rv.counters(P)
    assert all(n >= 20 * steps for n in rv.outliers(P)), rv.outliers(P)
