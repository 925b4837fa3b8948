"""CPU known-answer / self-consistency tests of the rasteriser oracle: the
reference ships no tests, so the restatement is pinned by closed forms, structural invariants and an
independent fp64 autograd formulation of the same maths."""
import math

import numpy as np
import pytest

from fluidnexus_amd import synthetic as S


def _cam(W, H):
    cam = S.front_camera(W, H, device="cpu")
    return cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), cam.camera_center.numpy()


def test_expf_accuracy(oracle):
    x = np.concatenate([np.linspace(-87, 0, 4001), -np.logspace(-6, 1.9, 500)]).astype(np.float32)
    got = oracle.expf(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    ulp = np.abs(got - ref) / np.spacing(ref.astype(np.float32)).astype(np.float64)
    assert ulp.max() <= 1.0, ulp.max()
    assert oracle.expf(np.array([-88.0], np.float32))[0] == 0.0 and oracle.expf(np.array([0.0], np.float32))[0] == 1.0


def test_single_isotropic_gaussian_closed_form(oracle):
    """One isotropic splat on the optical axis: pixel = c * min(.99, o * exp(-d^2 / (2 var))) + T * bg with
    var = (focal * s / z)^2 + 0.3 (forward.cu:105-106) and the centre at ((W - 1) / 2, (H - 1) / 2)."""
    W = H = 32
    view, proj, campos = _cam(W, H)
    tan = math.tan(0.4)
    s, o, z = 0.05, 0.8, 2.0
    col = np.array([[0.2, 0.5, 0.9]], np.float32)
    bg = np.array([0.1, 0.1, 0.3], np.float32)
    f = oracle.forward(np.zeros((1, 3), np.float32), np.array([[o]], np.float32), bg, view, proj, campos, W, H, tan,
                       tan, colors_precomp=col, scales=np.full((1, 3), s, np.float32),
                       rotations=np.array([[1, 0, 0, 0]], np.float32))
    focal = W / (2 * tan)
    var = (focal * s / z) ** 2 + 0.3
    assert f["radii"][0] == math.ceil(3 * math.sqrt(var))
    assert abs(f["depths"][0] - z) < 1e-6
    yy, xx = np.mgrid[0:H, 0:W]
    d2 = (xx - (W - 1) / 2) ** 2 + (yy - (H - 1) / 2) ** 2
    alpha = np.minimum(0.99, o * np.exp(-0.5 * d2 / var))
    alpha[alpha < 1 / 255] = 0
    exp = col[0][:, None, None] * alpha + (1 - alpha) * bg[:, None, None]
    assert np.abs(f["color"] - exp).max() < 2e-6
    assert (f["n_contrib"] == (alpha > 0)).all()
    assert np.allclose(f["final_T"], 1 - alpha, atol=1e-6)
    assert (f["depth"] == 15.0).all() or np.isclose(f["depth"][0][alpha > 0.5], z).all()


@pytest.mark.parametrize("P,W,H", [(500, 64, 48), (3000, 100, 70)])
def test_structural_invariants(oracle, P, W, H):
    g = S.random_gaussians(P, seed=P, log_scale=(-4.5, -2.5))
    view, proj, campos = _cam(W, H)
    tan = math.tan(0.4)
    f = oracle.forward(g["means3D"], g["opacities"], np.zeros(3, np.float32), view, proj, campos, W, H, tan, tan,
                       colors_precomp=g["colors"], scales=g["scales"], rotations=g["rotations"])
    R = f["num_rendered"]
    assert int(f["tiles_touched"].sum()) == R == int(f["point_offsets"][-1])
    assert (np.diff(f["keys_sorted"].astype(np.uint64)) >= 0).all() if R > 1 else True
    rg = f["ranges"].astype(np.int64)
    ne = rg[rg[:, 1] > rg[:, 0]]
    assert ne[0, 0] == 0 and ne[-1, 1] == R and (ne[1:, 0] == ne[:-1, 1]).all()  # partition of [0, R)
    tiles = (f["keys_sorted"] >> np.uint64(32)).astype(np.int64)
    for t in np.unique(tiles):
        assert (tiles[rg[t, 0]:rg[t, 1]] == t).all()
    gx = (W + 15) // 16
    ys, xs = np.mgrid[0:H, 0:W]
    tile_of_px = (ys // 16) * gx + xs // 16
    assert (f["n_contrib"] <= (rg[:, 1] - rg[:, 0])[tile_of_px]).all()
    # depth order inside every tile, ties broken by id
    d = f["depths"][f["point_list"]]
    for t in np.unique(tiles)[:50]:
        seg = slice(rg[t, 0], rg[t, 1])
        key = list(zip(d[seg].view(np.uint32).tolist(), f["point_list"][seg].tolist()))
        assert key == sorted(key)
    # culled splats: radius 0, no tiles
    assert ((f["radii"] > 0) == (f["tiles_touched"] > 0)).all()


@pytest.mark.parametrize("case", ["precomp", "sh", "cov"])
def test_backward_matches_fp64_autograd(oracle, case):
    """Hand-written backward (restating backward.cu) vs autograd of the independent fp64 forward."""
    from oracle import torch_ref as TR
    P, W, H = 60, 48, 32
    g = S.random_gaussians(P, seed=3, box=0.4, log_scale=(-3.0, -1.8))
    g["opacities"] = np.minimum(g["opacities"], 0.95).astype(np.float32)
    view, proj, campos = _cam(W, H)
    tan = math.tan(0.4)
    rng = np.random.RandomState(1)
    kw = dict(scales=g["scales"], rotations=g["rotations"])
    if case == "sh":
        kw.update(shs=(rng.normal(size=(P, 16, 3)) * 0.3).astype(np.float32), sh_degree=3)
    else:
        kw.update(colors_precomp=g["colors"])
    bg = np.array([0.2, 0.5, 0.7], np.float32)
    if case == "cov":
        f0 = oracle.forward(g["means3D"], g["opacities"], bg, view, proj, campos, W, H, tan, tan, **kw)
        kw = dict(colors_precomp=g["colors"], cov3D_precomp=f0["cov3D"].copy())
    f = oracle.forward(g["means3D"], g["opacities"], bg, view, proj, campos, W, H, tan, tan, **kw)
    dL = rng.normal(size=(3, H, W)).astype(np.float32)
    go, gt = oracle.backward(f, dL), TR.gradients(f, dL)
    assert np.abs(gt["color"] - f["color"]).max() < 2e-6
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dcov3D", "dL_dcolors",
              "dL_dsh"):
        if gt.get(k) is None:
            continue
        a, b = go[k].astype(np.float64).reshape(gt[k].shape), gt[k]
        assert np.abs(a - b).max() <= 3e-6 * np.abs(b).max(), k


def test_empty_and_non_rgb(oracle):
    view, proj, campos = _cam(16, 16)
    f = oracle.forward(np.zeros((0, 3), np.float32), np.zeros((0, 1), np.float32), np.zeros(3, np.float32), view, proj,
                       campos, 16, 16, 0.4, 0.4, colors_precomp=np.zeros((0, 3), np.float32))
    assert f["num_rendered"] == 0 and (f["color"] == 0).all()  # rasterize_points.cu:81
    with pytest.raises(RuntimeError):  # rasterizer_impl.cu:226-228
        oracle.forward(np.zeros((4, 3), np.float32), np.ones((4, 1), np.float32), np.zeros(3, np.float32), view, proj,
                       campos, 16, 16, 0.4, 0.4, shs=np.zeros((4, 16, 3), np.float32), channels=1,
                       scales=np.ones((4, 3), np.float32), rotations=np.ones((4, 4), np.float32))
