"""CPU tests: the package's Python helpers vs golden vectors generated from the reference itself
(tests/golden/gen_reference_golden.py)."""
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def test_graphics_utils():
    from fluidnexus_amd.utils import graphics_utils as gu
    g = np.load(os.path.join(G, "graphics_utils.npz"))
    for i in range(4):
        fovx, fovy = g[f"fov{i}"]
        w2v = gu.get_world_2_view2(g[f"R{i}"], g[f"T{i}"], g[f"trans{i}"], float(g[f"scale{i}"]))
        assert np.array_equal(w2v, g[f"w2v{i}"])
        assert np.array_equal(gu.get_projection_matrix(0.01, 100.0, fovx, fovy).numpy(), g[f"proj{i}"])
        pcv = gu.get_projection_matrix_cv(0.01, 100.0, fovx, fovy, cx=0.1 * i - 0.15, cy=0.05 * i).numpy()
        assert np.allclose(pcv, g[f"projcv{i}"], rtol=1e-6, atol=1e-7)
        assert np.allclose([gu.fov2focal(fovx, 512), gu.focal2fov(600.0 + i, 512)], g[f"focal{i}"], rtol=1e-12)


def test_camera_matches_reference_convention():
    from fluidnexus_amd.scene.camera import Camera
    g = np.load(os.path.join(G, "graphics_utils.npz"))
    fovx, fovy = g["fov1"]
    cam = Camera(g["R1"], g["T1"], fovx, fovy, 64, 48, trans=g["trans1"], scale=float(g["scale1"]), device="cpu")
    assert np.array_equal(cam.world_view_transform.numpy(), g["w2v1"].T)
    assert np.array_equal(cam.projection_matrix.numpy(), g["proj1"].T)
    assert np.allclose(cam.full_proj_transform.numpy(), g["w2v1"].T @ g["proj1"].T, rtol=1e-6, atol=1e-7)
    c2w = np.linalg.inv(g["w2v1"].astype(np.float64))
    assert np.allclose(cam.camera_center.numpy(), c2w[:3, 3], atol=1e-5)


def test_sh_utils():
    from fluidnexus_amd.utils.sh_utils import eval_sh, rgb2sh, sh2rgb
    g = np.load(os.path.join(G, "sh_utils.npz"))
    sh, dirs = torch.tensor(g["sh"]), torch.tensor(g["dirs"])
    for deg in range(4):
        s = sh.clone().requires_grad_(True)
        rgb = torch.clamp_min(eval_sh(deg, s.transpose(1, 2), dirs) + 0.5, 0.0)
        assert np.allclose(rgb.detach().numpy(), g[f"rgb{deg}"], rtol=1e-6, atol=1e-6)
        (rgb * torch.tensor(g[f"w{deg}"])).sum().backward()
        assert np.allclose(s.grad.numpy(), g[f"dsh{deg}"], rtol=1e-5, atol=1e-6)
    assert np.allclose(rgb2sh(torch.tensor([0.1, 0.5, 0.9])).numpy(), g["rgb2sh"])
    assert np.allclose(sh2rgb(torch.tensor([-1.0, 0.0, 1.0])).numpy(), g["sh2rgb"])


def test_loss_utils():
    from fluidnexus_amd.utils import loss_utils as lu
    from fluidnexus_amd.utils.image_utils import psnr
    g = np.load(os.path.join(G, "loss_utils.npz"))
    for tag in "abc":
        y = torch.tensor(g[f"y_{tag}"])
        for nm, fn in (("l1", lu.l1_loss), ("l2", lu.l2_loss), ("ssim", lu.ssim)):
            x = torch.tensor(g[f"x_{tag}"], requires_grad=True)
            v = fn(x, y)
            v.backward()
            assert abs(v.item() - g[f"{nm}_{tag}"]) < 2e-6, (nm, tag)
            assert np.allclose(x.grad.numpy(), g[f"d{nm}_{tag}"], rtol=1e-4, atol=1e-8), (nm, tag)
        assert np.allclose(psnr(torch.tensor(g[f"x_{tag}"]), y).numpy(), g[f"psnr_{tag}"], rtol=1e-6)
    pos = torch.tensor(g["dist_pos"], requires_grad=True)
    v = lu.distance_loss(pos, float(g["dist_thr"]))
    v.backward()
    assert abs(v.item() - g["dist"]) < 1e-5 * abs(g["dist"]) + 1e-7
    assert np.allclose(pos.grad.numpy(), g["ddist"], rtol=1e-4, atol=1e-6)
    assert abs(lu.l2_loss_consistency(torch.tensor(g["cons_a"]), torch.tensor(g["cons_b"])).item() - g["cons"]) < 1e-6


def test_general_utils():
    from fluidnexus_amd.utils.general_utils import get_expon_lr_func, inv_sigmoid
    g = np.load(os.path.join(G, "general_utils.npz"))
    f = get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    h = get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_steps=100, lr_delay_mult=0.01, max_steps=1000)
    assert np.allclose([f(int(s)) for s in g["steps"]], g["lr_a"], rtol=1e-12)
    assert np.allclose([h(int(s)) for s in g["steps"]], g["lr_b"], rtol=1e-12)
    assert np.allclose(inv_sigmoid(torch.tensor(g["inv_sigmoid_x"])).numpy(), g["inv_sigmoid"])
