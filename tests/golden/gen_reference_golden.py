"""Generates the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (it needs /root/reference):
    python tests/golden/gen_reference_golden.py
It imports the importable pure-Python parts of FluidNexus/FluidDynamics on the CPU:
  utils/sh_utils.py (eval_sh), utils/loss_utils.py (l1/l2/ssim/distance_loss/l2_loss_consistency),
  utils/image_utils.py (psnr), utils/graphics_utils.py (camera matrices), utils/general_utils.py
  (get_expon_lr_func, inv_sigmoid), and gaussian_splatting/gm_dynamics.py with the four missing
  third-party modules stubbed (plyfile, simple_knn, torch_cluster = brute-force radius search,
  torch_scatter), and stores seeded inputs + the reference's outputs (+ autograd gradients).
Only DATA is written (npz); no reference source travels.  The clouds are built so that every
particle has fewer than KNN_K neighbours, so the third-party truncation rule cannot matter.
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/FluidDynamics"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
torch.set_num_threads(4)


# ---- stubs for un-vendored third-party packages (torch_cluster 1.6.3 semantics) ----
def _radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32, num_workers=1):
    """row indexes y (queries), col indexes x; all pairs with ||x - y|| < r."""
    d = torch.cdist(y.double(), x.double())
    row, col = torch.nonzero(d < r, as_tuple=True)
    cnt = torch.bincount(row, minlength=y.shape[0])
    assert int(cnt.max()) <= max_num_neighbors, "fixture cloud exceeds max_num_neighbors"
    return torch.stack([row, col], 0)


def _radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target", num_workers=1):
    d = torch.cdist(x.double(), x.double())
    m = d < r
    if not loop:
        m.fill_diagonal_(False)
    col, row = torch.nonzero(m, as_tuple=True)  # row = neighbour/source, col = query/target
    cnt = torch.bincount(col, minlength=x.shape[0])
    assert int(cnt.max()) <= max_num_neighbors + 1, "fixture cloud exceeds max_num_neighbors"
    return torch.stack([row, col], 0)


for name in ("plyfile", "simple_knn", "simple_knn._C", "torch_cluster", "torch_scatter"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["plyfile"].PlyData = object
sys.modules["plyfile"].PlyElement = object
sys.modules["simple_knn._C"].distCUDA2 = None
sys.modules["torch_cluster"].radius = _radius
sys.modules["torch_cluster"].radius_graph = _radius_graph
sys.modules["torch_scatter"].scatter_min = None


def gen_graphics():
    from utils.graphics_utils import (focal2fov, fov2focal, get_projection_matrix, get_projection_matrix_cv,
                                      get_world_2_view2)
    rng = np.random.RandomState(0)
    out = {}
    for i in range(4):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        T = rng.normal(size=3)
        trans = rng.normal(size=3) * (i % 2)
        scale = 1.0 + 0.5 * (i // 2)
        fovx, fovy = 0.5 + 0.2 * i, 0.6 + 0.15 * i
        out[f"R{i}"], out[f"T{i}"], out[f"trans{i}"], out[f"scale{i}"] = R, T, trans, np.float64(scale)
        out[f"fov{i}"] = np.array([fovx, fovy])
        out[f"w2v{i}"] = get_world_2_view2(R, T, trans, scale)
        out[f"proj{i}"] = get_projection_matrix(0.01, 100.0, fovx, fovy).numpy()
        out[f"projcv{i}"] = get_projection_matrix_cv(0.01, 100.0, fovx, fovy, cx=0.1 * i - 0.15, cy=0.05 * i).numpy()
        out[f"focal{i}"] = np.array([fov2focal(fovx, 512), focal2fov(600.0 + i, 512)])
    np.savez(os.path.join(OUT, "graphics_utils.npz"), **out)


def gen_sh():
    from utils.sh_utils import eval_sh, rgb2sh, sh2rgb
    rng = np.random.RandomState(1)
    P = 256
    sh = torch.tensor(rng.normal(size=(P, 16, 3)) * 0.5, dtype=torch.float32)  # kernel layout [P, M, 3]
    dirs = torch.tensor(rng.normal(size=(P, 3)), dtype=torch.float32)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = dict(sh=sh.numpy(), dirs=dirs.numpy())
    for deg in range(4):
        s = sh.clone().requires_grad_(True)
        rgb = torch.clamp_min(eval_sh(deg, s.transpose(1, 2), dirs) + 0.5, 0.0)  # renderer/pipe.py:77-82
        w = torch.tensor(rng.normal(size=(P, 3)), dtype=torch.float32)
        (rgb * w).sum().backward()
        out[f"rgb{deg}"] = rgb.detach().numpy()
        out[f"w{deg}"] = w.numpy()
        out[f"dsh{deg}"] = s.grad.numpy()
    out["rgb2sh"] = rgb2sh(torch.tensor([0.1, 0.5, 0.9])).numpy()
    out["sh2rgb"] = sh2rgb(torch.tensor([-1.0, 0.0, 1.0])).numpy()
    np.savez(os.path.join(OUT, "sh_utils.npz"), **out)


def gen_sh_positions():
    """SH colour as the SH pipe evaluates it (renderer/pipe.py:74-82): directions from the camera centre to the
    Gaussian centres, normalised, eval_sh + 0.5 clamped at 0 -- with autograd gradients with respect to the SH
    coefficients AND the positions (the direction-normalisation path, backward.cu:125-131 / auxiliary.h:95-118)."""
    from utils.sh_utils import eval_sh
    rng = np.random.RandomState(11)
    P = 300
    campos = np.array([0.3, -0.2, 0.1], np.float32)
    d = rng.normal(size=(P, 3))
    pos = (campos + d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.6, 1.0, size=(P, 1))).astype(np.float32)
    sh = (rng.normal(size=(P, 16, 3)) * 0.5).astype(np.float32)
    out = dict(pos=pos, campos=campos, sh=sh)
    for deg in range(4):
        x = torch.tensor(pos, requires_grad=True)
        s = torch.tensor(sh, requires_grad=True)
        dir_pp = x - torch.tensor(campos).repeat(P, 1)
        dirs = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh(deg, s.transpose(1, 2), dirs) + 0.5, 0.0)
        w = torch.tensor(rng.normal(size=(P, 3)), dtype=torch.float32)
        (rgb * w).sum().backward()
        out[f"rgb{deg}"], out[f"w{deg}"] = rgb.detach().numpy(), w.numpy()
        out[f"dsh{deg}"] = s.grad.numpy()
        out[f"dpos{deg}"] = x.grad.numpy() if x.grad is not None else np.zeros_like(pos)  # degree 0 ignores the direction
    np.savez_compressed(os.path.join(OUT, "sh_positions.npz"), **out)


def gen_losses():
    from utils.image_utils import psnr
    from utils.loss_utils import distance_loss, l1_loss, l2_loss, l2_loss_consistency, ssim
    rng = np.random.RandomState(2)
    out = {}
    for tag, (C, H, W) in dict(a=(3, 64, 64), b=(1, 48, 80), c=(3, 37, 53)).items():
        x = torch.tensor(rng.uniform(0, 1, size=(C, H, W)), dtype=torch.float32, requires_grad=True)
        y = torch.tensor(rng.uniform(0, 1, size=(C, H, W)), dtype=torch.float32)
        out[f"x_{tag}"], out[f"y_{tag}"] = x.detach().numpy(), y.numpy()
        for nm, fn in (("l1", l1_loss), ("l2", l2_loss), ("ssim", ssim)):
            x.grad = None
            v = fn(x, y)
            v.backward()
            out[f"{nm}_{tag}"] = np.float32(v.item())
            out[f"d{nm}_{tag}"] = x.grad.numpy().copy()
        out[f"psnr_{tag}"] = psnr(x.detach(), y).numpy()
        # the physical stage's grey-mean image loss (train_physical_particle.py:356-363), lambda_dssim 0.2
        if C == 3:
            x.grad = None
            yg = torch.cat([torch.mean(y, dim=0, keepdim=True)] * 3, dim=0)
            xg = torch.cat([torch.mean(x, dim=0, keepdim=True)] * 3, dim=0)
            v = 0.8 * l1_loss(xg, yg) + 0.2 * (1.0 - ssim(xg, yg))
            v.backward()
            out[f"grey_{tag}"] = np.float32(v.item())
            out[f"dgrey_{tag}"] = x.grad.numpy().copy()
    pos = torch.tensor(rng.uniform(0, 1, size=(200, 3)), dtype=torch.float32, requires_grad=True)
    v = distance_loss(pos, 0.08)
    v.backward()
    out.update(dist_pos=pos.detach().numpy(), dist_thr=np.float32(0.08), dist=np.float32(v.item()), ddist=pos.grad.numpy())
    a = torch.tensor(rng.normal(size=(50, 3)), dtype=torch.float32)
    b = torch.tensor(rng.normal(size=(50, 3)), dtype=torch.float32)
    out.update(cons_a=a.numpy(), cons_b=b.numpy(), cons=np.float32(l2_loss_consistency(a, b).item()))
    np.savez(os.path.join(OUT, "loss_utils.npz"), **out)


def gen_distance():
    """distance_loss (loss_utils.py:98-121) on clouds shaped like the benchmark's visual particles.  The reference
    evaluates torch.cdist in fp32 (matrix-multiply form for N > 25: |x|^2 + |y|^2 - 2 x.y, which loses digits for
    close pairs), so both its fp32 value / gradient and the same function on float64 inputs are stored: the fused
    kernel is compared tightly with the float64 result and the fp32 reference must lie within its own noise of it."""
    from utils.loss_utils import distance_loss
    rng = np.random.RandomState(7)
    out = {}
    clouds = {
        # plume-like: cylinder r 0.1, height 0.6, world units; ~0.3 neighbours within thr on average
        "plume": (np.stack([0.34 + 0.1 * np.sqrt(rng.uniform(size=4000)) * np.cos(t := rng.uniform(0, 2 * np.pi, 4000)),
                            rng.uniform(-0.02, 0.6, 4000), -0.225 + 0.1 * np.sqrt(rng.uniform(size=4000)) * np.sin(t)], 1),
                  0.0125),
        # dense blob in scaled units: many pairs under the threshold, some exactly coincident
        "blob": (np.concatenate([rng.normal(scale=0.6, size=(1500, 3)), np.zeros((3, 3)), np.full((2, 3), 0.25)], 0), 0.2),
        # no pair under the threshold
        "sparse": (rng.uniform(0, 10, size=(300, 3)), 0.01),
    }
    for tag, (xyz, thr) in clouds.items():
        x32 = torch.tensor(xyz, dtype=torch.float32, requires_grad=True)
        x64 = torch.tensor(x32.detach().numpy().astype(np.float64), requires_grad=True)  # the same points
        v32 = distance_loss(x32, thr)
        v32.backward()
        v64 = distance_loss(x64, float(np.float32(thr)))
        v64.backward()
        out[f"pos_{tag}"], out[f"thr_{tag}"] = x32.detach().numpy(), np.float32(thr)
        out[f"loss32_{tag}"], out[f"grad32_{tag}"] = np.float32(v32.item()), x32.grad.numpy()
        out[f"loss64_{tag}"], out[f"grad64_{tag}"] = np.float64(v64.item()), x64.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "distance_loss.npz"), **out)


def gen_general():
    from utils.general_utils import get_expon_lr_func, inv_sigmoid
    out = {}
    f = get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    g = get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_steps=100, lr_delay_mult=0.01, max_steps=1000)
    steps = np.array([0, 1, 10, 100, 500, 999, 1000, 5000, 30000])
    out["steps"] = steps
    out["lr_a"] = np.array([f(int(s)) for s in steps])
    out["lr_b"] = np.array([g(int(s)) for s in steps])
    x = torch.tensor([0.01, 0.1, 0.5, 0.9])
    out["inv_sigmoid_x"], out["inv_sigmoid"] = x.numpy(), inv_sigmoid(x).numpy()
    np.savez(os.path.join(OUT, "general_utils.npz"), **out)


def gen_physics():
    import gaussian_splatting.gm_dynamics as gmd
    rng = np.random.RandomState(3)
    out = {}
    for tag, (n_side, V, buoy_max_y) in dict(a=(6, 300, 0.0), b=(8, 700, 60.0)).items():
        gm = gmd.GaussianModel.__new__(gmd.GaussianModel)
        # hand-set constants (setup_constants allocates on "cuda", gm_dynamics.py:84): values of
        # configs/fluid_nexus_smoke_dynamics.json + arguments/__init__.py:308,312
        gm.H, gm.KNN_K, gm.p0, gm._secs, gm.scale_factor, gm.EPSILON = 2.0, 100, 1.5, 0.033, 100.0, 1e-8
        gm.H2, gm.H6, gm.H9 = gm.H ** 2, gm.H ** 6, gm.H ** 9
        gm.poly6_term1 = 315.0 / (64.0 * np.pi * gm.H9)
        gm.spiky_grad_term1 = 45.0 / (np.pi * gm.H6)
        gm.buoyancy_max_y = buoy_max_y
        N = n_side ** 3
        grid = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3)
        x_prev = (grid * 0.9 + rng.uniform(-0.15, 0.15, size=(N, 3))).astype(np.float32)
        x_est = (x_prev + rng.normal(size=(N, 3)) * 0.05).astype(np.float32)
        x_nn = ((x_est + rng.normal(size=(N, 3)) * 0.03) / 100.0).astype(np.float32)
        gm._xyz = torch.tensor(x_prev)
        gm._estimate_xyz = torch.tensor(x_est)
        gm._imass = torch.tensor(rng.uniform(0.8, 1.2, size=(N, 1)).astype(np.float32))
        gm._buoyancy = torch.tensor(np.tile(np.array([[0.0, 1.96, 0.0]], np.float32), (N, 1)))
        gm._force = torch.tensor((rng.normal(size=(N, 3)) * 0.1).astype(np.float32))
        gm._visual_xyz = torch.tensor(rng.uniform(-1.0, n_side * 0.9 + 1.0, size=(V, 3)).astype(np.float32))
        gm._estimate_xyz_nn = torch.tensor(x_nn, requires_grad=True)
        out.update({f"{k}_{tag}": v for k, v in dict(
            x_prev=x_prev, x_est=x_est, x_nn=x_nn, imass=gm._imass.numpy(), buoyancy=gm._buoyancy.numpy(),
            force=gm._force.numpy(), visual_xyz=gm._visual_xyz.numpy(),
            consts=np.array([gm.H, gm.KNN_K, gm.p0, gm._secs, gm.scale_factor, gm.EPSILON, buoy_max_y])).items()})

        def grad_of(fn, w):
            gm._estimate_xyz_nn.grad = None
            val = fn()
            (val * torch.tensor(w)).sum().backward()
            return val.detach().numpy(), gm._estimate_xyz_nn.grad.numpy().copy()

        w1 = rng.normal(size=(N, 1)).astype(np.float32)
        out[f"w_gas_{tag}"] = w1
        out[f"p_ratio_{tag}"], out[f"d_gas_{tag}"] = grad_of(gm.get_gas_constraints_from_exyz_nn, w1)
        w2 = rng.normal(size=(N, 1)).astype(np.float32)
        out[f"w_next_{tag}"] = w2
        out[f"p_ratio_next_{tag}"], out[f"d_next_{tag}"] = grad_of(gm.get_gas_constraints_from_vel_nn_guess, w2)
        w3 = rng.normal(size=(V, 3)).astype(np.float32)
        out[f"w_vis_{tag}"] = w3
        out[f"vis_{tag}"], out[f"d_vis_{tag}"] = grad_of(gm.get_visual_xyz_from_nn, w3)
        out[f"guess_{tag}"] = gm.get_guess_hidden_particles_from_nn().detach().numpy()
        r2 = torch.tensor(rng.uniform(0, 5.0, size=(64,)).astype(np.float32))
        out[f"poly6_r2_{tag}"], out[f"poly6_{tag}"] = r2.numpy(), gm.poly6(r2).numpy()
        # the physical-stage loss weights of fluid_nexus_smoke_dynamics.json on the same state
        gm._estimate_xyz_nn.grad = None
        from utils.loss_utils import l2_loss
        pr, pn = gm.get_gas_constraints_from_exyz_nn(), gm.get_gas_constraints_from_vel_nn_guess()
        loss = (0.1 * l2_loss(gm._estimate_xyz_nn * gm.scale_factor, gm._estimate_xyz)
                + 1.0 * l2_loss(pr, torch.ones_like(pr)) + 0.1 * l2_loss(pn, torch.ones_like(pn)))
        loss.backward()
        out[f"phys_loss_{tag}"] = np.float32(loss.item())
        out[f"d_phys_loss_{tag}"] = gm._estimate_xyz_nn.grad.numpy().copy()
    np.savez(os.path.join(OUT, "physics.npz"), **out)


def gen_pbf():
    """PBF predictor / solver of one frame step: guess_hidden_particles -> solver_iterations x
    project_gas_constraints -> confirm_guess_hidden_particles -> update_visual_particles, plus the neighbour
    counts of remove_invalid_particles, run by the reference's own gm_dynamics.py on the CPU.  The reference
    hard-codes device="cuda" in a few tensor constructors (gm_dynamics.py:986-1010,1337); those keyword
    arguments / .cuda() calls are redirected to the CPU here, nothing else is touched."""
    import gaussian_splatting.gm_dynamics as gmd

    def to_cpu(fn):
        def wrapped(*a, **k):
            if k.get("device") == "cuda":
                k = dict(k, device="cpu")
            return fn(*a, **k)
        return wrapped

    saved = (torch.ones_like, torch.zeros_like, torch.Tensor.cuda)
    torch.ones_like, torch.zeros_like = to_cpu(torch.ones_like), to_cpu(torch.zeros_like)
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        rng = np.random.RandomState(11)
        out = {}
        for tag, (n_side, V, alpha, buoy_max_y, decay, iters) in dict(a=(6, 400, -1.0, 0.0, 0.0, 3),
                                                                       b=(7, 600, 0.5, 0.6, 0.98, 4)).items():
            gm = gmd.GaussianModel.__new__(gmd.GaussianModel)
            gm.H, gm.KNN_K, gm.p0, gm._secs, gm.scale_factor, gm.EPSILON = 2.0, 100, 1.5, 0.033, 100.0, 1e-8
            gm.H2, gm.H6, gm.H9 = gm.H ** 2, gm.H ** 6, gm.H ** 9
            gm.poly6_term1 = 315.0 / (64.0 * np.pi * gm.H9)
            gm.spiky_grad_term1 = 45.0 / (np.pi * gm.H6)
            gm.k, gm.RELAXATION, gm.K_P, gm.E_P, gm.DQ_P = 3, 0.01, 0.2, 4, 0.25       # gm_dynamics.py:100-111
            gm.lamb_corr_denom = gm.poly6(torch.tensor(gm.DQ_P * gm.DQ_P * gm.H * gm.H))   # :133
            gm.alpha, gm.buoyancy_max_y, gm.buoyancy_decay_rate, gm.min_neighbors = alpha, buoy_max_y, decay, 1
            gm._gravity = torch.tensor([0.0, -9.8, 0.0]).reshape(1, 3)
            N = n_side ** 3
            grid = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3)
            x = (grid * 0.9 + rng.uniform(-0.2, 0.2, size=(N, 3)) + np.array([0.0, 3.0, 0.0])).astype(np.float32)
            x[-1] += 40.0  # two stragglers without neighbours (remove_invalid_particles) ...
            x[-2] += np.array([-50.0, 0.0, 30.0], np.float32)
            x[-3] = x[-4] + np.array([0.3, 0.0, 0.0], np.float32) + 25.0  # ... and a far pair
            x[-4] += 25.0
            gm._xyz = torch.tensor(x)
            gm._estimate_xyz = torch.tensor(x.copy())
            gm._velocity = torch.tensor((rng.normal(size=(N, 3)) * 2.0 + np.array([0.0, 20.0, 0.0])).astype(np.float32))
            gm._force = torch.tensor((rng.normal(size=(N, 3)) * 0.5).astype(np.float32))
            gm._buoyancy = torch.tensor(rng.uniform(0.5, 1.5, size=(N, 3)).astype(np.float32))
            gm._imass = torch.tensor(rng.uniform(0.8, 1.2, size=(N, 1)).astype(np.float32))
            gm._counts = torch.tensor(rng.randint(0, 3, size=(N, 1)).astype(np.float32))
            gm._visual_xyz = torch.tensor((rng.uniform(-1.0, n_side * 0.9 + 1.0, size=(V, 3))
                                           + np.array([0.0, 3.0, 0.0])).astype(np.float32))
            o = dict(xyz0=x, velocity0=gm._velocity.numpy().copy(), force0=gm._force.numpy().copy(),
                     buoyancy0=gm._buoyancy.numpy().copy(), imass=gm._imass.numpy().copy(),
                     counts0=gm._counts.numpy().copy(), visual0=gm._visual_xyz.numpy().copy(),
                     consts=np.array([gm.H, gm.p0, gm._secs, gm.scale_factor, gm.EPSILON, gm.k, gm.RELAXATION, gm.K_P,
                                      gm.E_P, gm.DQ_P, alpha, buoy_max_y, decay, iters], np.float64),
                     gravity=gm._gravity.numpy().copy())
            # neighbour counts of remove_invalid_particles (:1032-1049; edges without self loops).  The reference
            # calls radius_graph with torch_cluster's default max_num_neighbors = 32 there; only
            # `count >= min_neighbors` (= 1) is used, which truncation cannot change, so the fixture stores the
            # untruncated counts and the keep-mask.
            ei = gmd.radius_graph(x=gm._xyz, r=gm.H, loop=False, max_num_neighbors=10 ** 9)
            o["neighbor_counts"] = torch.bincount(ei[0], minlength=N).numpy().astype(np.int32)
            o["keep_mask"] = (o["neighbor_counts"] >= gm.min_neighbors)
            gm.guess_hidden_particles(stable=False, use_wind=False)
            o.update(velocity1=gm._velocity.numpy().copy(), buoyancy1=gm._buoyancy.numpy().copy(),
                     force1=gm._force.numpy().copy(), estimate1=gm._estimate_xyz.numpy().copy(),
                     counts1=gm._counts.numpy().copy())
            for _ in range(iters):
                gm.update_solver_counts()
            o["counts2"] = gm._counts.numpy().copy()
            for it in range(iters):
                ret = gm.project_gas_constraints()
                o[f"estimate_it{it}"] = gm._estimate_xyz.numpy().copy()
                o[f"force_it{it}"] = gm._force.numpy().copy()
                o[f"p_ratio_mean_it{it}"] = np.float64(ret["p_ratio"])
                o[f"lambdas_mean_it{it}"] = np.float64(ret["lambdas"])
            gm.confirm_guess_hidden_particles()
            o.update(xyz3=gm._xyz.numpy().copy(), velocity3=gm._velocity.numpy().copy())
            gm.update_visual_particles()
            o["visual3"] = gm._visual_xyz.numpy().copy()
            out.update({f"{k}_{tag}": v for k, v in o.items()})
        np.savez(os.path.join(OUT, "pbf.npz"), **out)
    finally:
        torch.ones_like, torch.zeros_like, torch.Tensor.cuda = saved


def gen_checkpoint():
    """A per-frame checkpoint written by the reference's own save_hidden / save_visual (gm_dynamics.py:1834-1923)
    from a small seeded state, plus the state itself for the loader test."""
    import shutil
    import gaussian_splatting.gm_dynamics as gmd
    rng = np.random.RandomState(21)
    gm = gmd.GaussianModel.__new__(gmd.GaussianModel)
    N, V = 12, 9
    f = lambda *s: torch.tensor(rng.normal(size=s).astype(np.float32))  # noqa: E731
    gm.scale_factor, gm._secs, gm.alpha, gm.k, gm.p0 = 100.0, 0.033, 0.0, 3, 1.5
    gm.buoyancy_decay_rate, gm.buoyancy_max_y, gm.min_neighbors, gm.remove_out_boundary = 0.98, 0.6, 1, False
    gm.emit_ratio_hidden, gm.emit_ratio_visual, gm.emit_counter = 0.5, 2.0, 7
    gm.total_iterations, gm.total_sim_iterations, gm.total_tb_log_iterations, gm._particle_id_max = 1234, 56, 78, 40
    gm._xyz, gm._estimate_xyz, gm._buoyancy, gm._force, gm._velocity = f(N, 3) * 30, f(N, 3) * 30, f(N, 3), f(N, 3), f(N, 3)
    gm._imass, gm._counts = torch.tensor(rng.uniform(0.8, 1.2, size=(N, 1)).astype(np.float32)), torch.tensor(
        rng.randint(0, 10, size=(N, 1)).astype(np.float32))
    gm._gravity = torch.tensor([[0.0, -9.8, 0.0]])
    gm._particle_id = torch.tensor(rng.permutation(40)[:N].astype(np.int32))
    gm._visual_xyz, gm._visual_color, gm._visual_scales = f(V, 3) * 30, torch.tensor(rng.uniform(size=(V, 1)).astype(np.float32)), f(V, 3)
    gm._visual_rotation, gm._visual_opacity = f(V, 4), f(V, 1)
    d = os.path.join(OUT, "checkpoint")
    shutil.rmtree(d, ignore_errors=True)
    gm.save_hidden(d, 7)
    gm.save_visual(d, 7)
    state = {k: getattr(gm, k).numpy() for k in ("_xyz", "_estimate_xyz", "_buoyancy", "_force", "_velocity", "_imass",
                                                  "_counts", "_gravity", "_particle_id", "_visual_xyz", "_visual_color",
                                                  "_visual_scales", "_visual_rotation", "_visual_opacity")}
    np.savez(os.path.join(OUT, "checkpoint_state.npz"), **state)


def gen_background():
    """Background-stage model (gaussian_splatting/gm_background.py:150-476): optimiser surgery of
    densify / clone / split / prune, densification statistics, opacity reset and the pruning helpers, run by the
    reference's own class on the CPU (device="cuda" keywords / .cuda() redirected as in gen_pbf)."""
    from types import SimpleNamespace
    import gaussian_splatting.gm_background as gmb

    def to_cpu(fn):
        def wrapped(*a, **k):
            if k.get("device") == "cuda":
                k = dict(k, device="cpu")
            return fn(*a, **k)
        return wrapped

    saved = (torch.zeros, torch.ones, torch.Tensor.cuda, torch.cuda.empty_cache)
    torch.zeros, torch.ones = to_cpu(torch.zeros), to_cpu(torch.ones)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    try:
        rng = np.random.RandomState(31)
        N = 300
        init = dict(xyz=rng.uniform(-1, 1, size=(N, 3)), color=rng.uniform(0, 1, size=(N, 3)),
                    opacity=rng.normal(size=(N, 1)) * 2.0, scaling=rng.uniform(-6.0, -2.0, size=(N, 3)),
                    rotation=rng.normal(size=(N, 4)))
        init = {k: v.astype(np.float32) for k, v in init.items()}
        args = SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                               position_lr_delay_mult=0.01, position_lr_max_steps=30000, color_lr=0.0025,
                               opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)
        gm = gmb.GaussianModel()
        gm.spatial_lr_scale = 1.3
        for k, attr in (("xyz", "_xyz"), ("color", "_color"), ("opacity", "_opacity"), ("scaling", "_scaling"),
                        ("rotation", "_rotation")):
            setattr(gm, attr, torch.nn.Parameter(torch.tensor(init[k]).requires_grad_(True)))
        gm.max_radii2D = torch.zeros(N)
        gm.training_setup(args)
        out = {f"init_{k}": v for k, v in init.items()}
        out["lr_xyz_1000"] = np.float64(gm.update_learning_rate(1000))
        g1 = {k: rng.normal(size=v.shape).astype(np.float32) for k, v in init.items()}
        for k, attr in (("xyz", "_xyz"), ("color", "_color"), ("opacity", "_opacity"), ("scaling", "_scaling"),
                        ("rotation", "_rotation")):
            getattr(gm, attr).grad = torch.tensor(g1[k])
            out[f"grad_{k}"] = g1[k]
        gm.optimizer.step()
        gm.optimizer.zero_grad(set_to_none=True)
        for it in range(2):
            vs = torch.zeros(N, 3, requires_grad=True)
            vs.grad = torch.tensor((rng.normal(size=(N, 3)) * 0.0004).astype(np.float32))
            filt = torch.tensor(rng.uniform(size=N) < 0.7)
            out[f"vs_grad{it}"], out[f"filter{it}"] = vs.grad.numpy().copy(), filt.numpy().copy()
            gm.add_densification_stats(vs, filt)
        out["accum"], out["denom"] = gm.xyz_gradient_accum.numpy().copy(), gm.denom.numpy().copy()
        radii = rng.uniform(0, 30, size=N).astype(np.float32)
        gm.max_radii2D = torch.tensor(radii)
        out["max_radii2D"] = radii
        torch.manual_seed(123)
        gm.densify_and_prune(0.0002, 0.005, 2.0, 20)
        out["n_after_densify"] = np.int64(gm.get_xyz.shape[0])
        for k, attr in (("xyz", "_xyz"), ("color", "_color"), ("opacity", "_opacity"), ("scaling", "_scaling"),
                        ("rotation", "_rotation")):
            out[f"dp_{k}"] = getattr(gm, attr).detach().numpy().copy()
        for grp in gm.optimizer.param_groups:
            st = gm.optimizer.state[grp["params"][0]]
            out[f"dp_exp_avg_{grp['name']}"] = st["exp_avg"].numpy().copy()
            out[f"dp_exp_avg_sq_{grp['name']}"] = st["exp_avg_sq"].numpy().copy()
        out["dp_stats_shapes"] = np.array([gm.xyz_gradient_accum.shape[0], gm.denom.shape[0], gm.max_radii2D.shape[0]])
        gm.reset_opacity()
        out["reset_opacity"] = gm._opacity.detach().numpy().copy()
        out["reset_exp_avg_opacity_absmax"] = np.float64(gm.optimizer.state[gm._opacity]["exp_avg"].abs().max())
        gm.prune_large_points()
        out["n_after_large"] = np.int64(gm.get_xyz.shape[0])
        gm._valid_min_y, gm._valid_max_z = -0.2, 0.1
        gm.prune_near_points()
        out["n_after_near"] = np.int64(gm.get_xyz.shape[0])
        cams = rng.uniform(-1.5, 1.5, size=(4, 3)).astype(np.float32)
        out["cams"] = cams
        gm.set_cam_locations(cams)
        out["smoke_to_cams_dist"] = gm.smoke_to_cams_dist.numpy().copy()
        gm.prune_near_cam_points()
        out["n_after_cam"] = np.int64(gm.get_xyz.shape[0])
        out["final_xyz"] = gm._xyz.detach().numpy().copy()
        np.savez(os.path.join(OUT, "background.npz"), **out)
    finally:
        torch.zeros, torch.ones, torch.Tensor.cuda, torch.cuda.empty_cache = saved


def gen_emitter():
    """Frame-boundary bookkeeping of gm_dynamics.GaussianModel (gm_dynamics.py:504-608, 674-976, 1656-1691): the first
    frame's particle clouds, the nozzle lattices, emit_new_particles and the constant render attributes, run by the
    reference's own class on the CPU (device="cuda" keywords redirected as in gen_background).  numpy / torch host
    generators are seeded right before every call that draws from them."""
    from types import SimpleNamespace
    import gaussian_splatting.gm_dynamics as gmd

    def to_cpu(fn):
        def wrapped(*a, **k):
            if k.get("device") == "cuda":
                k = dict(k, device="cpu")
            return fn(*a, **k)
        return wrapped

    names = ("zeros", "ones", "tensor", "arange", "from_numpy")
    saved = tuple(getattr(torch, n) for n in names) + (torch.Tensor.cuda,)
    for n in names:
        setattr(torch, n, to_cpu(getattr(torch, n)))
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        optim = SimpleNamespace(secs=0.033, alpha=-1.5, buoyancy_decay_rate=0.0, buoyancy_max_y=0.0, beta=0.1, H=2.0,
                                min_neighbors=-1, remove_out_boundary=False, p0=1.5, k=3, KNN_K=100,
                                new_hidden_particles_per_sec=15, new_visual_particles_per_sec=15,
                                emitter_points_off_y0=False, emit_ratio_hidden=1.32, emit_ratio_visual=2.4,
                                fit_xyz=False, fit_color=True, fit_opacity=True, fit_scales=True, fit_rotation=True,
                                wind_force=[0.0, 0.0, 0.0], wind_power=1.0, rigid_body="sphere", rigid_particle_radius=0.5,
                                rigid_body_center=[0.0, 0.0, 0.0], rigid_cuboid_num_one_side=2, rigid_cuboid_num=8,
                                rigid_sphere_radius=1.0, rigid_sphere_num=10, rigid_cylinder_radius=1.0, rigid_cylinder_num=10,
                                extra_visual_ratio=0.05, extra_visual_num=7, extra_visual_y_min=0.16, extra_visual_min_num=3,
                                extra_visual_pilar_radius=0.06, extra_visual_pilar_radius_delta=0.0015,
                                pos_lr_scale_factor=1.0, init_hidden_velocity=0.25)
        model = SimpleNamespace(init_visual_num_pts=500, init_thick_visual_num_pts=120, init_x_mid=0.34, init_z_mid=-0.22,
                                init_visual_y_min=-0.05, init_visual_y_max=0.35, init_visual_y_thick_min=0.1,
                                init_visual_radius_small_max=0.03, init_visual_radius_max=0.07,
                                init_hidden_radius_max=0.06, init_hidden_y_min=-0.04, init_hidden_y_max=0.12,
                                init_hidden_delta=0.017, emitter_hidden_delta=0.017, emitter_visual_delta=0.004,
                                emitter_center_y_hidden=-0.05, emitter_center_y_visual=-0.045,
                                emitter_center_y_hidden_max=0.0, emitter_center_y_visual_max=0.0,
                                emitter_visual_radius_ratio=6.3, emitter_hidden_radius_ratio=2.6)
        out = {"optim": np.array(repr(vars(optim))), "model": np.array(repr(vars(model)))}
        gm = gmd.GaussianModel()
        gm.setup_constants(optim)
        np.random.seed(11)
        gm.create_particles_visual(model)
        out["visual_xyz0"] = gm._visual_xyz.numpy().copy()
        gm.create_particles_hidden(model)
        for n in ("xyz", "estimate_xyz", "buoyancy", "force", "velocity", "imass", "counts", "particle_id"):
            out[f"hidden0_{n}"] = getattr(gm, f"_{n}").numpy().copy()
        out["hidden0_id_max"] = np.int64(gm._particle_id_max)
        gm.prepare_emitter_points(model, is_future=True)
        out["emit_future_visual"] = gm.visual_emitter_points.numpy().copy()
        gm.prepare_emitter_points(model)
        out["emit_visual"], out["emit_hidden"] = gm.visual_emitter_points.numpy().copy(), gm.hidden_emitter_points.numpy().copy()
        gm.prepare_emitter_future_first_points(model)
        out["emit_first_visual"] = gm.visual_emitter_first_points.numpy().copy()
        out["emit_first_hidden"] = gm.hidden_emitter_first_points.numpy().copy()
        gm.detach_visual_and_scale()
        gm.prepare_visual_particles_for_rendering()
        for n in ("color", "scales", "rotation", "opacity"):
            out[f"render0_{n}"] = getattr(gm, f"_visual_{n}").numpy().copy()
        torch.manual_seed(5)
        gm.emit_new_particles()
        out["visual_xyz1"] = gm._visual_xyz.numpy().copy()
        for n in ("xyz", "estimate_xyz", "buoyancy", "force", "velocity", "imass", "counts", "particle_id"):
            out[f"hidden1_{n}"] = getattr(gm, f"_{n}").numpy().copy()
        gm.prepare_future_visual_particles_for_rendering(use_level_two_future=True)
        out["render1_shapes"] = np.array([getattr(gm, f"_visual_{n}").shape[0] for n in ("color", "scales", "rotation", "opacity")])
        out["render1_opacity_tail"] = gm._visual_opacity[-5:].numpy().copy()
        torch.manual_seed(6)
        gm.emit_new_particles(future_time_index=1)
        out["visual_xyz2"], out["hidden2_xyz"] = gm._visual_xyz.numpy().copy(), gm._xyz.numpy().copy()
        out["hidden2_id_max"], out["emit_counter"] = np.int64(gm._particle_id_max), np.int64(gm.emit_counter)
        rng = np.random.RandomState(3)
        r = torch.tensor(rng.normal(size=(64, 3)).astype(np.float32) * 1.2)
        r[5] = 0.0
        rlen = torch.norm(r, dim=1)
        out["spiky_r"], out["spiky_grad"] = r.numpy().copy(), gm.spiky_grad(r, rlen).numpy().copy()
        np.savez(os.path.join(OUT, "emitter.npz"), **out)
        # the ScalarReal model's emitter (gm_fluid.py:594-632): no arguments, hard-coded nozzle geometry
        import gaussian_splatting.gm_fluid as gmf
        gf = gmf.GaussianModel()
        gf.emit_ratio_visual, gf.emit_ratio_hidden = 1.0, 1.0  # only printed
        gf.prepare_emitter_points()
        np.savez(os.path.join(OUT, "emitter_fluid.npz"), emit_visual=gf.visual_emitter_points.numpy().copy(),
                 emit_hidden=gf.hidden_emitter_points.numpy().copy())
    finally:
        for n, f in zip(names, saved[:-1]):
            setattr(torch, n, f)
        torch.Tensor.cuda = saved[-1]


def gen_entry_surface():
    """(1) NAMES ONLY of every `gaussians.<attr>` the reference's entry scripts touch (train_physical_particle.py,
    train_visual_particle.py, train_background.py of entries_fluid_nexus / entries_scalar_real; lines that are comments
    are skipped): entry_script_names.json.  (2) The per-frame / per-iteration position dumps those scripts call
    unconditionally (gm_dynamics.py:1753-1831), written by the reference's own class from a small seeded state:
    file names + arrays.  (3) init_quantities_current_level_two (gm_dynamics.py:399-414) on the reference's class with
    simple-knn's distCUDA2 replaced by its definition (mean squared distance to the three nearest other points,
    submodules/simple-knn/simple_knn.cu:134-165) evaluated by brute force in float64, .cuda() redirected."""
    import glob
    import json
    import re
    import shutil
    import tempfile
    from types import SimpleNamespace
    import gaussian_splatting.gm_dynamics as gmd
    names = {}
    for pth in sorted(glob.glob(os.path.join(REF, "entries_fluid_nexus", "train_*.py")) +
                      glob.glob(os.path.join(REF, "entries_scalar_real", "train_*.py"))):
        code = "\n".join(l for l in open(pth).read().splitlines() if not l.lstrip().startswith("#"))
        names[os.path.relpath(pth, REF)] = sorted(set(re.findall(r"\bgaussians\.([A-Za-z_]\w*)", code)))
    with open(os.path.join(OUT, "entry_script_names.json"), "w") as f:
        json.dump(names, f, indent=0, sort_keys=True)
    # (1b) what KIND of thing each name is on the reference's class the script's shipped configuration selects, and for
    # methods their parameter names (with "=" behind those that have a default): a stub attribute, or a method with
    # another signature, must not pass for the real thing.  Names and parameter lists only.
    import inspect
    import gaussian_splatting.gm_background as gmb
    import gaussian_splatting.gm_fluid as gmf
    cls_of = {"entries_fluid_nexus/train_background.py": gmb.GaussianModel,
              "entries_fluid_nexus/train_physical_particle.py": gmd.GaussianModel,
              "entries_fluid_nexus/train_visual_particle.py": gmd.GaussianModel,
              "entries_scalar_real/train_physical_particle.py": gmf.GaussianModel,
              "entries_scalar_real/train_visual_particle.py": gmf.GaussianModel}
    sigs = {}
    for script, cls in cls_of.items():
        ent = {}
        for n in names[script]:
            a = inspect.getattr_static(cls, n, None)
            if isinstance(a, property):
                ent[n] = {"kind": "property"}
            elif inspect.isfunction(a):
                ps = [p for p in inspect.signature(a).parameters.values() if p.name != "self"]
                ent[n] = {"kind": "method", "params": [("**" if p.kind is p.VAR_KEYWORD else "*" if p.kind is p.VAR_POSITIONAL else "")
                                                       + p.name + ("=" if p.default is not inspect.Parameter.empty else "") for p in ps]}
            else:
                ent[n] = {"kind": "attribute"}  # set on the instance (in __init__ or by a set-up method)
        sigs[script] = ent
    with open(os.path.join(OUT, "entry_script_signatures.json"), "w") as f:
        json.dump(sigs, f, indent=0, sort_keys=True)

    rng = np.random.RandomState(31)
    f32 = lambda *s: torch.tensor(rng.normal(size=s).astype(np.float32))  # noqa: E731
    gm = gmd.GaussianModel.__new__(gmd.GaussianModel)
    N, V = 10, 7
    gm.scale_factor = 100.0
    gm._xyz, gm._estimate_xyz, gm._visual_xyz = f32(N, 3) * 30, f32(N, 3) * 30, f32(V, 3) * 30
    gm._estimate_xyz_nn = f32(N, 3) * 0.3
    gm._visual_color, gm._visual_scales = torch.tensor(rng.uniform(size=(V, 1)).astype(np.float32)), f32(V, 3)
    gm._visual_rotation, gm._visual_opacity = f32(V, 4), f32(V, 1)
    other_visual = f32(V, 3)
    out = {k: getattr(gm, k).numpy().copy() for k in ("_xyz", "_estimate_xyz", "_visual_xyz", "_estimate_xyz_nn",
                                                       "_visual_color", "_visual_scales", "_visual_rotation", "_visual_opacity")}
    out["other_visual"] = other_visual.numpy().copy()
    d = tempfile.mkdtemp()
    try:
        gm.save_particles_frame(d, 3)
        gm.save_particles_simulation(d, 4)
        gm.save_particles_simulation_guess(d, 5)
        gm.save_particles_optimization_first(d, 0, 120)
        gm.save_particles_optimization(d, other_visual, 6, 250)
        gm.save_particles_optimization_level_two(d, 7, 999)
        files = sorted(os.listdir(d))
        out["files"] = np.array(files)
        for fn in files:
            out["file:" + fn] = np.load(os.path.join(d, fn))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    np.savez(os.path.join(OUT, "save_particles.npz"), **out)

    def dist2_bruteforce(p):
        d2 = torch.cdist(p.double(), p.double()) ** 2
        d2.fill_diagonal_(float("inf"))
        return d2.topk(3, dim=1, largest=False).values.mean(1).float()

    saved = (gmd.distCUDA2, torch.Tensor.cuda)
    gmd.distCUDA2 = dist2_bruteforce
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        out = {}
        for case, flags in enumerate(((True, True, True, True, True), (False, True, False, True, False))):
            V, Vp = 40, 25
            gm = gmd.GaussianModel.__new__(gmd.GaussianModel)
            gm.fit_color = gm.fit_opacity = gm.fit_scales = gm.fit_rotation = True
            gm._visual_xyz = torch.tensor(rng.uniform(-3, 3, size=(V, 3)).astype(np.float32))
            gm._visual_xyz[3] = gm._visual_xyz[4]  # coincident pair: the clamp_min at 1e-7 is not hit, log of a tiny mean is
            gm._visual_color, gm._visual_opacity = torch.tensor(rng.uniform(size=(V, 1)).astype(np.float32)), f32(V, 1)
            gm._visual_scales, gm._visual_rotation = f32(V, 3), f32(V, 4)
            prev = dict(color=torch.tensor(rng.uniform(size=(Vp, 1)).astype(np.float32)), opacity=f32(Vp, 1),
                        scales=f32(Vp, 3), rotation=f32(Vp, 4))
            oa = SimpleNamespace(init_scales_w_xyz_dist=flags[0], inherit_prev_color=flags[1], inherit_prev_opacity=flags[2],
                                 inherit_prev_scales=flags[3], inherit_prev_rotation=flags[4])
            for k in ("xyz", "color", "opacity", "scales", "rotation"):
                out[f"c{case}_in_{k}"] = getattr(gm, f"_visual_{k}").numpy().copy()
            for k, v in prev.items():
                out[f"c{case}_prev_{k}"] = v.numpy().copy()
            out[f"c{case}_flags"] = np.array(flags)
            gm.init_quantities_current_level_two(oa, prev["color"], prev["opacity"], prev["scales"], prev["rotation"])
            for k in ("color", "opacity", "scales", "rotation"):
                out[f"c{case}_out_{k}"] = getattr(gm, f"_visual_{k}").numpy().copy()
        np.savez(os.path.join(OUT, "level_two_init.npz"), **out)
    finally:
        gmd.distCUDA2, torch.Tensor.cuda = saved


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference is only available in the build container"
    gen_graphics()
    gen_sh()
    gen_sh_positions()
    gen_losses()
    gen_distance()
    gen_general()
    gen_physics()
    gen_pbf()
    gen_checkpoint()
    gen_background()
    gen_emitter()
    gen_entry_surface()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
