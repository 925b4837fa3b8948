"""-m gpu: ONE end-to-end parity test of the iteration bench.py times, with bench.py's own switches all on at once --
fast blend arithmetic, static split, gradient limit, positions-only backward (MODE 3), lean geometry state, narrow /
coherent depth sort, view batching, hipGraph x 5, the list-based distance loss, fnx_adam_step_grid -- against a HOST
composition that uses none of it: the CPU oracle rasteriser per view (oracle/raster_oracle.c, exact arithmetic) +
oracle/physics_oracle.py (brute-force neighbours, float64) + the golden-pinned utils.loss_utils on the host +
torch.optim.Adam(eps = 1e-15).  Every other test reaches the oracle through a chain of HIP-vs-HIP equivalences; this one
closes the chain at its optimised end (VERDICT r3 "weak 2").

Reference: entries_fluid_nexus/train_physical_particle.py:329-432 (zero cache -> per view: render_dynamics
(guess_visual_nn, scale) -> grey-mean L1 + D-SSIM -> distance_loss -> exyz / gas / next-gas terms -> backward ->
cache) -> batch mean -> Adam step)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host_iteration(oracle, po, st, x, cams, cfg, V):
    """d (sum over the views of the per-view loss) / d x on the host; x: float64 leaf [N, 3] (world units)."""
    from fluidnexus_amd.utils.loss_utils import l1_loss, l2_loss, ssim
    from oracle.physics_oracle import distance_loss_oracle
    sf = po.scale_factor
    n_fluid = st["visual_xyz"].shape[0]
    visual = po.visual_xyz_from_nn(x, st["x_prev"], st["visual_xyz"])       # simulation units, [n_fluid, 3]
    render_xyz = visual / sf
    means = np.concatenate([render_xyz.detach().numpy().astype(np.float32), st["gs_xyz"]], 0)
    g_means = np.zeros((n_fluid, 3), np.float64)
    for cam in cams:
        f = oracle.forward(means, st["opacity"], st["bg"], cam["view"], cam["proj"], cam["campos"], st["W"], st["H"],
                           cam["tan"], cam["tan"], colors_precomp=st["colors"], scales=st["scales"],
                           rotations=st["rotations"], channels=3)
        img = torch.tensor(f["color"], dtype=torch.float64, requires_grad=True)
        gt = cam["gt"]
        gt3 = torch.cat([torch.mean(gt, dim=0, keepdim=True)] * 3, dim=0)              # tpp:356-360
        im3 = torch.cat([torch.mean(img, dim=0, keepdim=True)] * 3, dim=0)
        loss = ((1.0 - cfg["lambda_dssim"]) * l1_loss(im3, gt3) + cfg["lambda_dssim"] * (1.0 - ssim(im3, gt3))) * cfg["lambda_image"]
        dimg, = torch.autograd.grad(loss, img)
        g = oracle.backward(f, dimg.numpy().astype(np.float32))
        g_means += g["dL_dmeans3D"][:n_fluid].astype(np.float64)
    # the view-independent terms, once per view (tpp:365-404): distance loss on the rendered positions, physics terms on x
    _, gd = distance_loss_oracle(render_xyz.detach().numpy(), cfg["distance_threshold_visual"])
    g_means += V * cfg["lambda_current_distance"] * gd
    phys = cfg["lambda_exyz"] * l2_loss(x * sf, st["estimate_xyz"])
    pr = po.gas_constraints_from_exyz_nn(x, st["imass"])
    phys = phys + cfg["lambda_gas_constraints"] * l2_loss(pr, torch.ones_like(pr))
    pn = po.gas_constraints_from_vel_nn_guess(x, st["x_prev"], st["imass"], st["buoyancy"], st["force"])
    phys = phys + cfg["lambda_next_gas_constraints"] * l2_loss(pn, torch.ones_like(pn))
    total = V * phys + (render_xyz * torch.from_numpy(g_means)).sum()   # the rasteriser's gradient enters as a cotangent
    gx, = torch.autograd.grad(total, x)
    return gx / V  # set_batch_gradient_current (gm_dynamics.py:461-472)


def _mixed(got, ref, rel, abs_of_max):
    """Worst element of |got - ref| / (rel |ref| + abs_of_max max|ref|) and where."""
    err = (got - ref).abs()
    bound = rel * ref.abs() + abs_of_max * ref.abs().max()
    ratio = err / bound
    k = int(ratio.argmax())
    return float(ratio.flatten()[k]), k, float(got.flatten()[k]), float(ref.flatten()[k])


def test_bench_iteration_against_the_host_composition(oracle):
    from fluidnexus_amd import harness as Hn, rasterizer
    from oracle.physics_oracle import PhysicsOracle
    V, size = 3, 128
    gm, cams = Hn.build_smoke_frame(P_fluid=8000, P_background=3000, hidden_dims=(6, 14, 6), n_views=V, size=size, seed=5)
    cfg = dict(Hn.SMOKE, distance_threshold_visual=0.004)  # a threshold this small cloud has pairs under
    torch.backends.cudnn.enabled = False
    rasterizer.set_host_sync(False)
    rasterizer.set_blend_math("fast")
    rasterizer.set_lean_geometry(True)
    rasterizer.set_coherent_sort(True)
    rasterizer.max_sort_span_bits = 0
    try:
        loop = Hn.HotLoop(gm, cams, image_loss="fused", fused_physics=True, defer_visual_backward=True, capturable=True,
                          batched_views=True, fused_step=True, cfg=cfg)
        loop.make_targets()
        for _ in range(2):
            loop.iteration()
        rasterizer.check_status()
        if 0 < rasterizer.max_sort_span_bits <= rasterizer.SORT_NARROW_MAX_BITS:
            rasterizer.set_sort_narrow(True)
        loop.capture(warmup=1, iterations=5)
        torch.cuda.synchronize()

        def snapshot():
            p = gm._estimate_xyz_nn
            s = gm.optimizer.state[p]
            return dict(x=p.detach().double().cpu().clone(), m=s["exp_avg"].double().cpu().clone(),
                        v=s["exp_avg_sq"].double().cpu().clone(), step=float(s["step"]))

        cpu = lambda t: t.detach().double().cpu()  # noqa: E731
        from fluidnexus_amd.renderer.pipes import _static_attributes
        opac, scales, rots, cols = (t.detach().float().cpu().numpy() for t in _static_attributes(gm, "guess_visual_nn", False))
        st = dict(x_prev=cpu(gm._xyz), visual_xyz=cpu(gm._visual_xyz), estimate_xyz=cpu(gm._estimate_xyz), imass=cpu(gm._imass),
                  buoyancy=cpu(gm._buoyancy), force=cpu(gm._force), gs_xyz=gm._gs_xyz.detach().float().cpu().numpy(),
                  opacity=opac, scales=scales, rotations=rots, colors=cols, bg=np.zeros(3, np.float32), W=size, H=size)
        hc = [dict(view=c.world_view_transform.cpu().numpy(), proj=c.full_proj_transform.cpu().numpy(),
                   campos=c.camera_center.cpu().numpy(), tan=math.tan(c.FoVx * 0.5), gt=c.original_image.double().cpu())
              for c in cams]
        po = PhysicsOracle(H=cfg["H"], p0=cfg["p0"], secs=cfg["secs"], scale_factor=gm.scale_factor,
                           buoyancy_max_y=gm.buoyancy_max_y)
        lr = float(gm.optimizer.param_groups[0]["lr"])

        def host_steps(s0, n):
            x = torch.nn.Parameter(s0["x"].clone())
            opt = torch.optim.Adam([x], lr=lr, eps=1e-15)
            opt.state[x] = dict(step=torch.tensor(s0["step"]), exp_avg=s0["m"].clone(), exp_avg_sq=s0["v"].clone())
            grads = []
            for _ in range(n):
                x.grad = _host_iteration(oracle, po, st, x, hc, cfg, V)
                grads.append(x.grad.clone())
                opt.step()
            return x.detach().clone(), grads

        # ---- five iterations: ONE replay of the captured graph against five host iterations from the same state
        s0 = snapshot()
        loop.iteration()
        torch.cuda.synchronize()
        rasterizer.check_status()
        s5 = snapshot()
        assert s5["step"] == s0["step"] + 5
        x5_ref, g_ref = host_steps(s0, 5)
        # ---- one iteration, eager, same switches: the batch gradient itself (out of Adam's first moment) and the step
        loop.use_graph(False)
        loop.iteration()
        torch.cuda.synchronize()
        s6 = snapshot()
        g_hip = (s6["m"] - 0.9 * s5["m"]) / 0.1
        x6_ref, g6_ref = host_steps(s5, 1)
        assert all(f == 0 for vb in list(rasterizer._VIEW_BATCHES) for k in vb._sort_state for _, f in vb.sort_counters(*k))
    finally:
        rasterizer.set_sort_narrow(False)
        rasterizer.set_coherent_sort(False)
        rasterizer.set_lean_geometry(False)
        rasterizer.set_blend_math("exact")
        rasterizer.set_host_sync(True)

    # the batch gradient, element by element (fast arithmetic: the stated tolerance of tests/test_fast_math_gpu.py)
    worst, k, a, b = _mixed(g_hip, g6_ref[0], 1e-3, 2e-5)
    print(f"[bench iteration] batch gradient: worst element {k}: hip {a:.6e} host {b:.6e}, {worst:.3f} of the mixed bound "
          f"(1e-3 |ref| + 2e-5 max|ref|), max|ref| {float(g6_ref[0].abs().max()):.3e}")
    assert worst <= 1.0
    # positions: Adam (eps 1e-15) turns a gradient component that is summation noise into a step of +-lr, so an element
    # counts only where the host gradient is not noise; everywhere the two may differ by at most the steps taken
    for name, got, ref, n, g0 in (("after 1 step", s6["x"], x6_ref, 1, g6_ref[0]), ("after 5 steps", s5["x"], x5_ref, 5, g_ref[0])):
        moved = (ref - (s5["x"] if n == 1 else s0["x"])).abs()
        err = (got - ref).abs()
        solid = g0.abs() > 1e-3 * g0.abs().max()
        assert float(err.max()) <= 2.0 * n * lr * 1.001, name
        bad = (err > 0.02 * n * lr) & solid
        print(f"[bench iteration] positions {name}: max |hip - host| {float(err.max()):.3e} (lr {lr:.1e}, moved up to "
              f"{float(moved.max()):.3e}); elements with a solid gradient off by > 2 % of the step: {int(bad.sum())} of {int(solid.sum())}")
        assert int(bad.sum()) <= max(2, int(solid.sum()) // 500), name
