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


from oracle.host_iteration import host_iteration as _host_iteration  # noqa: E402  (the host composition, shared with bench.py's cpu_baseline)


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

        from oracle.host_iteration import frame_state
        st, hc = frame_state(gm, cams, size)
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
