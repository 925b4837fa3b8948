#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native FluidNexus hot path.

Metric (BASELINE.json): train iters/sec + rasterise ms, 300k Gaussians x 5 views @ 512^2.
One "step" = one iteration of the per-frame optimisation loop over one batch of synthetic views
(fluidnexus_amd/harness.py); all inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--config {2,3,4,5}] [--scaling {auto,weak,strong}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (BASELINE.json `configs`):
  config 2  ScalarReal frame: 100k grey Gaussians (gm_fluid, render_fluid, 1-channel rasteriser), 5 views,
            render-only loss (L1 + D-SSIM + distance_loss), first-frame stage: the positions are the leaf.
  config 3  FluidNexus-Smoke frame: 200k fluid + 100k static background Gaussians (render_dynamics, 3 channels),
            24.8k hidden particles, image + distance + exyz + gas + next-gas losses, Adam.   [default at --gpus 1, 2]
  config 4  = config 3's scene, its 5 views sharded over 4 ranks (2/1/1/1).                  [default at --gpus 4]
  config 5  FluidNexus-Ball scene: 350k fluid + 150k background Gaussians + 28k hidden particles, 8 views on a
            ring, every view rendered by the 3-channel AND the 1-channel rasteriser.          [default at --gpus 8]
Multi-GPU (one process per GPU, Gaussians / particles replicated, one RCCL all-reduce(sum) of the leaf gradient per
iteration, then the reference's 1/batch scaling and a replicated Adam step):
  --scaling strong (default): the config's views are sharded round-robin over the ranks (view v -> rank v mod N);
            `value` = iterations of the whole batch per second.
  --scaling weak: every rank renders the config's full view count of an N-times larger batch (per-GPU work fixed);
            `value` counts nominal-batch iterations: N * steps / seconds.

Prints ONE JSON line on rank 0 (see the keys at the bottom).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: RCCL between processes needs this before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# Caching-allocator settings of this process, unless the caller has set some (read at the first device allocation): the
# sequence leg re-captures its loop frame after frame while every per-splat array grows by 0.3 % per frame.  Rounding
# requests up to eighths of a power of two and never splitting blocks above 32 MB lets a frame reuse the previous frame's
# blocks instead of taking new segments from the driver (a hipMalloc costs 0.1 ms on one box and several ms on another:
# profiles/r06_sequence_setup.md); it costs a few per cent of reserved memory.
_ALLOC_VARS = ("PYTORCH_HIP_ALLOC_CONF", "PYTORCH_CUDA_ALLOC_CONF", "PYTORCH_ALLOC_CONF")
if not any(os.environ.get(_k) for _k in _ALLOC_VARS):
    for _k in _ALLOC_VARS[:2]:
        os.environ[_k] = "roundup_power2_divisions:8,max_split_size_mb:32"
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

SIZE = 512
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_SPEC_WAVE_INSTR = 256 * 4 * 2.4e9 / 2 * 1.0  # 1024 SIMD-32s x one wave64 VALU instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)
VALU_MEASURED_WAVE_INSTR = 933e9                   # tools/micro/pk_rate.hip on this chip
PROFILE_TAG = "r06"

CONFIGS = {
    2: dict(workload="BASELINE configs[1]: ScalarReal single frame, 100k grey Gaussians (gm_fluid / render_fluid / ch1), "
                     "5 views @512^2, render-only loss (L1 + D-SSIM + distance_loss), first-frame stage (positions = leaf), Adam",
            views=5, channels=1, stage="first"),
    3: dict(workload="BASELINE configs[2]: FluidNexus-Smoke frame, 200k fluid + 100k background Gaussians, "
                     "24800 hidden particles, ch3, L1+D-SSIM + distance + exyz + gas + next-gas losses, Adam",
            views=5, channels=3, stage="physical"),
    4: dict(workload="BASELINE configs[3]: FluidNexus-Smoke frame (config 3's scene), 5 views sharded over the ranks "
                     "(2/1/1/1 on 4 GPUs), RCCL all-reduce of the leaf gradient",
            views=5, channels=3, stage="physical"),
    5: dict(workload="BASELINE configs[4]: FluidNexus-Ball scene, 350k fluid + 150k background Gaussians, 28072 hidden "
                     "particles, 8 views on a ring, ch3 (fluid + background) AND ch1 (fluid) rasterisers per view, "
                     "L1+D-SSIM on both + distance + exyz + gas + next-gas losses, Adam",
            views=8, channels=3, stage="physical"),
}


def build_workload(cfg_id, n_views, dev, rank, world, a, use_dist):
    """(gm, cams, loop) of the configuration with `n_views` cameras in the batch."""
    from fluidnexus_amd import harness as Hn
    graph = not (a.no_graph or a.host_sync)
    if cfg_id == 2:
        gm, cams = Hn.build_scalar_real_frame(100_000, n_views=n_views, size=SIZE, seed=0, device=dev)
        cfg = dict(Hn.SCALAR_REAL)
        if a.no_distance:
            cfg["lambda_first_distance"] = 0.0
        loop = Hn.FirstFrameLoop(gm, cams, rd_pipe="render_fluid", rank=rank, world=world, cfg=cfg, capturable=graph,
                                 force_all_reduce=use_dist)
        return gm, cams, loop
    if cfg_id in (3, 4):
        gm, cams = Hn.build_smoke_frame(200_000, 100_000, (20, 62, 20), n_views=n_views, size=SIZE, seed=0, device=dev,
                                        occluding=a.scene == "r01")
    else:
        gm, cams = Hn.build_ball_frame(350_000, 150_000, (22, 58, 22), n_views=n_views, size=SIZE, seed=0, device=dev)
    if a.stage == "visual":
        # the stage after the physical-particle one (train_visual_particle.py): positions fixed, colour / opacity /
        # scales / rotation of the fluid Gaussians are the leaves; the rasteriser's full backward
        cfg = dict(Hn.SMOKE_L2)
        loop = Hn.HotLoopLevelTwo(gm, cams, rank=rank, world=world, cfg=cfg, batched_views=True, capturable=graph,
                                  force_all_reduce=use_dist)
        loop.view_mode = "batched"
        return gm, cams, loop
    if a.stage == "first":
        # the first frame of a dynamics scene (entries_fluid_nexus/train_physical_particle.py:103-163): the visual
        # particles' positions are the leaf, world units, grey-mean image term + distance loss, no physics
        gm._visual_xyz = gm._visual_xyz / gm.scale_factor
        cfg = dict(Hn.SCALAR_REAL)
        if a.no_distance:
            cfg["lambda_first_distance"] = 0.0
        loop = Hn.FirstFrameLoop(gm, cams, rd_pipe="render_dynamics", rank=rank, world=world, cfg=cfg, capturable=graph,
                                 force_all_reduce=use_dist)
        loop.view_mode = "batched"
        return gm, cams, loop
    cfg = dict(Hn.SMOKE)
    if a.no_distance:
        cfg["lambda_current_distance"] = 0.0
    if getattr(a, "freeze", False):
        # learning rate 0: every iteration does the same work on the same particles.  For --emulate-world with a placement of
        # the view-independent terms that leaves a rank without some of them (last-rank, spread): without the all-reduce that
        # rank would follow another objective, its particles drift (the K-cap watch fires) and the share it times is no
        # longer the configuration's.
        cfg["position_lr_init"] = cfg["position_lr_final"] = 0.0
    view_mode = a.views
    if a.unfused_physics or a.image_loss == "torch":
        view_mode = "serial"  # the batched / branch modes build on the fused loss nodes
    if view_mode == "branches" and not graph:
        view_mode = "serial"
    if cfg_id == 5 and view_mode != "batched":
        raise SystemExit("config 5 (both rasterisers per view) needs --views batched")
    # view-independent terms (physics, distance): on every rank once per local view (default), or on ONE rank `batch`
    # times: the last rank of a multi-rank run (fewest views under round-robin sharding)
    shard_n = a.emulate_world if a.emulate_world > 1 else world
    shared_rank = None
    # (an emulated share keeps the per-view form unless asked: without the all-reduce a rank that drops the terms would
    # optimise a different objective, its particles drift and the timed workload is no longer the configuration's)
    # auto: last-rank as soon as there is more than one rank (the sum after the all-reduce is the same, gloo-tested in
    # tests/test_distributed_cpu.py, and the work lands on the rank with the fewest views); a single rank has nobody to share with
    if a.shared_terms == "spread":
        shared_rank = "spread"  # round 6: one term per rank (harness.spread_owners), summed by the same all-reduce
    elif a.shared_terms == "last-rank" or (a.shared_terms == "auto" and shard_n > 1 and view_mode == "batched"
                                           and (not (a.emulate_world > 1) or getattr(a, "freeze", False))):
        # (measured on every rank's emulated share with a frozen scene, profiles/r06_emulated_shares.md: the slowest rank of
        #  config 4 on 4 ranks runs 1631 it/s per-view, 1664 last-rank, 1659 spread; of config 5 on 8 ranks 1214 / 1244 /
        #  1211 -- the view-independent terms ride on side streams, where they go moves the bound by 2 %)
        shared_rank = shard_n - 1
    loop = Hn.HotLoop(gm, cams, rank=rank, world=world, force_all_reduce=use_dist, physics_per_view=not a.physics_once,
                      shared_terms_rank=shared_rank,
                      image_loss="torch" if a.image_loss == "torch" else "fused", fused_physics=not a.unfused_physics,
                      defer_visual_backward=not a.unfused_physics, capturable=graph, cfg=cfg,
                      parallel_views=view_mode == "branches", batched_views=view_mode == "batched",
                      fused_step=view_mode == "batched" and graph and not a.torch_adam, dual_channel=cfg_id == 5)
    loop.view_mode = view_mode
    return gm, cams, loop


def scene_arrays(gm, cfg_id, stage="physical"):
    """(xyz, opacity, scales, rotations, colours) of everything the configuration's main render sees, on the host."""
    with torch.no_grad():
        if cfg_id != 2 and stage in ("visual", "first"):
            fluid = gm._visual_xyz.detach() / gm.scale_factor if stage == "visual" else gm._visual_xyz.detach()
            xyz = torch.cat([fluid, gm.get_gs_xyz], 0).cpu().numpy()
            opac = torch.cat([gm.get_visual_opacity, gm.get_gs_opacity], 0).cpu().numpy()
            scales = torch.cat([gm.get_visual_scaling, gm.get_gs_scaling], 0).cpu().numpy()
            rots = torch.cat([gm.get_visual_rotation, gm.get_gs_rotation], 0).cpu().numpy()
            cols = torch.cat([gm.get_visual_color.repeat(1, 3), gm.get_gs_color], 0).cpu().numpy()
            return xyz, opac, scales, rots, cols
        if cfg_id == 2:
            return (gm.get_visual_xyz.detach().cpu().numpy(), gm.get_visual_opacity.cpu().numpy(),
                    gm.get_visual_scaling.cpu().numpy(), gm.get_visual_rotation.cpu().numpy(), gm.get_visual_color.cpu().numpy())
        xyz = torch.cat([gm.get_visual_xyz_from_nn() / gm.scale_factor, gm.get_gs_xyz], 0).cpu().numpy()
        opac = torch.cat([gm.get_visual_opacity, gm.get_gs_opacity], 0).cpu().numpy()
        scales = torch.cat([gm.get_visual_scaling, gm.get_gs_scaling], 0).cpu().numpy()
        rots = torch.cat([gm.get_visual_rotation, gm.get_gs_rotation], 0).cpu().numpy()
        cols = torch.cat([gm.get_visual_color.repeat(1, 3), gm.get_gs_color], 0).cpu().numpy()
    return xyz, opac, scales, rots, cols


def cpu_baseline(gm, cams, bg, cfg_id, views, channels, stage="physical"):
    """The CPU oracle (oracle/raster_oracle.c, OpenMP over all host cores) timed on ONE view of the same workload:
    rasteriser forward + backward.  Reported as whole-batch iterations per second of the rasteriser alone (losses /
    physics / Adam are not in the CPU sample)."""
    from oracle import raster_oracle as O
    O.build()
    cores = os.cpu_count() or 1
    O.set_threads(cores)
    cam = cams[0]
    xyz, opac, scales, rots, cols = scene_arrays(gm, cfg_id, stage)
    tan = math.tan(cam.FoVx * 0.5)
    t0 = time.perf_counter()
    f = O.forward(xyz, opac, bg.cpu().numpy(), cam.world_view_transform.cpu().numpy(),
                  cam.full_proj_transform.cpu().numpy(), cam.camera_center.cpu().numpy(), SIZE, SIZE, tan, tan,
                  colors_precomp=cols, scales=scales, rotations=rots, channels=channels)
    t1 = time.perf_counter()
    O.backward(f, np.ones((channels, SIZE, SIZE), np.float32))
    t2 = time.perf_counter()
    per_view = t2 - t0
    return {"value": 1.0 / (views * per_view), "unit": "iters/s", "cores": cores, "kind": "port",
            "scope": "rasteriser only (losses, physics terms and the optimiser step are not in this sample)",
            "sample": f"oracle/raster_oracle.c (OpenMP, {cores} threads), rasteriser only (ch{channels}): 1 of {views} views "
                      f"fwd {t1 - t0:.2f}s + bwd {t2 - t1:.2f}s, R={f['num_rendered']}; value = 1/({views} x that)"}


def cpu_baseline_whole_iteration(gm, cams, cfg, views):
    """BASELINE.md section 2's protocol on the HEADLINE configuration: the WHOLE iteration of the physical-particle loop on the
    host cores -- all `views` views through oracle/raster_oracle.c (OpenMP) forward + backward, the grey-mean L1 + D-SSIM loss
    and its image gradient with the torch-CPU utils.loss_utils, the hidden -> visual interpolation, the distance loss and the
    exyz / gas / next-gas terms with oracle/physics_oracle.py on k-d tree neighbour lists (oracle/host_iteration.py; same
    edge rule as the brute-force oracle, tests/test_host_iteration.py), torch.optim.Adam -- 1 warm-up, then the median of
    5 iterations (of 2 when the warm-up took more than 8 s: the sample is bounded to about half a minute of CPU work)."""
    import statistics
    from oracle import raster_oracle as O
    from oracle.host_iteration import KdPhysicsOracle, distance_loss_kdtree, frame_state, host_iteration
    O.build()
    # two OpenMP pools (the oracle's and torch's) that alternate: beyond a few dozen threads each they only fight each other
    # (their idle threads spin) -- the first whole-iteration sample ran 90 s per iteration with 256 + 256 threads, as the
    # config-1 sample had found before (12.8 s against 27 ms); both pools are held to 32 threads and the record says so
    cores = min(os.cpu_count() or 1, 32)
    O.set_threads(cores)
    torch_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    st, hc = frame_state(gm, cams[:views], SIZE)
    po = KdPhysicsOracle(H=cfg["H"], p0=cfg["p0"], secs=cfg["secs"], scale_factor=gm.scale_factor,
                         buoyancy_max_y=gm.buoyancy_max_y)
    x = torch.nn.Parameter(gm._estimate_xyz_nn.detach().double().cpu().clone())
    opt = torch.optim.Adam([x], lr=float(gm.optimizer.param_groups[0]["lr"]), eps=1e-15)
    times = []
    n_timed = 5
    it = 0
    while it < 1 + n_timed:
        t0 = time.perf_counter()
        x.grad = host_iteration(O, po, st, x, hc, cfg, views, distance=distance_loss_kdtree, image_dtype=torch.float32)
        opt.step()
        dt = time.perf_counter() - t0
        if it == 0 and dt > 8.0:
            n_timed = 2
        if it:
            times.append(dt)
        it += 1
    torch.set_num_threads(torch_threads)
    med = statistics.median(times)
    return {"value": 1.0 / med, "unit": "iters/s", "cores": cores, "cores_available": os.cpu_count(), "kind": "port", "scope": "whole iteration",
            "statistic": f"median of {len(times)} iterations after 1 warm-up (BASELINE.md section 2)",
            "seconds_per_iteration": {"median": round(med, 3), "min": round(min(times), 3), "max": round(max(times), 3)},
            "sample": f"{views} views x (oracle/raster_oracle.c forward + backward, OpenMP {cores} threads; torch-CPU grey-mean "
                      f"L1 + D-SSIM and its gradient) + hidden->visual interpolation, distance loss, exyz / gas / next-gas terms "
                      f"(oracle/physics_oracle.py on scipy k-d tree neighbour lists, float64) + torch.optim.Adam; "
                      f"{st['visual_xyz'].shape[0]} fluid + {st['gs_xyz'].shape[0]} background Gaussians, "
                      f"{x.shape[0]} hidden particles, {SIZE} x {SIZE}"}


def cpu_baseline_config1(dev):
    """BASELINE.md section 2, config 1 verbatim: 10k random Gaussians, one 256 x 256 view, forward + L1 / D-SSIM loss +
    backward + Adam on the host cores (oracle/raster_oracle.c forward / backward, OpenMP over all cores; the loss and its
    image gradient with the torch-CPU restatement of loss_utils.py; torch.optim.Adam, eps 1e-15, on all five attribute
    groups), median of 5 iterations after 1 warm-up -- next to the SAME iteration through the HIP rasteriser on the GPU."""
    import statistics
    from oracle import raster_oracle as O
    from fluidnexus_amd import rasterizer, synthetic as S
    from fluidnexus_amd.utils.loss_utils import l1_loss, ssim
    O.build()
    # 256 tiles and a 256 x 256 image: beyond a few dozen threads the two OpenMP pools (the oracle's and torch's) only
    # fight each other (measured on the 256-core GPU box: 12.8 s per iteration with 256 + 256 threads, the loss alone
    # taking > 12 s); the sample uses at most 32 and says so
    cores = min(os.cpu_count() or 1, 32)
    O.set_threads(cores)
    torch_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    W = H = 256
    tan = math.tan(0.4)
    g = S.random_gaussians(10_000, seed=0, box=0.5, log_scale=(-5.5, -3.5))
    cam = S.front_camera(W, H, device="cpu")
    bg = np.zeros(3, np.float32)
    target = torch.from_numpy(np.random.RandomState(0).uniform(size=(3, H, W)).astype(np.float32))
    names = ("means3D", "opacities", "colors", "scales", "rotations")
    lrs = dict(means3D=1.6e-4, opacities=0.05, colors=0.0025, scales=0.005, rotations=0.001)

    def image_loss(img, tgt):
        return 0.8 * l1_loss(img, tgt) + 0.2 * (1.0 - ssim(img, tgt))

    # host
    P = {k: torch.from_numpy(g[k].copy()).requires_grad_(True) for k in names}
    opt = torch.optim.Adam([{"params": [P[k]], "lr": lrs[k]} for k in names], lr=0.0, eps=1e-15)
    view, proj, campos = (cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), cam.camera_center.numpy())
    t_iter, t_fwd, t_bwd = [], [], []
    for it in range(6):
        t0 = time.perf_counter()
        a = {k: P[k].detach().numpy() for k in names}
        f = O.forward(a["means3D"], a["opacities"], bg, view, proj, campos, W, H, tan, tan, colors_precomp=a["colors"],
                      scales=a["scales"], rotations=a["rotations"], channels=3)
        t1 = time.perf_counter()
        img = torch.from_numpy(f["color"]).requires_grad_(True)
        loss = image_loss(img, target)
        dimg, = torch.autograd.grad(loss, img)
        t2 = time.perf_counter()
        go = O.backward(f, dimg.numpy())
        t3 = time.perf_counter()
        for k, gk in (("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("colors", "dL_dcolors"),
                      ("scales", "dL_dscales"), ("rotations", "dL_drotations")):
            P[k].grad = torch.from_numpy(np.ascontiguousarray(go[gk]).reshape(P[k].shape))
        opt.step()
        t4 = time.perf_counter()
        if it:
            t_iter.append(t4 - t0)
            t_fwd.append(t1 - t0)
            t_bwd.append(t3 - t2)
    # the same iteration on the device (eager launches, exact blend arithmetic: the oracle's own)
    from diff_gaussian_rasterization_ch3 import GaussianRasterizationSettings, GaussianRasterizer
    mode = rasterizer.get_blend_math()
    rasterizer.set_blend_math("exact")
    try:
        D = {k: torch.from_numpy(g[k].copy()).to(dev).requires_grad_(True) for k in names}
        optd = torch.optim.Adam([{"params": [D[k]], "lr": lrs[k]} for k in names], lr=0.0, eps=1e-15)
        camd = S.front_camera(W, H, device=dev)
        rs = GaussianRasterizationSettings(H, W, tan, tan, torch.zeros(3, device=dev), 1.0, camd.world_view_transform,
                                           camd.full_proj_transform, 0, camd.camera_center, False)
        tgt = target.to(dev)
        t_dev = []
        for it in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m2d = torch.zeros(10_000, 3, device=dev, requires_grad=True)
            img, _, _ = GaussianRasterizer(rs)(D["means3D"], m2d, D["opacities"], colors_precomp=D["colors"],
                                               scales=D["scales"], rotations=D["rotations"])
            image_loss(img, tgt).backward()
            optd.step()
            optd.zero_grad()
            torch.cuda.synchronize()
            if it:
                t_dev.append(time.perf_counter() - t0)
    finally:
        rasterizer.set_blend_math(mode)
        torch.set_num_threads(torch_threads)
        O.set_threads(os.cpu_count() or 1)
    med = statistics.median
    return {"workload": "BASELINE configs[0] verbatim: 10k random Gaussians, one 256x256 view, forward + L1 / D-SSIM loss + "
                        "backward + Adam (all five attribute groups)",
            "host": {"iters_per_s": 1.0 / med(t_iter), "ms_per_iter": med(t_iter) * 1e3, "forward_ms": med(t_fwd) * 1e3,
                     "backward_ms": med(t_bwd) * 1e3, "cores": cores, "cores_available": os.cpu_count(),
                     "code": "oracle/raster_oracle.c (OpenMP) + torch-CPU losses + torch.optim.Adam", "statistic": "median of 5 after 1 warm-up"},
            "mi355x": {"iters_per_s": 1.0 / med(t_dev), "ms_per_iter": med(t_dev) * 1e3,
                       "code": "HIP rasteriser (exact arithmetic) + torch losses + torch.optim.Adam, eager launches, host-synchronised per iteration"},
            "num_rendered": int(f["num_rendered"])}


def rasterise_timing(gm, cams, views, cfg_id, channels, bg, stage="physical"):
    """Rasteriser alone on this rank's views, HIP-event timed on the current stream: forward, and forward + backward
    with ALL gradients (means, opacity, colour, scales, rotations: the visual-particle stage's MODE 0 backward)."""
    from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
    from fluidnexus_amd.rasterizer import GaussianRasterizerViews
    from fluidnexus_amd.renderer.pipes import _settings
    xyz, opac, scales, rots, cols = scene_arrays(gm, cfg_id, stage)
    dev = bg.device
    _, GRsetting, _ = get_render_pipe("render_fluid" if channels == 1 else "render_dynamics")
    # (a stand-alone render: the depth sort from scratch, not the loop's repair of the previous iteration's order)
    rv = GaussianRasterizerViews([_settings(GRsetting, cams[v], bg, 1.0, 0) for v in views], channels=channels,
                                 options=dict(coherent_sort=0))
    L = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in
         dict(means3D=xyz, opacities=opac, scales=scales, rotations=rots, colors=cols).items()}
    P, V = xyz.shape[0], len(views)
    dL = torch.ones(V, channels, SIZE, SIZE, device=dev)

    def once(backward):
        screen = torch.zeros(V, P, 3, device=dev, requires_grad=True)
        im, _, _ = rv(means3D=L["means3D"], means2D=screen, opacities=L["opacities"], colors_precomp=L["colors"],
                      scales=L["scales"], rotations=L["rotations"])
        if backward:
            torch.autograd.grad([im], list(L.values()), grad_outputs=[dL])

    out = {}
    for name, bw in (("forward", False), ("forward_backward_all_gradients", True)):
        once(bw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            once(bw)
        e1.record()
        e1.synchronize()
        out[name] = e0.elapsed_time(e1) / 5 / V
    return out


def sh_timing(gm, cams, views, cfg_id, bg, degree, stage="physical"):
    """The SH pipe's rasteriser on the configuration's Gaussians (row d3 of the scope table): colours as [P, 16, 3]
    spherical-harmonics coefficients evaluated per view in the preprocess kernel (ch3 forward.cu:20-67) and
    differentiated in the per-splat backward (backward.cu:20-132).  Returns per-view milliseconds of the forward and of
    forward + backward with all gradients (SH coefficients included), the preprocess kernel's duration (HIP events in
    the library) with its algorithmic bytes, and the MFMA statement north_star asks for."""
    from fluidnexus_amd import _lib
    from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
    from fluidnexus_amd.rasterizer import GaussianRasterizerViews
    from fluidnexus_amd.renderer.pipes import _settings
    xyz, opac, scales, rots, _ = scene_arrays(gm, cfg_id, stage)
    dev = bg.device
    _, GRsetting, _ = get_render_pipe("render_gs")
    rv = GaussianRasterizerViews([_settings(GRsetting, cams[v], bg, 1.0, degree) for v in views], channels=3, options=dict(coherent_sort=0))
    P, V = xyz.shape[0], len(views)
    rng = np.random.RandomState(5)
    shs = np.zeros((P, 16, 3), np.float32)
    shs[:, 0] = rng.uniform(-1.0, 1.5, size=(P, 3))
    shs[:, 1:] = rng.normal(size=(P, 15, 3)) * 0.25
    L = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in
         dict(means3D=xyz, opacities=opac, scales=scales, rotations=rots, shs=shs).items()}
    dL = torch.ones(V, 3, SIZE, SIZE, device=dev)

    def once(backward):
        screen = torch.zeros(V, P, 3, device=dev, requires_grad=True)
        im, radii, _ = rv(means3D=L["means3D"], means2D=screen, opacities=L["opacities"], shs=L["shs"],
                          scales=L["scales"], rotations=L["rotations"])
        if backward:
            torch.autograd.grad([im], list(L.values()), grad_outputs=[dL])
        return radii

    out = {}
    for name, bw in (("sh_forward", False), ("sh_forward_backward", True)):
        once(bw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            once(bw)
        e1.record()
        e1.synchronize()
        out[name] = e0.elapsed_time(e1) / 5 / V
    def pre_stage_us():
        _lib.profile_enable(True)
        for _ in range(5):
            r = once(False)
        torch.cuda.synchronize()
        ms, n = _lib.profile_read(3)
        _lib.profile_enable(False)
        return r, ms, n

    radii, pre_ms, pre_n = pre_stage_us()
    # The SH colours of a view batch: on the matrix cores in the fast arithmetic (csrc/sh_mfma.h, v_mfma_f32_4x4x1_16B_f32:
    # one Gaussian per 4 x 4 block, 16 per wave instruction), the scalar kernel in the exact one.  Both timed here, in the
    # same process, by pinning the choice through the environment.
    variants = {}
    try:
        screen = torch.zeros(V, P, 3, device=dev)
        ims = {}
        for name, env in (("scalar", "0"), ("mfma", "1")):
            os.environ["FNX_LAB_SH_MFMA"] = env
            with torch.no_grad():
                ims[name] = rv(means3D=L["means3D"], means2D=screen, opacities=L["opacities"], shs=L["shs"], scales=L["scales"],
                               rotations=L["rotations"])[0].clone()
            _, m_ms, m_n = pre_stage_us()
            variants[name] = m_ms / max(m_n, 1) * 1e3
        n_mfma = ((P + 15) // 16) * ((V + 3) // 4) * 16  # wave-level v_mfma_f32_4x4x1_16B_f32 instructions per launch
        variants = {"per_splat_stage_us": variants, "max_abs_image_difference": float((ims["mfma"] - ims["scalar"]).abs().max()),
                    "mfma_wave_instructions_per_launch": n_mfma,
                    # 8 cycles per instruction (2 passes) on one of 1024 matrix pipes at 2.4 GHz, over the stage's duration
                    "mfma_pipe_busy_frac_computed": n_mfma * 8 / 1024 / 2.4e9 / (variants["mfma"] * 1e-6),
                    "what": "per-splat stage = SH colour kernel + preprocess kernel; `mfma`: sh_colors_views_mfma_kernel (block = one "
                            "Gaussian, rows = four views, columns = RGB, one v_mfma_f32_4x4x1_16B_f32 per SH coefficient), the "
                            "fast arithmetic's kernel; `scalar`: sh_colors_views_kernel, the exact arithmetic's"}
    except Exception as e:
        print(f"[bench] SH variants failed: {type(e).__name__}: {e}", file=sys.stderr)
        variants = None
    finally:
        os.environ.pop("FNX_LAB_SH_MFMA", None)
    p_vis = float((radii > 0).sum().item()) / V
    # per splat and view: reads xyz 12 + scale 12 + rotation 16 + opacity 4 + SH 16 x 12 = 236 B; a visible splat writes
    # the per-splat state of SURVEY 8(d) (60 B) + sort key 4 + rect 8 + blend record 64 + rgb 12 + clamped 3 = 151 B
    bytes_launch = V * (P * 236 + p_vis * 151)
    # round 4: a view batch evaluates the colours once per GAUSSIAN for all views (sh_colors_views_kernel: 192 + 12 B in,
    # 15 B out per view), the preprocess then reads 44 + 12 B per (splat, view): what the two kernels have to move
    moved = P * (204 + 15 * V) + V * (P * 56 + p_vis * 151)
    us = pre_ms / max(pre_n, 1) * 1e3
    return {"degree": degree, "gaussians": P, "views_per_launch": V, "rasterise_ms_per_view": out,
            "preprocess": {"avg_launch_us": us, "algorithmic_bytes_per_launch": int(bytes_launch),
                           "GBps": bytes_launch / (us * 1e-6) / 1e9 if us > 0 else None,
                           "frac_of_hbm_peak": bytes_launch / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if us > 0 else None,
                           "kernels": "sh_colors_views_kernel (coefficients read once per Gaussian for all views) + preprocess_kernel",
                           "moved_bytes_per_launch": int(moved),
                           "moved_frac_of_hbm_peak": moved / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if us > 0 else None,
                           "bound": "hbm"},
            "mfma_util": (variants or {}).get("mfma_pipe_busy_frac_computed", 0) if rasterizer_fast() else 0,
            "variants": variants,
            "why": "the SH contraction is, per Gaussian, a (V x 16) . (16 x 3) product whose operands both belong to that "
                   "Gaussian: 192 B of coefficients feed 96 multiply-adds per view (0.5 flop/B against a machine balance of ~20), so "
                   "the stage is bound by reading the coefficients whatever evaluates them.  The batched outer-product instruction "
                   "(16 Gaussians x 4 views x 4 channels per issue) still pays, through the LAYOUT it imposes: a wave's loads cover 16 "
                   "Gaussians' coefficient rows and a lane evaluates one (Gaussian, view) basis -- see `variants`; the matrix pipe "
                   "itself stays a few percent busy (DESIGN.md 4.8)"}


def rasterizer_fast():
    from fluidnexus_amd import rasterizer
    return rasterizer.get_blend_math() == "fast"


def drop_in_timing(a, dev, cfg_id, steps=6, auto=False):
    """What a FluidNexus user gets from `PYTHONPATH=<this repo>` alone: the reference's own op sequence
    through the plug-in seam -- one GaussianRasterizer autograd node per view (train_physical_particle.py:338-405), the
    image / physics / distance terms as separate autograd nodes, gm.cache_gradient_current per view, torch.optim.Adam,
    the per-forward host sync of rasterizer_impl.cu:264, bit-exact blend arithmetic, no static split, no graph, no
    view batching.  Everything faster than this needs fluidnexus_amd.harness (or the entry script edited to call the
    view-batched pipe)."""
    from types import SimpleNamespace
    from fluidnexus_amd import rasterizer
    from fluidnexus_amd.renderer import pipes
    import fluidnexus_amd
    keep = dict(rasterizer._OPTS)
    keep_sync, keep_split, keep_cudnn = rasterizer._HOST_SYNC, pipes._STATIC_SPLIT, torch.backends.cudnn.enabled
    keep_auto = fluidnexus_amd.auto_enabled()
    try:
        rasterizer.set_blend_math("exact")
        rasterizer.set_lean_geometry(False)
        rasterizer.set_sort_narrow(False)
        rasterizer.set_coherent_sort(False)
        rasterizer.set_host_sync(True)
        pipes.set_static_split(bool(auto))
        # `auto`: the SAME loop -- the reference's per-view op sequence, torch.optim.Adam, exact arithmetic -- with the
        # library-side automation of the seam switched on (FNX_AUTO=1 for a user): nothing in the calling code changes
        fluidnexus_amd.set_auto(bool(auto))
        torch.backends.cudnn.enabled = False  # utils.loss_utils.ssim's conv2d on ATen's own kernels (MIOpen's cold find step)
        b = SimpleNamespace(**vars(a))
        b.no_graph, b.host_sync, b.image_loss, b.unfused_physics, b.torch_adam, b.views = True, True, "torch", True, True, "serial"
        gm, cams, loop = build_workload(cfg_id, CONFIGS[cfg_id]["views"], dev, 0, 1, b, False)
        loop.make_targets()
        for _ in range(3):
            loop.iteration()
        torch.cuda.synchronize()
        # a host-bound loop on a box whose cores are shared: the MEDIAN of `steps` groups of 2 iterations (one outlier -- a
        # first-use allocation, a scheduler hiccup -- used to halve the figure of an 8-iteration mean)
        groups = []
        for _ in range(steps):
            t0 = time.perf_counter()
            loop.iteration()
            loop.iteration()
            torch.cuda.synchronize()
            groups.append((time.perf_counter() - t0) / 2)
        dt = sorted(groups)[len(groups) // 2]
        if auto:
            rasterizer.check_status()
            return {"drop_in_iters_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "steps": 2 * steps, "statistic": "median of 2-iteration groups",
                    "what": "the same per-view op sequence and the same calling code as `drop_in` with fluidnexus_amd's automation of "
                            "the seam on (FNX_AUTO=1): render_dynamics(camera, ...) routes through the view-batched rasteriser with one "
                            "view (background binned once per camera, no host sync per forward, positions-only backward, coherent "
                            "depth sort per camera), utils.loss_utils.ssim runs the fused kernel; exact blend arithmetic, torch.optim.Adam, "
                            "per-view physics / distance autograd nodes, per-view gradient cache"}
        return {"drop_in_iters_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "steps": 2 * steps, "statistic": "median of 2-iteration groups",
                "what": "reference op sequence through the plug-in seam: per-view rasteriser autograd nodes, torch image / "
                        "physics / distance terms, per-view gradient cache, torch.optim.Adam, host sync per forward, exact "
                        "blend arithmetic, no static split / view batching / graph (bench.py --views serial --no-graph "
                        "--image-loss torch --torch-adam --unfused-physics --host-sync --blend-math exact --no-static-split)"}
    finally:
        fluidnexus_amd.set_auto(keep_auto)  # (first: switching the automation off restores the host-sync mode IT found)
        rasterizer.set_blend_math(("exact", "fast")[keep["blend_math"]])
        rasterizer.set_lean_geometry(bool(keep["lean_geometry"]))
        rasterizer.set_sort_narrow(bool(keep["sort_narrow"]))
        rasterizer.set_coherent_sort(bool(keep["coherent_sort"]))
        rasterizer.set_host_sync(keep_sync)
        pipes.set_static_split(keep_split)
        torch.backends.cudnn.enabled = keep_cudnn


def sequence_timing(a, dev, cfg_id, rank, world, use_dist, steady_ms):
    """BASELINE config 3 as what it names: a multi-FRAME sequence.  Per frame the reference's boundary in its call order
    (entries_fluid_nexus/train_physical_particle.py:283-302: remove_invalid_particles -> emit_new_particles ->
    guess_hidden_particles -> update_solver_counts x solver_iterations -> project_gas_constraints x solver_iterations ->
    training_setup_current -> prepare_visual_particles_for_rendering), then n optimisation iterations (tpp:329-432),
    then the hand-over (tpp:456-458: confirm_guess_hidden_particles_from_nn -> update_visual_xyz_from_nn ->
    confirm_guess_hidden_particles_wo_velocity).  The emitter changes the particle counts every frame, so the static
    background is re-binned, the binning high-water mark and the sort state start over and the hipGraph is re-captured
    -- all inside the timed region.  Dataset I/O (the frame's target images) is not part of the path: the targets of the
    first frame are kept."""
    from fluidnexus_amd import harness as Hn, rasterizer
    K, n = int(a.frames), int(a.iters_per_frame)
    gm, cams, loop = build_workload(cfg_id, CONFIGS[cfg_id]["views"], dev, rank, world, a, use_dist)
    loop.make_targets()
    # a synthetic emitter at the foot of the plume (world units, like prepare_emitter_points leaves them): one lattice
    # layer of hidden particles and a disc of visual particles per frame
    rng = np.random.RandomState(11)
    cx, cz = 0.34, -0.225
    hx, hz = np.meshgrid(np.arange(-4, 5) * 0.01, np.arange(-4, 5) * 0.01, indexing="ij")
    keep = (hx ** 2 + hz ** 2) <= 0.045 ** 2
    hid = np.stack([cx + hx[keep], np.full(keep.sum(), -0.025), cz + hz[keep]], 1)
    ang, rad = rng.uniform(0, 2 * np.pi, 600), 0.09 * np.sqrt(rng.uniform(0, 1, 600))
    vis = np.stack([cx + rad * np.cos(ang), rng.uniform(-0.02, -0.015, 600), cz + rad * np.sin(ang)], 1)
    gm.hidden_emitter_points = torch.tensor(hid, dtype=torch.float32, device=dev)
    gm.visual_emitter_points = torch.tensor(vis, dtype=torch.float32, device=dev)
    gm.emit_ratio_hidden, gm.emit_ratio_visual, gm.extra_visual_ratio, gm.extra_visual_num = 1.0, 1.0, 0.0, 0
    N0 = gm._xyz.shape[0]
    gm._particle_id = torch.arange(N0, device=dev).unsqueeze(1)
    gm._particle_id_max = N0
    gm._counts = torch.zeros(N0, 1, dtype=torch.float32, device=dev)
    solver_iterations = 3  # configs/fluid_nexus_smoke_dynamics.json
    optim = loop.optim_args
    graph = loop.capturable
    gi = max(1, min(int(a.seq_graph_iters), n))
    seg = dict(simulate=0.0, setup=0.0, optimise=0.0, accept=0.0)
    counts, sort_switched, frame_s, knn_checked, setup_s, full_sorts_setup, full_sorts_frame = [], [], [], [], [], [], []
    setup_split = []  # [prepare, loop object, eager iterations, status, capture, first replay + counters] ms per frame
    full_sort_why = []
    # one memory pool for the captures of all frames (HotLoop.capture: private pools of destroyed graphs pile up otherwise)
    # (a torch.cuda.MemPool keeps the pool alive between one frame's graph going and the next one's capture: a bare handle's
    #  pool is dropped with its last graph and capture_begin then asserts)
    seq_pool_obj = torch.cuda.MemPool() if (graph and os.environ.get("FNX_SEQ_SHARED_POOL", "1") != "0") else None
    seq_pool = seq_pool_obj.id if seq_pool_obj is not None else None
    seg_split = []

    def tick():
        torch.cuda.synchronize()
        return time.perf_counter()

    def one_frame(timed):
        nonlocal loop
        t0 = tick()
        gm.remove_invalid_particles()
        gm.emit_new_particles()
        gm.guess_hidden_particles()
        for _ in range(solver_iterations):
            gm.update_solver_counts()
        for _ in range(solver_iterations):
            gm.project_gas_constraints()
        t1 = tick()
        # training_setup_current + the loop object of the frame (its warm-up iterations are optimisation iterations)
        gm.prepare_visual_particles_for_rendering()
        rasterizer.release_captured_status()
        from fluidnexus_amd import physics as _physics
        _physics.release_captured_distance_state()  # (the distance loss's pair-list state the previous frame's graph pinned)
        if os.environ.get("FNX_SEQ_RELEASE_GRAPH", "1") != "0":
            loop.release_graph()  # the previous frame's graph and its memory go now, not when the collector finds the loop
        if os.environ.get("FNX_SEQ_GC") == "1":  # developer switch
            import gc
            gc.collect()
        ts = [tick()]  # set-up split: five more synchronisations per frame (~0.1 ms of a 10 ms set-up)
        ms0 = torch.cuda.memory_stats(dev)
        loop = Hn.HotLoop(gm, cams, rank=rank, world=world, force_all_reduce=use_dist, physics_per_view=loop.physics_per_view,
                          shared_terms_rank=loop.shared_terms_rank, image_loss=loop.image_loss, fused_physics=loop.fused_physics,
                          defer_visual_backward=True, capturable=graph, cfg=loop.cfg, batched_views=loop.batched_views,
                          fused_step=loop.fused_step, dual_channel=loop.dual_channel,
                          reuse_streams_of=loop if os.environ.get("FNX_SEQ_REUSE_STREAMS", "1") != "0" else None)
        done = 0
        rasterizer.set_coherent_sort(a.coh)
        P_frame = int(gm._visual_xyz.shape[0])  # the per-call splats of this frame: only their sort states count below
        c0 = rasterizer.coherent_sort_counters(P_frame)
        # eager: sizes the binning buffers for the new particle count, seeds the sort state, and shows whether this
        # frame's particles stay inside the coherent sort's repair window (else: radix passes for this frame)
        if ts is not None:
            ts.append(tick())
        for _ in range(max(1, int(a.seq_eager))):
            loop.iteration()
            done += 1
        if ts is not None:
            ts.append(tick())
        rasterizer.check_status()
        if ts is not None:
            ts.append(tick())
        captured = False
        if graph and n - done - 1 >= gi:
            loop.capture(warmup=int(a.seq_capture_warmup), iterations=gi, pool=seq_pool)
            done += int(a.seq_capture_warmup)
            captured = True
        if ts is not None:
            ts.append(tick())
        if a.sort == "coherent":
            # does the frame stay inside the coherent sort's reach?  Judged on the first replay (round 5: two eager iterations
            # fewer per frame than judging before the capture); a frame that does not is re-captured on the radix passes
            if captured and n - done >= gi:
                loop.iteration()
                done += gi
            else:
                for _ in range(min(2, n - done)):
                    loop.iteration()
                    done += loop.iterations_per_call
            c1 = rasterizer.coherent_sort_counters(P_frame)
            full_sorts_setup.append(c1[1] - c0[1])
            if c1[1] - c0[1] > rasterizer.coherent_sort_states(P_frame):
                rasterizer.set_coherent_sort(False)
                sort_switched.append(len(counts))
                if captured and n - done - 1 >= gi:
                    loop.use_graph(False)
                    loop.capture(warmup=1, iterations=gi, pool=seq_pool)
                    done += 1
        t2 = tick()
        if ts is not None:
            setup_split.append([round((b_ - a_) * 1e3, 2) for a_, b_ in zip([t1] + ts, ts + [t2])])
            ms1 = torch.cuda.memory_stats(dev)
            # device-memory segments the caching allocator took from / gave back to the driver during the set-up, reserved GiB
            seg_split.append([int(ms1.get("segment.all.allocated", 0) - ms0.get("segment.all.allocated", 0)),
                              int(ms1.get("segment.all.freed", 0) - ms0.get("segment.all.freed", 0)),
                              round(ms1.get("reserved_bytes.all.current", 0) / 2**30, 2),
                              round(ms1.get("allocated_bytes.all.current", 0) / 2**30, 2)])
        while done < n:
            if loop.iterations_per_call > n - done:
                loop.use_graph(False)
            done += loop.iterations_per_call
            loop.iteration()
        rasterizer.check_status()
        if a.sort == "coherent":
            full_sorts_frame.append(rasterizer.coherent_sort_counters(P_frame)[1] - c0[1])
            why = 0  # sticky reasons of this frame's states (csrc/fnx_state.h COH_WHY: 1 record not of this call, 2 bucket
            for vb_ in list(rasterizer._VIEW_BATCHES or ()):  # overflow, 4 chunk not increasing, 8 chunk boundary, 16 unseeded)
                for key_ in list(vb_._sort_state):
                    if key_[1] == P_frame:
                        for row_ in vb_.sort_counters(key_[0], key_[1], why=True, sort_key=key_[2] if len(key_) > 2 else None):
                            why |= row_[2]
            full_sort_why.append(why)
        if getattr(gm, "_knn_flags", None) is not None:
            gm.check_knn_k()  # raises when a fused neighbour search of this frame met a list longer than KNN_K (device flag)
            knn_checked.append(1)
        t3 = tick()
        gm.confirm_guess_hidden_particles_from_nn()
        gm.update_visual_xyz_from_nn()
        gm.confirm_guess_hidden_particles_wo_velocity()
        t4 = tick()
        if timed:
            seg["simulate"] += t1 - t0
            seg["setup"] += t2 - t1
            seg["optimise"] += t3 - t2
            seg["accept"] += t4 - t3
            counts.append((int(gm._xyz.shape[0]), int(gm._visual_xyz.shape[0])))
            frame_s.append(t4 - t0)
            setup_s.append(t2 - t1)
        return t4 - t0

    one_frame(False)  # untimed: first-use allocations of every stage
    if os.environ.get("FNX_SEQ_SNAPSHOT") == "2":
        torch.cuda.memory._record_memory_history(enabled="all", context="alloc", stacks="python")
    sort_switched.clear()
    knn_checked.clear()
    full_sorts_setup.clear()
    full_sorts_frame.clear()
    setup_split.clear()
    seg_split.clear()
    full_sort_why.clear()
    total = sum(one_frame(True) for _ in range(K))
    rasterizer.set_coherent_sort(a.coh)
    if os.environ.get("FNX_SEQ_SNAPSHOT") == "2":  # developer switch: who allocated the blocks that are still alive
        import collections
        live = collections.Counter()
        for sg in torch.cuda.memory_snapshot():
            for b in sg["blocks"]:
                if b["state"] == "active_allocated":
                    fr = [f for f in b.get("frames", []) if "/repo/" in f.get("filename", "")][:3]
                    live[" < ".join(f"{os.path.basename(f['filename'])}:{f['line']}" for f in fr)] += b["size"]
        for k, v in live.most_common(14):
            print(f"[live] {v / 2**20:9.1f} MiB  {k}", file=sys.stderr, flush=True)
    if os.environ.get("FNX_SEQ_SNAPSHOT") == "1":  # developer switch: where the reserved device memory sits after the sequence
        import collections
        by_pool = collections.defaultdict(lambda: [0, 0, 0, 0])
        for sg in torch.cuda.memory_snapshot():
            e = by_pool[tuple(sg.get("segment_pool_id", (0, 0)))]
            e[0] += 1
            e[1] += sg["total_size"]
            e[2] += sg["allocated_size"]
            e[3] += sum(1 for b in sg["blocks"] if b["state"] == "active_allocated")
        for k, e in sorted(by_pool.items(), key=lambda kv: -kv[1][1])[:12]:
            print(f"[snapshot] pool {k}: {e[0]} segments, {e[1] / 2**30:.2f} GiB reserved, {e[2] / 2**30:.2f} GiB allocated, {e[3]} live blocks",
                  file=sys.stderr, flush=True)
    per_frame = total / K
    return {"frames": K, "iters_per_frame": n, "seq_iters_per_s": K * n / total, "ms_per_frame": per_frame * 1e3,
            "frame_boundary_ms": (per_frame - n * steady_ms * 1e-3) * 1e3,
            "vs_steady_state": (K * n / total) / (1e3 / steady_ms),
            "segments_ms_per_frame": {k: v / K * 1e3 for k, v in seg.items()},
            "particles_last_frame": {"hidden": counts[-1][0], "visual": counts[-1][1]},
            "emitted_per_frame": {"hidden": int(hid.shape[0]), "visual": int(vis.shape[0])},
            "frames_on_radix_sort": len(sort_switched),
            "per_frame_iters_per_s": {"min": n / max(frame_s), "median": n / sorted(frame_s)[len(frame_s) // 2], "max": n / min(frame_s),
                                      "first": n / frame_s[0], "last": n / frame_s[-1]},
            "particles_first_frame": {"hidden": counts[0][0], "visual": counts[0][1]},
            "setup_ms_by_frame": [round(x * 1e3, 1) for x in setup_s],
            "frames_switched_to_radix": list(sort_switched),
            # in-launch full sorts of the coherent depth sort (all views): during a frame's set-up (eager iterations + first
            # replay), and over the whole frame (a frame switched to the radix passes stops counting)
            "full_sorts_in_setup_by_frame": list(full_sorts_setup), "full_sorts_by_frame": list(full_sorts_frame),
            "full_sort_reasons_by_frame": list(full_sort_why),
            **({"setup_split_ms_by_frame": setup_split, "allocator_segments_by_frame": seg_split} if setup_split else {}),
            "knn_watch": (f"checked at every frame boundary ({len(knn_checked)} frames): no fused neighbour search met a list longer "
                          f"than KNN_K = {int(gm.KNN_K)}") if knn_checked else None,
            "note": "per frame: remove -> emit -> predict -> solver counts x3 -> project x3 | new Adam + loop, two eager "
                    "iterations (the first seeds the depth sort's state), static background re-binned, hipGraph re-captured, its first "
                    "replay shows whether the frame stays inside the coherent sort's reach (setup: its iterations count towards n) | "
                    "replayed iterations | confirm + advect + confirm; frame_boundary_ms = ms_per_frame - n x the steady-state "
                    "ms_per_step of this record; a host sync at each segment boundary (4 per frame)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5,
                    help="how many times the timed region (exactly --steps steps) is run: value = the median pass, `spread` = all")
    ap.add_argument("--config", default="auto", choices=["auto", "2", "3", "4", "5"],
                    help="BASELINE.json configuration (1-based); auto: 3 at 1-2 GPUs, 4 at 4 GPUs, 5 at 8 GPUs")
    ap.add_argument("--stage", default="physical", choices=["physical", "visual", "first"],
                    help="configs 3-5: physical = the per-frame hidden-particle loop BASELINE's metric is quoted on; visual = "
                         "the visual-particle stage that follows it (attributes are the leaves, full rasteriser backward); "
                         "first = the first-frame stage of a dynamics scene (visual positions are the leaf, no physics)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"],
                    help="strong (auto): the config's views sharded over the ranks; weak: the config's view count per rank")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="single process: run rank 0's share of the views of an N-rank strong-scaling run, no communication")
    ap.add_argument("--scene", default="backdrop", choices=["backdrop", "r01"],
                    help="configs 3 / 4: background behind the plume (default), or the round-1 layout (a cloud AROUND "
                         "the plume that hides it from every camera: zero image gradient; for like-for-like comparisons)")
    ap.add_argument("--blend-math", default="fast", choices=["fast", "exact"],
                    help="arithmetic of the blend kernels (include/fnx_raster.h fnx_set_blend_math): fast = fused multiply-adds "
                         "+ v_exp_f32, stated tolerance against the oracle (tests/test_fast_math_gpu.py); exact = the "
                         "bit-reproducible sequence the oracle repeats")
    ap.add_argument("--sort-four-passes", action="store_true",
                    help="always launch the fourth depth-sort pass (default: skipped once the warm-up has shown spans < 2^26 ulps)")
    ap.add_argument("--sort", default="coherent", choices=["coherent", "coherent-always", "radix"],
                    help="depth sort of the per-call splats: coherent = one launch that repairs the previous iteration's order "
                         "(verified on the device, in-launch full sort when the check fails: exact by construction, "
                         "tests/test_coherent_sort_gpu.py); radix = the 9-bit LSD passes every call")
    ap.add_argument("--full-geometry", action="store_true",
                    help="write every per-view copy of the reference's GeometryState (default: fnx_set_lean_geometry(1))")
    ap.add_argument("--emulate-rank", type=int, default=0, help="with --emulate-world: which rank's share to time")
    ap.add_argument("--freeze", action="store_true",
                    help="learning rate 0: every iteration repeats the first one's work (for --emulate-world with --shared-terms "
                         "last-rank / spread, where a rank without the all-reduce would drift off the configuration)")
    ap.add_argument("--shared-terms", default="auto", choices=["auto", "per-view", "last-rank", "spread"],
                    help="who evaluates the view-independent terms (physics, distance loss) in a multi-rank run: every rank, "
                         "once per local view (per-view, default: the reference's evaluation count, rank by rank), or only the "
                         "last rank -- the one with the fewest views -- `batch` times (last-rank: same sum after the "
                         "all-reduce, tests/test_distributed_cpu.py), or one TERM per rank (spread, round 6: the gas term, the "
                         "next-gas + exyz terms and the distance loss each on the rank with the least work, harness.spread_owners; "
                         "same sum, same all-reduce); auto = last-rank whenever there is more than one rank (emulated shares: the "
                         "placement moves the slowest rank by 2 %).  None of them is measured on multi-GPU hardware")
    ap.add_argument("--deep-kernel", type=int, default=None, choices=[0, 1, 2, 3, 4, 5],
                    help="which blend forward kernel takes which tiles (fnx_raster_opts_t.deep_kernel): 0 per-tile kernel only, 1 / 2 lab "
                         "super-batch kernel (launches of <= 2 views / always), 3 / 4 staging waves (deep tiles / every tile), "
                         "5 staging waves by the launch's view count (library default)")
    ap.add_argument("--sh-degree", type=int, default=-1, choices=[-1, 0, 1, 2, 3],
                    help="also time the SH pipe's rasteriser (colours as spherical-harmonics coefficients of this degree) "
                         "on the configuration's Gaussians: record key `sh` (not part of the timed training step)")
    ap.add_argument("--no-drop-in", action="store_true",
                    help="skip the `drop_in` leg (the reference's op sequence through the plug-in seam, ~1 s)")
    ap.add_argument("--no-exact-leg", action="store_true",
                    help="skip the `exact_mode` leg (the same loop re-captured with the bit-exact blend arithmetic, ~1 s)")
    ap.add_argument("--frames", type=int, default=-1,
                    help="also time a sequence of this many frames (config 3 / 5, physical stage): the reference's frame "
                         "boundary in its call order + --iters-per-frame optimisation iterations per frame, record key "
                         "`sequence` (seq_iters_per_s, frame_boundary_ms); 0 = off; -1 (default) = 3 frames of 250 iterations "
                         "in a single-GPU config-3 run (the reference's level-two stage runs 250 per frame: ~1 s), else off")
    ap.add_argument("--iters-per-frame", type=int, default=0,
                    help="optimisation iterations per frame of --frames (configs/fluid_nexus_smoke_dynamics.json: 1000; "
                         "default: 1000 with an explicit --frames, 250 in the default leg)")
    ap.add_argument("--seq-eager", type=int, default=1,
                    help="sequence leg: eager iterations of a frame in front of its capture (the first sizes the binning buffers and "
                         "seeds the depth sort's state)")
    ap.add_argument("--seq-capture-warmup", type=int, default=0, help="sequence leg: eager iterations inside HotLoop.capture")
    ap.add_argument("--seq-graph-iters", type=int, default=3,
                    help="iterations per hipGraph inside the sequence leg (every frame re-captures: short graphs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-sync", action="store_true", help="reference behaviour: read num_rendered every forward")
    ap.add_argument("--image-loss", default="fused", choices=["torch", "fused"])
    ap.add_argument("--physics-once", action="store_true",
                    help="add the view-independent physics terms once per iteration instead of once per view")
    ap.add_argument("--no-distance", action="store_true", help="drop the distance_loss term (round-1 behaviour)")
    ap.add_argument("--no-static-split", action="store_true", help="bin the static background Gaussians every iteration")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--graph-iters", type=int, default=20,
                    help="iterations recorded per hipGraph (reduced to a divisor of --steps; 1 in multi-GPU runs)")
    ap.add_argument("--views", default="batched", choices=["batched", "branches", "serial"],
                    help="the views of an iteration: one view-batched launch sequence (default), one rasteriser call "
                         "per view on parallel streams / graph branches, or one call per view in series")
    ap.add_argument("--torch-adam", action="store_true", help="gradient mean + optimiser step with torch ops / torch.optim.Adam")
    ap.add_argument("--unfused-physics", action="store_true",
                    help="physics terms as separate autograd nodes (the reference's op-by-op structure)")
    a = ap.parse_args()
    default_seq = a.frames < 0
    if a.iters_per_frame <= 0:
        a.iters_per_frame = 250 if default_seq else 1000
    # coherent: where the repair launch fits one round of workgroups (rasterizer.coherent_sort_pays); -always: wherever it runs
    a.coh = {"coherent": 1, "coherent-always": 2, "radix": 0}[a.sort]
    if a.sort == "coherent-always":
        a.sort = "coherent"

    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # A bare `python bench.py --gpus N` IS an N-rank run: start one process per GPU under torch.distributed.run
        # (RCCL over xGMI), rank 0 prints the one JSON line.  Never run fewer ranks than asked for.
        import socket
        n_dev = torch.cuda.device_count()
        if n_dev < a.gpus and os.environ.get("FNX_SINGLE_DEVICE") != "1":
            raise SystemExit(f"--gpus {a.gpus} but only {n_dev} device(s) visible (FNX_SINGLE_DEVICE=1 FNX_DIST_BACKEND=gloo "
                             "runs every rank on device 0: a smoke test of the multi-rank path, not a measurement)")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # developer smoke test of the multi-rank code path on a one-GPU box: FNX_SINGLE_DEVICE=1 puts every rank on cuda:0,
    # FNX_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device); the numbers mean nothing then
    if os.environ.get("FNX_SINGLE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("FNX_DIST_BACKEND", "nccl")
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("FNX_FORCE_DIST") == "1"  # the latter: 1-rank smoke of the RCCL path
    if use_dist:
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"  # keep stdout to the one JSON line (the banner goes to stdout)
        # RCCL writes its warnings to stdout; send them to a file so that stdout stays the one JSON line
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/fnx_rccl_debug_%h_%p.log")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from fluidnexus_amd import _lib, rasterizer
    from fluidnexus_amd.harness import shard_views
    from fluidnexus_amd.renderer import pipes
    _lib.raster()  # fail loudly if the HIP library is missing
    rasterizer.set_blend_math(a.blend_math)
    rasterizer.set_lean_geometry(not a.full_geometry)
    rasterizer.set_coherent_sort(a.coh)
    if a.deep_kernel is not None:
        rasterizer.set_deep_kernel(a.deep_kernel)
    if a.no_static_split:
        pipes.set_static_split(False)

    cfg_id = int(a.config) if a.config != "auto" else {4: 4, 8: 5}.get(world, 3)
    C = CONFIGS[cfg_id]
    if cfg_id == 2:
        a.stage = "first"
    scaling = "strong" if a.scaling == "auto" else a.scaling
    if world == 1:
        scaling = "strong" if a.scaling == "auto" else a.scaling  # one rank: the two coincide
    nominal_views = C["views"]
    n_views = nominal_views * world if scaling == "weak" else nominal_views
    shard_world = world
    graph_ar = None
    if use_dist:
        # multi-rank: the all-reduce goes INSIDE the hipGraph (one launch gap per k iterations instead of a replay + an eager
        # collective + an eager step per iteration) if a captured collective completes on every rank: tried first on a
        # throw-away process group, with a deadline (harness.graph_allreduce_self_test)
        from fluidnexus_amd import harness as _H
        graph_ar = _H.graph_allreduce_self_test(dev) if _H._GRAPH_ALLREDUCE == "auto" else _H._GRAPH_ALLREDUCE == "1"
    gm, cams, loop = build_workload(cfg_id, n_views, dev, rank, world, a, use_dist)
    loop_views = shard_views(len(cams), rank, world)
    if a.emulate_world > 1:
        assert world == 1, "--emulate-world is a single-process mode"
        shard_world = a.emulate_world
        loop_views = loop.view_subset = shard_views(len(cams), a.emulate_rank, shard_world)
        loop.emulated = (a.emulate_rank, shard_world)
    loop.make_targets()
    if not a.host_sync:
        rasterizer.set_host_sync(False)

    for _ in range(max(a.warmup, 1)):  # at least one eager pass sizes the binning buffers before anything is captured
        loop.iteration()
    if not a.host_sync:
        rasterizer.check_status()  # also records the binning high-water mark and the views' depth-key spans
        # three 9-bit passes order any view whose keys span < 2^27 ulps: the warm-up has shown how wide this scene's are
        if 0 < rasterizer.max_sort_span_bits <= rasterizer.SORT_NARROW_MAX_BITS and not a.sort_four_passes:
            rasterizer.set_sort_narrow(True)  # a later view that needs the fourth pass fails the run (check_status)
    sort_note = None
    if a.sort == "coherent":
        # the coherent sort is exact whatever the scene does, but a scene whose splats jump further than its repair window
        # pays a full in-launch sort per call: look at the warm-up's counters before committing to it
        calls, falls = rasterizer.coherent_sort_counters()
        falls = max(0, falls - rasterizer.coherent_sort_states())  # a state's first repair call may pay it once (see there)
        if rasterizer.coherent_sort_states() == 0:
            sort_note = ("coherent sort not used: more repair workgroups per launch than pay "
                         f"(rasterizer.coherent_sort_pays, limit {rasterizer.COHERENT_MAX_WORKGROUPS})")
        elif calls and falls > 0.02 * calls:
            rasterizer.set_coherent_sort(False)
            sort_note = f"coherent sort switched off after the warm-up: {falls} in-launch full sorts in {calls} view calls"
    graph_mode = False
    if loop.capturable:
        try:
            # several iterations per graph: one launch gap per replay; the timed region still runs exactly --steps
            gi = max(1, min(int(a.graph_iters), a.steps))  # a remainder of --steps is run eagerly at the end
            loop.capture(warmup=1, iterations=gi)
            for _ in range(2):
                loop.iteration()
            rasterizer.check_status()
            graph_mode = True
        except Exception as e:  # fall back to eager launches, say so in the JSON
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            loop.use_graph(False)
            if hasattr(loop, "parallel_views"):
                loop.parallel_views = False
    if not graph_mode:
        _lib.profile_enable(True)
    def timed_pass():
        """EXACTLY --steps steps between two barrier + synchronize brackets; max over the ranks."""
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps_done = 0
        while steps_done < a.steps:
            if graph_mode and a.steps - steps_done < loop.iterations_per_call:
                loop.use_graph(False)  # fewer steps left than one graph holds: the same iteration, launched eagerly
            steps_done += loop.iterations_per_call
            loop.iteration()
        assert steps_done == a.steps
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        d = time.perf_counter() - t0
        if graph_mode:
            loop.use_graph(True)
        if not a.host_sync:
            rasterizer.check_status()  # raises if a replayed forward overflowed its binning capacity
        if use_dist:
            tmax = torch.tensor([d], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            d = float(tmax.item())
        return d

    # The timed region is run --repeats times (a 20-step region is 20 ms; one sample cannot show its own
    # noise): `value` / `ms_per_step` are the MEDIAN pass, `spread` carries every pass.  Each pass times exactly --steps
    # steps with the contract's brackets.
    pass_s = [timed_pass() for _ in range(max(1, int(a.repeats)))]
    dt = sorted(pass_s)[len(pass_s) // 2]
    spread = {"repeats": len(pass_s), "iters_per_s": [a.steps / d for d in pass_s],
              "min": a.steps / max(pass_s), "max": a.steps / min(pass_s), "value_is": "median pass"}

    # Roofline of the dominant kernel (blend backward), measured live with HIP events on its stream.  Events cannot
    # be read back from a replayed graph, so in graph mode the same iteration is run eagerly a few more times
    # (outside the timed region) with the event hooks on.
    if graph_mode:
        loop.use_graph(False)
        if hasattr(loop, "parallel_views"):
            loop.parallel_views = False  # kernels one at a time, so the event pairs time single kernels
        _lib.profile_enable(True)
        rasterizer.keep_last_blobs(True)
        for _ in range(5):
            loop.iteration()
        torch.cuda.synchronize()
    walked = rasterizer.walked_entries()  # the blend kernels' own counts of the list entries they read (last eager iteration)
    rasterizer.keep_last_blobs(False)
    prof = {name: _lib.profile_read(i) for i, name in enumerate(("blend_forward", "blend_backward", "sort_and_counts",
                                                                 "preprocess", "emit", "blend_forward_ch1",
                                                                 "blend_backward_ch1"))}
    _lib.profile_enable(False)
    Cn = C["channels"]
    if Cn == 1:  # the configuration's main render is the 1-channel one
        prof["blend_forward"], prof["blend_backward"] = prof.pop("blend_forward_ch1"), prof.pop("blend_backward_ch1")
    view_mode = getattr(loop, "view_mode", "batched")
    views_per_launch = len(loop_views) if view_mode == "batched" else 1
    # instance / visible counts of this rank's views (one-set binning over all splats: the algorithmic byte count of
    # SURVEY 8(d) does not depend on how the implementation splits the work)
    R_views, P_vis_views = [], []
    xyz, opac, scales, rots, cols = scene_arrays(gm, cfg_id, a.stage)
    P_total = xyz.shape[0]
    if loop_views:
        from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
        from fluidnexus_amd.renderer.pipes import _settings
        _, GRsetting, GRzer = get_render_pipe("render_fluid" if Cn == 1 else "render_dynamics")
        t = [torch.tensor(x, device=dev) for x in (xyz, opac, scales, rots, cols)]
        with torch.no_grad():
            for v in loop_views:
                rz = GRzer(_settings(GRsetting, cams[v], loop.background, 1.0, 0))
                _, radii, _ = rz(means3D=t[0], means2D=torch.zeros_like(t[0]), opacities=t[1], colors_precomp=t[4],
                                 scales=t[2], rotations=t[3])
                P_vis_views.append(int((radii > 0).sum().item()))
                rasterizer.check_status()
                R_views.append(rasterizer.last_num_rendered)
    R = sum(R_views) / max(len(R_views), 1)
    P_vis = sum(P_vis_views) / max(len(P_vis_views), 1)
    bwd_ms, bwd_n = prof["blend_backward"]
    fwd_ms, fwd_n = prof["blend_forward"]
    fwd_s, bwd_s = (fwd_ms / max(fwd_n, 1)) * 1e-3, (bwd_ms / max(bwd_n, 1)) * 1e-3
    # SURVEY 8(d), per view: the blend FORWARD reads per instance id 4 + xy 8 + conic_opacity 16 + depth 4 + colour 4C and
    # writes per pixel colour C + depth + final_T + n_contrib; the blend BACKWARD reads the same per instance, per pixel
    # dL_dpix C + final_T + n_contrib, and writes per visible splat 2+3+1+C accumulated gradients.  A view-batched launch
    # processes all of the rank's views.
    alg = {"blend_forward": int(views_per_launch * (R * (32 + 4 * Cn) + SIZE * SIZE * 4 * (Cn + 3))),
           "blend_backward": int(views_per_launch * (R * (32 + 4 * Cn) + SIZE * SIZE * 4 * (Cn + 2) + P_vis * 4 * (6 + Cn)))}
    # ... and what the kernels actually read: both stop early (saturated pixels, the gradient limit), so the lists they walk
    # are shorter than R.  Counted by the kernels (header words 10 / 11), one eager iteration outside the timed region.
    touched = None
    if walked is not None and view_mode == "batched":
        fw, bw = sum(walked[0]), walked[1]
        touched = {"blend_forward": int(fw * (32 + 4 * Cn) + views_per_launch * SIZE * SIZE * 4 * (Cn + 3)),
                   "blend_backward": int(bw * (32 + 4 * Cn) + views_per_launch * (SIZE * SIZE * 4 * (Cn + 2) + P_vis * 4 * (6 + Cn))),
                   "entries": {"blend_forward": int(fw), "blend_backward": int(bw), "instances": int(R * views_per_launch)}}
    # the dominant kernel is the one that takes longer, measured
    dom = "blend_forward" if fwd_s >= bwd_s else "blend_backward"
    avg_s = fwd_s if dom == "blend_forward" else bwd_s
    alg_bytes = alg[dom]
    achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
    # whole-iteration algorithmic bytes (B_view = 324 P + 132 R + 44 HW for ch3, 308 P + 116 R + 28 HW for ch1)
    per_view = (324 * P_total + 132 * R + 44 * SIZE * SIZE) if Cn == 3 else (308 * P_total + 116 * R + 28 * SIZE * SIZE)
    iter_bytes = per_view * len(cams)
    # HBM traffic / VALU instructions of the two kernels: separate rocprofv3 --pmc passes of this command
    # (tools/collect_profiles.sh -> profiles/<tag>_pmc_traffic.json, <tag>_sq_counters.json), null if absent
    # backward modes (raster_backward.hip): 3 = positions only (the position stages: the flush adds straight into
    # dL/dmeans3D), 1 = geometry only (the same with FNX_SCREEN_GRAD=1), 2 = fixed positions (visual-particle stage:
    # appearance + shape gradients), 0 = everything (per-view autograd physics)
    bwd_mode = 2 if a.stage == "visual" else (0 if a.unfused_physics else
                                              1 if os.environ.get("FNX_SCREEN_GRAD", "0") == "1" else 3)
    fast_s = 'true' if a.blend_math == 'fast' else 'false'
    split_s = 'true' if (pipes._STATIC_SPLIT and cfg_id != 2 and a.stage != "first") else 'false'
    from fluidnexus_amd import harness as _Hn
    dual_s = 'true' if (cfg_id == 5 and a.stage == "physical" and _Hn._DUAL_FUSED and split_s == 'true') else 'false'
    lanes = rasterizer.get_backward_form() == "lanes"
    knames = {"blend_forward": f"fnx::blend_forward_kernel<{Cn}, {split_s}, {fast_s}, {dual_s}, false>",
              "blend_backward": (f"fnx::blend_backward_lanes_kernel<{Cn}, {bwd_mode}, {fast_s}, {dual_s}>" if lanes else
                                 f"fnx::blend_backward_kernel<{Cn}, {bwd_mode}, {fast_s}, {dual_s}>")}
    kname = knames[dom]
    suffix = ("" if cfg_id == 3 else f"_config{cfg_id}") + ("" if a.stage == "physical" or cfg_id == 2 else f"_{a.stage}")
    counters = {}
    # (the counter files are read from profiles/, not collected in this run -- they carry the hash of the
    # kernel sources they were collected on, and a file whose hash differs from this checkout's is STALE: its numbers are
    # left out and the record says so)
    from fluidnexus_amd.build import csrc_hash
    sha_now, traffic_stale = csrc_hash(), False
    for which, kn in knames.items():
        ent = {}
        try:
            with open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}{suffix}_pmc_traffic.json")) as f:
                tj = json.load(f)
            t = tj.get("void " + kn)
            if tj.get("_csrc_sha16") != sha_now:
                t, traffic_stale = None, True
            if t:
                ent["hbm_bytes_per_launch"] = t["fetch_bytes"] + t["write_bytes"]
        except OSError:
            pass
        try:
            with open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}{suffix}_sq_counters.json")) as f:
                tj = json.load(f)
            t = tj.get("void " + kn)
            if tj.get("_csrc_sha16") != sha_now:
                t, traffic_stale = None, True
            if t:
                ent["valu_wave_instructions_per_launch"] = t["SQ_INSTS_VALU"]
                ent["counter_run_launch_us"] = t["us"]
                for src, dst in (("valu_busy", "valu_busy"), ("lds_busy", "lds_busy"), ("wait_frac", "wave_wait_frac")):
                    if src in t:
                        ent[dst] = t[src]
        except OSError:
            pass
        counters[which] = ent
    cdom = counters[dom]
    traffic = cdom.get("hbm_bytes_per_launch")
    src_note = (f"profiles/{PROFILE_TAG}{suffix}_pmc_traffic.json / _sq_counters.json: separate rocprofv3 --pmc passes of this "
                "command on the builder's box (tools/collect_profiles.sh), NOT collected in this run")
    frac = achieved / HBM_PEAK_GBS
    roofline = {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac,
                "traffic": traffic, "kernel": kname, "avg_launch_us": avg_s * 1e6,
                "launches": fwd_n if dom == "blend_forward" else bwd_n, "views_per_launch": views_per_launch,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_chosen_by": f"measured launch time: blend forward {fwd_s * 1e6:.1f} us, blend backward {bwd_s * 1e6:.1f} us",
                "note": "achieved / frac price the ALGORITHMIC bytes of SURVEY 8(d) (every instance read once) against HBM "
                        "peak, as the contract defines them.  The blends stop early (saturated pixels, gradient limit), so the "
                        "bytes they must touch are fewer: see `touched` (the kernels' own entry counts).  Neither kernel is "
                        "bound by HBM: the primary bound is the compute units' VALU / LDS pipelines, see `primary_bound`",
                "traffic_source": src_note if traffic else None,
                "traffic_stale": traffic_stale, "csrc_sha16": sha_now,
                "hbm_traffic_frac": (traffic / avg_s / 1e9 / HBM_PEAK_GBS) if traffic and avg_s > 0 else None,
                "other_kernels_avg_us": {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in prof.items()}}
    if touched:
        tb = touched[dom]
        roofline["touched"] = {"algorithmic_bytes_touched": tb, "GBps": tb / avg_s / 1e9 if avg_s > 0 else None,
                               "frac": tb / avg_s / 1e9 / HBM_PEAK_GBS if avg_s > 0 else None,
                               "list_entries": touched["entries"],
                               "traffic_over_touched": (traffic / tb) if traffic else None}
        if frac > 1.0:  # the formula prices bytes the kernel never reads (e.g. a scene whose tiles saturate early)
            roofline["frac_of_formula"], roofline["achieved_of_formula"] = frac, achieved
            roofline["frac"], roofline["achieved"] = roofline["touched"]["frac"], roofline["touched"]["GBps"]
            roofline["note"] += "; the 8(d) formula gave a fraction above 1 on this scene: frac / achieved are the TOUCHED bytes'"
    elif frac > 1.0:
        roofline["frac_of_formula"], roofline["frac"] = frac, None
    if cdom.get("valu_wave_instructions_per_launch") and avg_s > 0:
        rate = cdom["valu_wave_instructions_per_launch"] / avg_s
        roofline["primary_bound"] = {"valu_busy": cdom.get("valu_busy"), "lds_busy": cdom.get("lds_busy"),
                                     "wave_wait_frac": cdom.get("wave_wait_frac"),
                                     "valu_wave_instructions_per_launch": cdom["valu_wave_instructions_per_launch"],
                                     "frac_of_measured_issue_rate": rate / VALU_MEASURED_WAVE_INSTR,
                                     "frac_of_spec_issue_rate": rate / VALU_SPEC_WAVE_INSTR,
                                     "source": src_note,
                                     "spec": "1024 SIMDs x 1 wave64 VALU instruction / 2 cycles x 2.4 GHz = 1229 G wave-instr/s "
                                             "(157 TFLOP/s fp32); measured on this chip 933 G/s (tools/micro/pk_rate.hip)"}
    other = "blend_backward" if dom == "blend_forward" else "blend_forward"
    o_s = bwd_s if dom == "blend_forward" else fwd_s
    roofline["second_kernel"] = {"kernel": knames[other], "avg_launch_us": o_s * 1e6, "algorithmic_bytes_per_launch": alg[other],
                                 "frac": alg[other] / o_s / 1e9 / HBM_PEAK_GBS if o_s > 0 else None,
                                 "touched_frac": (touched[other] / o_s / 1e9 / HBM_PEAK_GBS) if touched and o_s > 0 else None,
                                 "traffic": counters[other].get("hbm_bytes_per_launch")}
    nominal_iters = (len(cams) / nominal_views) * a.steps  # weak scaling: N nominal batches per step
    value = nominal_iters / dt
    roofline["iteration_algorithmic"] = {"bytes": int(iter_bytes), "GBps": iter_bytes * a.steps / dt / 1e9,
                                         "frac_of_hbm_peak": iter_bytes * a.steps / dt / 1e9 / HBM_PEAK_GBS}
    metric = {2: "train iters/sec (ScalarReal: 100k Gaussians x 5 views @512^2, ch1, render-only loss)",
              3: "train iters/sec (300k Gaussians x 5 views @512^2, physics losses on)",
              4: "train iters/sec (300k Gaussians x 5 views @512^2, physics losses on)",
              5: "train iters/sec (500k Gaussians x 8 views @512^2, ch3 + ch1 rasterisers, physics losses on)"}[cfg_id]
    stage_note = ""
    if cfg_id != 2 and a.stage == "visual":
        metric = metric.replace("physics losses on", "visual-particle stage: attribute leaves, consistency + scale terms")
        stage_note = (" -- VISUAL-PARTICLE STAGE of this frame (train_visual_particle.py:133-222): positions fixed, colour / "
                      "opacity / scales / rotation of the fluid Gaussians optimised, L1 + D-SSIM (RGB) + consistency + scale "
                      "regulariser; not the stage BASELINE's metric is quoted on")
    elif cfg_id != 2 and a.stage == "first":
        metric = metric.replace("physics losses on", "first-frame stage: visual positions are the leaf, no physics terms")
        stage_note = (" -- FIRST-FRAME STAGE of this scene (entries_fluid_nexus/train_physical_particle.py:103-163): visual "
                      "particle positions optimised, grey-mean L1 + D-SSIM + distance loss; not the stage BASELINE's metric "
                      "is quoted on")
    sort_fallbacks = None
    try:  # how often the coherent sort had to fall back to its in-launch full sort (per view batch: [calls, fallbacks] per view)
        from fluidnexus_amd.renderer.pipes import _VIEW_BATCH_CACHE
        sort_fallbacks = [[list(c) for c in vb.sort_counters(key[0], key[1], sort_key=key[2])] for vb, _, _ in _VIEW_BATCH_CACHE.values()
                          for key in list(vb._sort_state)]
    except Exception as e:
        print(f"[bench] sort counters unavailable: {type(e).__name__}: {e}", file=sys.stderr)
    dist_lists = None
    try:  # Verlet pair lists of the distance loss: how many of the run's calls had to rebuild them (device counters)
        from fluidnexus_amd import physics as _ph
        if _ph._DIST_LISTS and _ph._DIST_VERLET and not a.no_distance:
            cs = _ph.distance_verlet_counters()
            dist_lists = {"what": "fnx_distance_loss_verlet: pairs within threshold + skin kept between calls, validity checked "
                                  "on the device every call, rebuilt inside the same launch sequence when a point has moved "
                                  "further than skin / 2",
                          "calls": sum(c[2] for c in cs), "rebuilds": sum(c[3] for c in cs),
                          "points_over_capacity_at_last_rebuild": sum(c[4] for c in cs), "slots_per_point": _ph.DIST_VERLET_K}
    except Exception as e:
        print(f"[bench] distance pair-list counters unavailable: {type(e).__name__}: {e}", file=sys.stderr)
    knn_watch = None
    if cfg_id != 2 and a.stage == "physical" and hasattr(gm, "check_knn_k") and getattr(gm, "_knn_flags", None) is not None:
        gm.check_knn_k()  # raises if a fused search met a list longer than KNN_K anywhere in the run (device flag)
        knn_watch = (f"armed: the fused density / interpolation kernels counted every query's neighbours in every iteration of "
                     f"this run; no list exceeded KNN_K = {int(gm.KNN_K)} (gm.check_knn_k())")
    knn = None
    if cfg_id != 2 and a.stage == "physical":
        knn = gm.knn_k_report()  # outside the timed region: are the reference's neighbour lists below their cap here?
    ms = None
    try:
        if loop_views:
            ms = rasterise_timing(gm, cams, loop_views, cfg_id, Cn, loop.background, a.stage)
    except Exception as e:
        print(f"[bench] rasteriser-only timing failed: {type(e).__name__}: {e}", file=sys.stderr)
    out = {
        "metric": metric,
        "value": value, "unit": "iters/s", "n_gpus": world, "ranks": dist.get_world_size() if use_dist else 1,
        "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "spread": spread, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": C["workload"] + stage_note, "baseline_config": cfg_id, "stage": a.stage,
                   "views_this_rank": len(loop_views), "global_views_per_step": len(cams), "image": f"{SIZE}x{SIZE}",
                   "gaussians": P_total, "num_rendered_per_view": R_views, "visible_per_view": P_vis_views,
                   "parallelism": (f"views sharded round-robin over {shard_world} rank(s)"
                                   + (f" (emulated: rank {a.emulate_rank}'s share, no communication)" if a.emulate_world > 1 else "")
                                   + ", RCCL all-reduce of the leaf gradient"),
                   "all_reduce": (None if not use_dist else
                                  ("inside the hipGraph (local phase | RCCL all-reduce | fused step, k iterations per graph); a captured "
                                   "collective was first replayed on a throw-away process group with a deadline on every rank"
                                   if (graph_ar and getattr(loop, "graph_finish", "x") is None and graph_mode) else
                                   "eager, between a graph replay of the local phase and the eager fused step"
                                   + ("" if graph_ar else " (the captured-collective self-test did not pass or is switched off)"))),
                   "shared_terms": ("physics terms + distance loss on every rank, added once per local view"
                                    if getattr(loop, "shared_terms_rank", None) is None else
                                    (f"spread: gas / next-gas + exyz / distance terms on ranks {_Hn.spread_owners(len(cams), shard_world)} "
                                     "(each added `batch` times by its owner, the all-reduce sums them)"
                                     if loop.shared_terms_rank == "spread" else
                                     f"physics terms + distance loss evaluated on rank {loop.shared_terms_rank} only (the rank with "
                                     "the fewest views), added `batch` times; the all-reduce distributes the sum")),
                   "host_sync": bool(a.host_sync), "image_loss": a.image_loss,
                   "depth_sort": sort_note or (("coherent: one launch per call repairs the previous call's order, verified on the device "
                                   f"(in-launch full sorts per view over the run: {sort_fallbacks}); first call: " if a.sort == "coherent" else "")
                                  + f"9-bit radix passes; key span of the views <= 2^{rasterizer.max_sort_span_bits} ulps; fourth pass "
                                  + ("not launched (device-checked)" if (0 < rasterizer.max_sort_span_bits <= rasterizer.SORT_NARROW_MAX_BITS
                                                                         and not a.sort_four_passes and not a.host_sync) else "launched")),
                   "blend_math": {"fast": "fast: fused multiply-adds + v_exp_f32 in the two blend kernels; pixels within 2e-5 of the "
                                          "bit-exact mode / oracle except counted threshold flips, binning bit-exact "
                                          "(tests/test_fast_math_gpu.py)",
                                  "exact": "exact: bit-reproducible blend arithmetic (equal to the CPU oracle bit for bit)"}[a.blend_math],
                   "static_split": bool(pipes._STATIC_SPLIT and cfg_id != 2),
                   "distance_loss": not a.no_distance, "distance_pair_lists": dist_lists,
                   "scene": a.scene if cfg_id in (3, 4) else "backdrop",
                   "scene_note": ("background Gaussians BEHIND the plume: the fluid is visible to every camera and its image "
                                  "gradient is non-zero" if (a.scene == "backdrop" or cfg_id not in (3, 4)) else
                                  "round-1 layout: background cloud around the plume (the fluid is occluded in every view, "
                                  "tiles saturate early); kept for like-for-like comparison with round 1"),
                   "launch": (f"hipGraph replay, {loop.graph_iterations} whole iteration(s) per graph" if graph_mode
                              else "eager"),
                   "views": {"batched": "one view-batched launch sequence per iteration (view = grid dimension y)",
                             "branches": "one rasteriser call per view, views as parallel graph branches",
                             "serial": "one rasteriser call per view, in series"}[view_mode],
                   "knn_k": knn, "knn_watch": knn_watch,
                   "physics": None if (cfg_id == 2 or a.stage != "physical") else
                   (("value and gradient evaluated once per iteration, the gradient added once per view "
                     "(equal to the reference's per-view evaluation, tpp:368-404)" if not a.physics_once
                     else "added once per iteration")
                    + (", op-by-op autograd" if a.unfused_physics else ", one fused launch sequence"))},
        "roofline": roofline,
        "rasterise_ms_per_view": dict(
            ms or {},
            hot_loop_forward=sum(prof[k][0] for k in ("preprocess", "sort_and_counts", "emit", "blend_forward", "blend_forward_ch1")
                                 if k in prof) / max(prof["blend_forward"][1], 1) / max(views_per_launch, 1),
            **{"hot_loop_backward_blend_" + ("all_gradients" if bwd_mode == 0 else "geometry_only" if bwd_mode == 1 else
                                             "appearance_and_shape" if bwd_mode == 2 else "positions_only"):
               bwd_ms / max(bwd_n, 1) / max(views_per_launch, 1)}),
    }
    # top-level switches of the number above (one record must not mix modes silently)
    out["modes"] = {"blend_math": a.blend_math, "lean_geometry": not a.full_geometry,
                    "depth_sort": "radix" if (a.sort == "radix" or sort_note) else "coherent",
                    "sort_narrow_max_bits": rasterizer.SORT_NARROW_MAX_BITS, "graph": bool(graph_mode),
                    "static_split": bool(pipes._STATIC_SPLIT and cfg_id != 2)}
    if (a.blend_math == "fast" and graph_mode and not a.no_exact_leg and world == 1 and a.emulate_world <= 1
            and hasattr(loop, "capture")):
        try:  # the same loop with the bit-exact blend arithmetic (what the oracle parity tests run), re-captured
            rasterizer.set_blend_math("exact")
            loop.capture(warmup=1, iterations=loop.graph_iterations)
            loop.iteration()
            torch.cuda.synchronize()
            n_calls = max(1, 100 // loop.iterations_per_call)
            t0 = time.perf_counter()
            for _ in range(n_calls):
                loop.iteration()
            torch.cuda.synchronize()
            e_dt = (time.perf_counter() - t0) / (n_calls * loop.iterations_per_call)
            rasterizer.check_status()
            out["exact_mode"] = {"iters_per_s": 1.0 / e_dt, "ms_per_step": e_dt * 1e3, "steps": n_calls * loop.iterations_per_call,
                                 "what": "the same configuration with --blend-math exact (bit-equal to the CPU oracle), re-captured"}
        except Exception as e:
            print(f"[bench] exact-mode leg failed: {type(e).__name__}: {e}", file=sys.stderr)
        finally:
            rasterizer.set_blend_math(a.blend_math)
    if not a.no_drop_in and cfg_id in (3, 4) and a.stage == "physical" and world == 1 and a.emulate_world <= 1:
        try:
            out["drop_in"] = drop_in_timing(a, dev, cfg_id)
            out["drop_in_auto"] = drop_in_timing(a, dev, cfg_id, steps=12, auto=True)
        except Exception as e:
            import traceback
            traceback.print_exc()
            print(f"[bench] drop-in leg failed: {type(e).__name__}: {e}", file=sys.stderr)
    if a.emulate_world > 1:
        # what the emulation leaves out, priced with a stated model: one ring all-reduce of the leaf gradient per iteration
        # over xGMI (point-to-point links, ~153 GB/s each: MI355X_MICROARCH.md / the task's hardware notes), 2 (n - 1) steps of
        # bytes / n each at 80 % of a link, 3 us per step, 10 us of launch + stream synchronisation around the collective
        n_r, nbytes = a.emulate_world, 4 * 3 * int(getattr(gm, "_estimate_xyz_nn", gm._visual_xyz).shape[0])
        t_comm = 10e-6 + 2 * (n_r - 1) * (3e-6 + nbytes / n_r / (0.8 * 153e9))
        out["emulated_collectives"] = {"all_reduce_bytes": nbytes, "ranks": n_r, "model_us": t_comm * 1e6,
                                       "iters_per_s_with_model": 1.0 / (dt / a.steps + t_comm),
                                       "model": "10 us + 2 (n - 1) x (3 us + bytes / n / (0.8 x 153 GB/s)): ring over point-to-point "
                                                "xGMI links; a bound, not a measurement"}
    if default_seq:  # the default record carries a short sequence leg: single-GPU config 3, graph replay
        a.frames = 3 if (cfg_id == 3 and world == 1 and graph_mode and a.stage == "physical" and a.emulate_world <= 1
                         and a.views == "batched" and not a.no_distance) else 0
    if a.frames > 0 and cfg_id != 2 and a.stage == "physical" and a.emulate_world <= 1:
        try:
            out["sequence"] = sequence_timing(a, dev, cfg_id, rank, world, use_dist, dt / a.steps * 1e3)
        except Exception as e:
            import traceback
            traceback.print_exc()
            print(f"[bench] sequence timing failed: {type(e).__name__}: {e}", file=sys.stderr)
    if a.sh_degree >= 0 and loop_views and cfg_id != 2:
        try:
            out["sh"] = sh_timing(gm, cams, loop_views, cfg_id, loop.background, a.sh_degree, a.stage)
        except Exception as e:
            print(f"[bench] SH timing failed: {type(e).__name__}: {e}", file=sys.stderr)
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            ras = cpu_baseline(gm, cams, loop.background, cfg_id, nominal_views, Cn, a.stage)
            out["cpu_baseline"] = ras
            if cfg_id in (3, 4) and a.stage == "physical" and not a.no_distance:
                # the headline configuration: the whole iteration on the host; the rasteriser-only
                # figure stays beside it
                try:
                    whole = cpu_baseline_whole_iteration(gm, cams, loop.cfg, nominal_views)
                    whole["rasteriser_only"] = ras
                    out["cpu_baseline"] = whole
                except Exception as e:
                    import traceback
                    traceback.print_exc()
                    print(f"[bench] whole-iteration CPU baseline failed: {type(e).__name__}: {e}", file=sys.stderr)
            try:
                out["cpu_baseline"]["config1"] = cpu_baseline_config1(dev)
            except Exception as e:
                print(f"[bench] config-1 CPU baseline failed: {type(e).__name__}: {e}", file=sys.stderr)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
