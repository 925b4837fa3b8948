#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native FluidNexus hot path.

Metric (BASELINE.json): train iters/sec, 300k Gaussians x 5 views @ 512^2 (config 3:
FluidNexus-Smoke frame, ch3 rasteriser, image + exyz + gas + next-gas losses, Adam step).
One "step" = one iteration of the per-frame optimisation loop over one batch of synthetic views,
exactly the op sequence of entries_fluid_nexus/train_physical_particle.py:329-432
(fluidnexus_amd/harness.py).  All inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU; Gaussians/particles replicated; every rank renders its own 5
views of a 5N-view batch (weak scaling: per-GPU work fixed); one RCCL all-reduce(sum) of the
leaf gradient per iteration, then the reference's 1/batch scaling and a replicated Adam step.
`value` counts 5-view iterations: (5N views per step / 5) * steps / seconds.

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

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

P_FLUID, P_BACKGROUND, VIEWS, SIZE = 200_000, 100_000, 5, 512
HIDDEN_DIMS = (20, 62, 20)  # 24,800 hidden particles (reference cap: 28,000)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(gm, cams, bg, seconds_budget=30.0):
    """The CPU oracle (oracle/raster_oracle.c, OpenMP over all host cores) timed on ONE view of the
    same workload: rasteriser forward + backward.  Reported as 5-view iterations per second of the
    rasteriser alone (losses / physics / Adam are not in the CPU sample)."""
    from oracle import raster_oracle as O
    O.build()
    cores = os.cpu_count() or 1
    O.set_threads(cores)
    cam = cams[0]
    with torch.no_grad():
        xyz = torch.cat([gm.get_visual_xyz_from_nn() / gm.scale_factor, gm.get_gs_xyz], 0).cpu().numpy()
        opac = torch.cat([gm.get_visual_opacity, gm.get_gs_opacity], 0).cpu().numpy()
        scales = torch.cat([gm.get_visual_scaling, gm.get_gs_scaling], 0).cpu().numpy()
        rots = torch.cat([gm.get_visual_rotation, gm.get_gs_rotation], 0).cpu().numpy()
        cols = torch.cat([gm.get_visual_color.repeat(1, 3), gm.get_gs_color], 0).cpu().numpy()
    tan = math.tan(cam.FoVx * 0.5)
    t0 = time.perf_counter()
    f = O.forward(xyz, opac, bg.cpu().numpy(), cam.world_view_transform.cpu().numpy(),
                  cam.full_proj_transform.cpu().numpy(), cam.camera_center.cpu().numpy(), SIZE, SIZE, tan, tan,
                  colors_precomp=cols, scales=scales, rotations=rots)
    t1 = time.perf_counter()
    O.backward(f, np.ones((3, SIZE, SIZE), np.float32))
    t2 = time.perf_counter()
    per_view = t2 - t0
    return {"value": 1.0 / (VIEWS * per_view), "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": f"oracle/raster_oracle.c (OpenMP, {cores} threads), rasteriser only: 1 of {VIEWS} views "
                      f"fwd {t1 - t0:.2f}s + bwd {t2 - t1:.2f}s, R={f['num_rendered']}; value = 1/({VIEWS} x that)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-sync", action="store_true", help="reference behaviour: read num_rendered every forward")
    ap.add_argument("--image-loss", default="auto", choices=["auto", "torch", "fused"])
    ap.add_argument("--physics-once", action="store_true",
                    help="evaluate the view-independent physics terms once per iteration instead of once per view")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--graph-iters", type=int, default=5,
                    help="iterations recorded per hipGraph (reduced to a divisor of --steps; 1 in multi-GPU runs)")
    ap.add_argument("--views", default="batched", choices=["batched", "branches", "serial"],
                    help="the views of an iteration: one view-batched launch sequence (default), one rasteriser call "
                         "per view on parallel streams / graph branches, or one call per view in series")
    ap.add_argument("--serial-views", action="store_true", help="same as --views serial")
    ap.add_argument("--torch-adam", action="store_true", help="gradient mean + optimiser step with torch ops / torch.optim.Adam")
    ap.add_argument("--unfused-physics", action="store_true",
                    help="physics terms as separate autograd nodes (the reference's op-by-op structure)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
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
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from fluidnexus_amd import _lib, rasterizer
    from fluidnexus_amd.harness import HotLoop, build_smoke_frame
    _lib.raster()  # fail loudly if the HIP library is missing

    # every rank renders VIEWS views of the VIEWS*world-view batch (weak scaling)
    gm, cams = build_smoke_frame(P_FLUID, P_BACKGROUND, HIDDEN_DIMS, n_views=VIEWS * world, size=SIZE, seed=0,
                                 device=dev)
    image_loss = a.image_loss
    if image_loss == "auto":
        try:
            from fluidnexus_amd import losses  # noqa: F401
            image_loss = "fused"
        except Exception:
            image_loss = "torch"
    view_mode = "serial" if a.serial_views else a.views
    if a.unfused_physics or image_loss != "fused":
        view_mode = "serial"  # the batched / branch modes build on the fused loss nodes
    if view_mode == "branches" and (a.no_graph or a.host_sync):
        view_mode = "serial"
    loop = HotLoop(gm, cams, rank=rank, world=world, force_all_reduce=use_dist, physics_per_view=not a.physics_once, image_loss=image_loss,
                   fused_physics=not a.unfused_physics, defer_visual_backward=not a.unfused_physics,
                   capturable=not (a.no_graph or a.host_sync),
                   parallel_views=view_mode == "branches", batched_views=view_mode == "batched",
                   fused_step=view_mode == "batched" and not (a.no_graph or a.host_sync) and not a.torch_adam)
    loop.make_targets()
    from fluidnexus_amd.harness import shard_views
    loop_views = shard_views(len(cams), rank, world)
    if not a.host_sync:
        rasterizer.set_host_sync(False)

    for _ in range(a.warmup):
        loop.iteration()
    if not a.host_sync:
        rasterizer.check_status()  # also records the binning high-water mark
    graph_mode = False
    if loop.capturable:
        try:
            # several iterations per graph: one launch gap per replay; the timed region still runs exactly --steps
            gi = max(1, int(a.graph_iters))
            while a.steps % gi:
                gi -= 1
            loop.capture(warmup=1, iterations=gi)
            for _ in range(2):
                loop.iteration()
            rasterizer.check_status()
            graph_mode = True
        except Exception as e:  # fall back to eager launches, say so in the JSON
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            loop.use_graph(False)
            loop.parallel_views = False
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    if not graph_mode:
        _lib.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps_done = 0
    while steps_done < a.steps:
        steps_done += loop.iterations_per_call
        loop.iteration()
    assert steps_done == a.steps
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if not a.host_sync:
        rasterizer.check_status()
    if use_dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # roofline of the dominant kernel (blend backward), measured live with HIP events on its stream.  Events
    # cannot be read back from a replayed graph, so in graph mode the same iteration is run eagerly a
    # few more times (outside the timed region) with the event hooks on.
    if graph_mode:
        loop.use_graph(False)
        loop.parallel_views = False  # kernels one at a time, so the event pairs time single kernels
        _lib.profile_enable(True)
        for _ in range(5):
            loop.iteration()
        torch.cuda.synchronize()
    prof = {name: _lib.profile_read(i) for i, name in enumerate(("blend_forward", "blend_backward", "sort_and_counts",
                                                                 "preprocess", "emit"))}
    _lib.profile_enable(False)
    # instance / visible counts of this rank's views (for the algorithmic byte count)
    R_views, P_vis_views = [], []
    with torch.no_grad():
        for v in loop_views:
            pkg = loop.render_func(cams[v], gm, None, loop.background, GRsetting=loop.GRsetting,
                                   GRzer=loop.GRzer, pos_type="guess_visual_nn", scale=True)
            P_vis_views.append(int((pkg["radii"] > 0).sum().item()))
            rasterizer.check_status()
            R_views.append(rasterizer.last_num_rendered)
    views_per_launch = len(loop_views) if view_mode == "batched" else 1
    R = sum(R_views) / len(R_views)
    P_vis = sum(P_vis_views) / len(P_vis_views)
    Cn = 3
    bwd_ms, bwd_n = prof["blend_backward"]
    # blend backward reads per instance id 4 + xy 8 + conic_opacity 16 + depth 4 + colour 4C,
    # per pixel dL_dpix C + final_T + n_contrib, and writes per visible splat 2+3+1+C accumulated gradients;
    # a view-batched launch processes all of the rank's views
    alg_bytes = int(views_per_launch * (R * (32 + 4 * Cn) + SIZE * SIZE * 4 * (Cn + 2) + P_vis * 4 * (6 + Cn)))
    avg_s = (bwd_ms / max(bwd_n, 1)) * 1e-3
    achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
    # HBM traffic of the same kernel from the PMC counters: a separate rocprofv3 --pmc run of this command
    # (profiles/r01_pmc_traffic.json, corrected as MI355X_MICROARCH.md prescribes); null if absent
    traffic, kname = None, "fnx::blend_backward_kernel<3, 1>" if not a.unfused_physics else "fnx::blend_backward_kernel<3, 0>"
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            t = json.load(f).get("void " + kname)
        if t:
            traffic = t["fetch_bytes"] + t["write_bytes"]
    except OSError:
        pass
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kname,
                "avg_launch_us": avg_s * 1e6, "launches": bwd_n, "views_per_launch": views_per_launch,
                "algorithmic_bytes_per_launch": alg_bytes,
                "other_kernels_avg_us": {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in prof.items()}}

    views_per_step = VIEWS * world
    value = (views_per_step / VIEWS) * a.steps / dt
    out = {
        "metric": "train iters/sec (300k Gaussians x 5 views @512^2, physics losses on)",
        "value": value, "unit": "iters/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: FluidNexus-Smoke frame, 200k fluid + 100k background Gaussians, "
                               f"{HIDDEN_DIMS[0] * HIDDEN_DIMS[1] * HIDDEN_DIMS[2]} hidden particles, ch3, "
                               "L1+D-SSIM + exyz + gas + next-gas losses, Adam",
                   "views_per_rank": VIEWS, "global_views_per_step": views_per_step, "image": f"{SIZE}x{SIZE}",
                   "num_rendered_per_view": R_views, "visible_per_view": P_vis_views,
                   "parallelism": f"views sharded over {world} rank(s), RCCL all-reduce of the leaf gradient",
                   "host_sync": bool(a.host_sync), "image_loss": image_loss,
                   "launch": (f"hipGraph replay, {loop.graph_iterations} whole iteration(s) per graph" if graph_mode
                              else "eager"),
                   "views": {"batched": "one view-batched launch sequence per iteration (view = grid dimension y)",
                             "branches": "one rasteriser call per view, views as parallel graph branches",
                             "serial": "one rasteriser call per view, in series"}[view_mode],
                   "physics": ("once per iteration" if a.physics_once else "per view (as the reference)")
                   + (", op-by-op autograd" if a.unfused_physics else ", one fused autograd node")},
        "roofline": roofline,
        "rasterise_ms_per_view": {"forward": sum(prof[k][0] for k in ("preprocess", "sort_and_counts", "emit", "blend_forward"))
                                  / max(prof["blend_forward"][1], 1) / views_per_launch,
                                  "backward_blend": bwd_ms / max(bwd_n, 1) / views_per_launch},
    }
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(gm, cams, loop.background)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
