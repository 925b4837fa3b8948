"""Synthetic driver of the per-frame optimisation hot loop.

Reproduces, op for op, the per-iteration sequence of the reference's physical-particle stage
(FluidDynamics/entries_fluid_nexus/train_physical_particle.py:329-432):
  zero grad cache -> for each view of the batch: render_dynamics(pos_type="guess_visual_nn",
  scale=True) -> grey-mean image loss (L1 + D-SSIM) -> exyz anchor L2 -> gas constraint L2 ->
  next-tick gas constraint L2 -> backward -> cache gradient -> mean over the batch -> Adam step,
and of the visual-particle stage (train_visual_particle.py:133-222) for the level-two attributes.
Dataset loading, TensorBoard and PNG dumps of the entry scripts are out of scope; the optional
`.item()` logging syncs of the reference (tpp:410-424) are off by default.

Multi-GPU: particles/Gaussians replicated, the views of the batch sharded
round-robin over ranks, one all-reduce(sum) of the leaf gradient per iteration followed by the
reference's 1/batch scaling (gm_dynamics.py:461-472)."""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import os
import torch

from ._lib import raw_stream as _lib_raw_stream
import torch.distributed as dist

from . import synthetic as S
from .gaussian_splatting.gm_dynamics import GaussianModel
from .helpers.helper_pipe import get_render_pipe
from .utils.loss_utils import l1_loss, l2_loss, ssim

# constants of configs/fluid_nexus_smoke_dynamics.json + arguments/__init__.py
SMOKE = dict(H=2.0, KNN_K=100, p0=1.5, secs=0.033, k=3, lambda_dssim=0.2, lambda_image=1.0, lambda_exyz=0.1,
             lambda_gas_constraints=1.0, lambda_next_gas_constraints=0.1, lambda_current_distance=0.1,
             distance_threshold_visual=0.002,
             position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
             position_lr_max_steps=30000)


# visual-particle ("level two") stage constants of the same config
SMOKE_L2 = dict(lambda_dssim=0.2, lambda_image=1.0, lambda_consistency_color=0.1, lambda_consistency_opacity=0.1,
                lambda_consistency_scales=0.1, lambda_consistency_rotation=0.1, lambda_reg_scaling=0.0,
                scaling_reg_ratio_threshold=5.0, visual_color_lr=0.0025, visual_opacity_lr=0.05,
                visual_scales_lr=0.005, visual_rotation_lr=0.001)


# first-frame stage of configs/scalar_real.json (ScalarReal, ch1): positions of the visual particles are the leaf
SCALAR_REAL = dict(lambda_dssim=0.2, lambda_first_distance=1.0, distance_threshold_visual=0.00625,
                   position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                   position_lr_max_steps=30000)


# Rough kernel-time weights (us on an MI355X, config 3 / 5) that decide which rank takes which view-independent term
# when they are spread (HotLoop(shared_terms_rank="spread")): one view of the rasteriser + image loss, the gas term
# (density over the optimised positions + its backward), the next-gas + exyz terms, the distance loss.
_SPREAD_COST = dict(view=150.0, gas=150.0, next=90.0, distance=60.0)


def spread_owners(n_views: int, world: int):
    """{"gas", "next" (+ exyz), "distance"} -> rank, for the view-independent terms of an iteration spread over the ranks
    of a view-sharded run: greedy, the heaviest term first onto the rank with the least kernel time so far (its views
    first).  The distance term needs a render on its rank.  Deterministic: every rank computes the same table."""
    load = [len(shard_views(n_views, r, world)) * _SPREAD_COST["view"] for r in range(world)]
    owners = {}
    for term in ("gas", "next", "distance"):
        ranks = [r for r in range(world) if term != "distance" or len(shard_views(n_views, r, world)) > 0]
        r = min(ranks, key=lambda q: (load[q], -q))
        owners[term] = r
        load[r] += _SPREAD_COST[term]
    return owners


def shard_views(n_views: int, rank: int, world: int):
    """view v of the iteration's batch -> rank v mod world (5 views on 4 GPUs = 2/1/1/1)."""
    return [v for v in range(n_views) if v % world == rank]


def build_scalar_real_frame(P=100_000, n_views=5, size=512, seed=0, device="cuda"):
    """BASELINE config 2 state: a ScalarReal-like plume of P grey Gaussians in a gm_fluid model (no background set),
    positions in world units as in the first-frame stage, cameras on the 120 degree arc."""
    from .helpers.helper_gaussian import get_model
    gm = get_model("gm_fluid")(device=device)
    gm.setup_constants(H=SMOKE["H"], KNN_K=SMOKE["KNN_K"], p0=SMOKE["p0"], secs=SMOKE["secs"], k=SMOKE["k"])
    fluid = S.plume_gaussians(P, seed=seed, channels=1)
    gm._visual_xyz = torch.from_numpy(fluid["means3D"]).to(device)
    gm.prepare_visual_particles_for_rendering()  # grey 0.7, log-scale -5.9, opacity 0.1 (gm_fluid.py:1529-1536)
    cams = S.arc_cameras(n_views, size, size, device=device)
    return gm, cams


def build_ball_frame(P_fluid=350_000, P_background=150_000, hidden_dims=(22, 58, 22), n_views=8, size=512, seed=0,
                     device="cuda"):
    """BASELINE config 5 state: 500k Gaussians (fluid plume + static background) + a 28k-particle velocity field
    (the reference's max_hidden_particles), 8 views on a full ring."""
    return build_smoke_frame(P_fluid, P_background, hidden_dims, n_views, size, seed, device, ring=True)


def build_smoke_frame(P_fluid=200_000, P_background=100_000, hidden_dims=(20, 62, 20), n_views=5, size=512, seed=0,
                      device="cuda", ring=False, occluding=False):
    """BASELINE config 3 state: V visual fluid Gaussians + static background Gaussians + N hidden
    particles on a jittered unit lattice filling the plume (scaled units, < KNN_K neighbours each)."""
    rng = np.random.RandomState(seed)
    gm = GaussianModel()
    gm.setup_constants(H=SMOKE["H"], KNN_K=SMOKE["KNN_K"], p0=SMOKE["p0"], secs=SMOKE["secs"], k=SMOKE["k"])
    center = np.array([0.34, 0.0, -0.225])
    fluid = S.plume_gaussians(P_fluid, seed=seed, channels=1)
    # behind the plume, never in front (occluding=True: the round-1 layout, a cloud around the plume that hides it)
    bgd = S.backdrop_gaussians(P_background, seed=seed + 1, ring=ring, channels=3, occluding=occluding)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)  # noqa: E731
    sf = gm.scale_factor
    # visual particles live in scaled units (x100); constant attributes as gm_dynamics.py:171-173
    gm._visual_xyz = t(fluid["means3D"] * sf)
    gm._visual_color = t(fluid["colors"])
    gm._visual_scales = t(np.log(fluid["scales"]))
    gm._visual_rotation = t(fluid["rotations"])
    gm._visual_opacity = t(np.log(fluid["opacities"] / (1 - fluid["opacities"])))
    gm._gs_xyz = t(bgd["means3D"])
    gm._gs_color = t(bgd["colors"])
    gm._gs_scales = t(np.log(bgd["scales"]))
    gm._gs_rotation = t(bgd["rotations"])
    gm._gs_opacity = t(np.log(bgd["opacities"] / (1 - bgd["opacities"])))
    nx, ny, nz = hidden_dims
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1).reshape(-1, 3)
    origin = center * sf + np.array([-nx / 2.0, -2.0, -nz / 2.0])
    x_prev = g * 1.0 + rng.uniform(-0.15, 0.15, size=g.shape) + origin
    N = x_prev.shape[0]
    vel = np.tile(np.array([[0.0, 20.0, 0.0]]), (N, 1)) + rng.normal(size=(N, 3)) * 2.0  # rising smoke, units/s
    gm._xyz = t(x_prev)
    gm._velocity = t(vel)
    gm._estimate_xyz = t(x_prev + SMOKE["secs"] * vel)
    gm._imass = t(np.ones((N, 1)))
    gm._buoyancy = t(np.tile(np.array([[0.0, 1.96, 0.0]]), (N, 1)))
    gm._force = t(np.zeros((N, 3)))
    cams = (S.ring_cameras if ring else S.arc_cameras)(n_views, size, size, device=device)
    return gm, cams


# ---- developer switches (environment; all off / default in every measured configuration; DESIGN 4.7 has what each one
# measured).  None of them changes results except the two *_NOOP probes, which REPLACE a side branch's work by a sleeping
# wave and a zero gradient to show what its fork and join cost alone.
# The position stages read no screen-space gradient ("viewspace_points" feeds the background stage's densification
# only): without it the rasteriser's backward adds straight into dL/dmeans3D (geometry_only = 3).  FNX_SCREEN_GRAD=1
# keeps the reference's behaviour (the 2D-mean gradient is produced as well).
_SCREEN_GRAD = os.environ.get("FNX_SCREEN_GRAD", "0") == "1"
# Where the physics side branch is enqueued: behind the rasteriser forward (default) or right at its fork point, under
# the latency-bound head of the iteration (interpolation, preprocess, depth sort, emission)
_PHYSICS_EARLY = os.environ.get("FNX_PHYSICS_EARLY", "0") == "1"
# "hook": enqueued between the rasteriser's binning stage and its emit / blend stage, in front of the distance branch
_PHYSICS_AT_HOOK = os.environ.get("FNX_PHYSICS_EARLY", "0") == "hook"
# Where the distance-loss branch forks: "hook" = between the rasteriser's binning stage and its emit / blend stage,
# "early" = as soon as the rendered positions exist (under preprocess / sort), "loss" = behind the image loss (under the
# blend backward)
_DIST_AT = os.environ.get("FNX_DIST_AT", "hook")
# ... and how long behind that point its first kernel starts (a sleeping wave in front of it, fnx_stream_delay): the fork
# point is a kernel boundary of the main chain, the delay moves the branch under the kernel BEHIND that boundary
_DIST_DELAY_US = float(os.environ.get("FNX_DIST_DELAY_US", "0"))
# Multi-rank runs: record the RCCL all-reduce INSIDE the hipGraph (local phase | all-reduce | fused step, k iterations per
# graph) instead of replaying the local phase and issuing the collective and the step eagerly.  Opt-in: it could only be
# tried with one rank on the builder's single-GPU box (bench.py FNX_FORCE_DIST=1), and a collective that hangs inside a
# graph on a multi-GPU node cannot be caught from here; if the capture raises, the eager path is used.
# Round 5: the DEFAULT of a multi-rank run once graph_allreduce_self_test() has shown, on a process group of its own, that
# this RCCL completes a captured collective on every rank within a deadline ("auto").  FNX_GRAPH_ALLREDUCE=1 forces it,
# =0 keeps the eager collective.
_GRAPH_ALLREDUCE = os.environ.get("FNX_GRAPH_ALLREDUCE", "auto")
_GRAPH_ALLREDUCE_OK = None  # result of the self-test (None: not run)
_TEST_GRAPH_AR_FAIL = os.environ.get("FNX_TEST_GRAPH_AR_FAIL", "")  # tests only: make the in-graph capture fail


def graph_allreduce_self_test(dev, timeout_s=20.0):
    """Can this process group replay an all-reduce that was recorded into a hipGraph?  Every rank records one small
    all-reduce on a process group OF ITS OWN (a collective that hangs inside a replayed graph cannot be cancelled; it must
    not wedge the communicator the run needs), replays it and polls a completion event against a deadline -- no blocking
    wait -- then the ranks agree through the main group: True only if every rank finished in time with the right sum.
    A rank that timed out leaves a spinning collective behind on its side stream; the run goes on with the eager
    collective and says so.  Collective: every rank must call it."""
    global _GRAPH_ALLREDUCE_OK
    if not (dist.is_available() and dist.is_initialized()):
        return False
    import time
    ok = 0
    try:
        world, rank = dist.get_world_size(), dist.get_rank()
        grp = dist.new_group(ranks=list(range(world)))
        x = torch.full((1024,), float(rank + 1), device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            dist.all_reduce(x, group=grp)  # eager: sets the communicator up outside the capture
            side.synchronize()
            x.fill_(float(rank + 1))
            side.synchronize()
            _drain_collective_watchdog()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):  # (the self-test keeps torch's own context)  # see HotLoop.capture
                dist.all_reduce(x, group=grp)
            g.replay()
            done = torch.cuda.Event()
            done.record(side)
        deadline = time.monotonic() + float(timeout_s)
        while not done.query() and time.monotonic() < deadline:
            time.sleep(0.005)
        if done.query():
            ok = int(abs(float(x[0].item()) - world * (world + 1) / 2.0) < 1e-3)
    except Exception as e:  # e.g. a backend that cannot be captured (gloo)
        print(f"[harness] all-reduce self-test inside a hipGraph: {type(e).__name__}: {e}", flush=True)
        ok = 0
    flag = torch.tensor([ok], device=dev, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # the main group, eagerly: every rank learns the common verdict
    _GRAPH_ALLREDUCE_OK = bool(int(flag.item()))
    return _GRAPH_ALLREDUCE_OK


def _drain_collective_watchdog(seconds=0.35):
    """Call before a capture that records a collective.  The process group's watchdog thread polls the completion events
    of earlier EAGER collectives every 100 ms until they have completed; recording a collective pulls the group's
    internal stream into the capture, and on this HIP an event query on ANY event of a stream that is capturing fails
    (hipErrorCapturedEvent) -- the watchdog thread dies with it and takes the process down (3 of 24 runs).  With the
    device idle, one polling period retires every outstanding work and the watchdog has nothing left to ask."""
    import time
    torch.cuda.synchronize()
    time.sleep(float(os.environ.get("FNX_WATCHDOG_DRAIN_S", seconds)))


class _graph_capture:
    """`with torch.cuda.graph(g, stream=...)` without its `torch.cuda.empty_cache()`: handing every cached block back to the
    driver in front of EVERY capture costs ~4 ms by itself and turns the allocations of the iterations that follow into
    hipMalloc calls -- per frame of a sequence (bench.py --frames), where the loop is re-captured 120 times."""

    def __init__(self, g, stream, pool=None, capture_error_mode="global"):
        self.g, self.pool, self.mode = g, pool, capture_error_mode
        self.stream_ctx = torch.cuda.stream(stream)

    def __enter__(self):
        torch.cuda.synchronize()
        self.stream_ctx.__enter__()
        if self.pool is None:
            self.g.capture_begin(capture_error_mode=self.mode)
        else:
            self.g.capture_begin(self.pool, capture_error_mode=self.mode)

    def __exit__(self, *args):
        self.g.capture_end()
        self.stream_ctx.__exit__(*args)


def _leak(obj):
    """One reference that is never given back.  For a torch.cuda.CUDAGraph whose capture was invalidated: its destructor
    raises a c10::Error on this torch (-> std::terminate), at the rebinding of the name or at interpreter exit."""
    import ctypes
    ctypes.pythonapi.Py_IncRef(ctypes.py_object(obj))


def _graph_allreduce_wanted():
    if _GRAPH_ALLREDUCE == "1":
        return True
    if _GRAPH_ALLREDUCE == "0":
        return False
    return bool(_GRAPH_ALLREDUCE_OK)  # auto: only after a passed self-test
# One fork point for both side branches (the rasteriser's between-stages hook) instead of two: every fork / join of the
# captured graph costs the main chain ~5 us in front of the kernel behind it
_SINGLE_FORK = os.environ.get("FNX_SINGLE_FORK", "0") == "1"
# The physical stage's targets as their grey means, formed once (FNX_GT_GREY=0: three planes, averaged per pixel and iteration)
_GT_GREY = os.environ.get("FNX_GT_GREY", "1") == "1"
# BASELINE config 5 (both rasterisers per view): the 1-channel fluid image out of the 3-channel rasteriser's own pass
# (rasterizer dual mode, include/fnx_raster.h fnx_raster_dual_t) instead of a second render.  FNX_DUAL_FUSED=0: two renders.
_DUAL_FUSED = os.environ.get("FNX_DUAL_FUSED", "1") != "0"
_DIST_NOOP = os.environ.get("FNX_DIST_NOOP") == "1"  # probe: the distance branch without its kernels (WRONG gradients)
_PHYS_NOOP = os.environ.get("FNX_PHYS_NOOP") == "1"  # probe: the physics branch without its kernels (WRONG gradients)


class HotLoop:
    """One frame's optimisation loop over `gm` and `cams` (physical-particle stage)."""

    def __init__(self, gm, cams, rd_pipe="render_dynamics", rank=0, world=1, log_scalars=False, cfg=SMOKE,
                 physics_per_view=True, shared_terms_rank=None, image_loss="torch", fused_physics=False, defer_visual_backward=False,
                 force_all_reduce=False, capturable=False, parallel_views=False, batched_views=False,
                 fused_step=False, dual_channel=False, reuse_streams_of=None):
        self.gm, self.cams, self.rank, self.world, self.cfg = gm, cams, rank, world, dict(cfg)
        self.view_subset = None
        self.render_func, self.GRsetting, self.GRzer = get_render_pipe(rd_pipe)
        # BASELINE config 5: every view is also rendered by the 1-channel rasteriser (fluid only, render_fluid) and
        # compared with a 1-channel target; both image terms drive the same particle positions
        self.dual_channel = bool(dual_channel)
        if self.dual_channel:
            assert batched_views, "dual_channel is implemented for the view-batched loop"
            _, self.GRsetting1, self.GRzer1 = get_render_pipe("render_fluid")
        self.log_scalars = log_scalars
        self.physics_per_view = physics_per_view
        # Multi-rank runs: the terms that do not depend on the view (physics terms, distance loss) are the same on every
        # rank.  None: every rank evaluates them and adds them once per LOCAL view (the reference's per-view evaluation,
        # tpp:365-404, rank by rank).  r: only rank r evaluates them and adds them `batch` times; the all-reduce hands the
        # sum to everybody.  bench.py --shared-terms last-rank picks the LAST rank: round-robin sharding gives
        # it the fewest views, so the extra work lands on the rank that would otherwise wait (view-batched loop only).
        # "spread" (round 6): the three terms go to DIFFERENT ranks (spread_owners: the gas term, the next-gas + exyz terms,
        # the distance loss -- each to the rank with the least work so far), each added `batch` times by its owner; the same
        # all-reduce sums them, no new collective.  One rank no longer carries all of the ~300 us of side-stream kernels.
        self.shared_terms_rank = shared_terms_rank
        if shared_terms_rank is not None and not batched_views:
            raise ValueError("shared_terms_rank is implemented by the view-batched loop only (batched_views=True): the serial / "
                             "parallel loops evaluate the view-independent terms per view or on rank 0")
        self.emulated = None  # (rank, world) of the run whose share `view_subset` is (bench.py --emulate-world)
        self.image_loss = image_loss
        self.fused_physics = fused_physics
        if getattr(gm, "knn_cap", False) and (fused_physics or defer_visual_backward):
            raise ValueError("gm.knn_cap (max_num_neighbors mode) runs through the per-term physics methods: "
                             "HotLoop(fused_physics=False, defer_visual_backward=False)")
        self.force_all_reduce = force_all_reduce
        # the fused kernels take every pair within H: have them flag lists longer than the reference's cap
        if (fused_physics or defer_visual_backward) and not getattr(gm, "knn_cap", False) and \
                os.environ.get("FNX_KNN_WATCH", "1") != "0" and gm._xyz.is_cuda:
            gm.arm_knn_watch()
        gm.defer_visual_backward = bool(defer_visual_backward)
        dev = gm._xyz.device
        self.background = torch.zeros(3, device=dev)
        self.optim_args = SimpleNamespace(**{k: cfg[k] for k in ("position_lr_init", "position_lr_final",
                                                               "position_lr_delay_mult", "position_lr_max_steps")})
        self.capturable = capturable
        self.parallel_views = bool(parallel_views)
        if self.parallel_views:
            assert fused_physics and defer_visual_backward, "parallel_views needs the fused physics node and the deferred visual backward"
        self.batched_views = bool(batched_views)
        if self.batched_views:
            assert fused_physics and defer_visual_backward and image_loss == "fused" and rd_pipe == "render_dynamics", \
                "batched_views needs render_dynamics, the fused image loss / physics node and the deferred visual backward"
        # one visual forward per iteration, consumed once: hand the memoised tensor out without a copy
        gm.share_visual_output = self.batched_views
        self._gt_cache = None
        self.graph_iterations = 1
        self.side_stream = None
        self.ch1_stream = None
        self.fused_step = bool(fused_step)  # gradient mean + Adam as one kernel (batched_views only)
        if self.fused_step:
            assert self.batched_views and capturable, "fused_step needs batched_views and capturable=True"
        self.view_streams = []
        # View groups (round 6, FNX_VIEW_GROUPS="3,2" or the attribute): the local views' render -> image loss -> backward
        # chains run as that many INDEPENDENT view batches on streams of their own (parallel branches of the captured graph)
        # instead of one launch sequence over all views: the tail of one group's blend kernels -- a few deep tiles on
        # otherwise idle compute units -- lies under the other group's kernels.  None: one batch (the default).
        vg = os.environ.get("FNX_VIEW_GROUPS", "")
        self.view_groups = [int(x) for x in vg.replace("+", ",").split(",") if x.strip()] if vg else None
        self.group_streams = []
        gm.training_setup_current(self.optim_args, capturable=capturable)
        self.itr = 0
        self.last = {}
        self.graph = None
        self.graph_finish = None
        self._reduce_buf = None
        self._replay = False
        # Graph mode keeps every iteration (eager or captured) on one dedicated stream: autograd's gradient
        # accumulation is bound to the stream a leaf was first used on, and a leaf first used on the
        # default stream would drag that stream into the capture.
        self.stream = torch.cuda.Stream(device=dev) if capturable else None
        # `reuse_streams_of`: a loop this one replaces (frame after frame of a sequence).  The caching allocator hands a freed
        # block only to requests on the stream it was allocated on: a new loop on NEW streams finds none of its predecessor's
        # blocks usable and takes ~1.2 GiB from the driver per frame of config 3 (28 hipMallocs, 70 GiB reserved after 60
        # frames; a hipMalloc costs 0.1 ms on a good day and 1.3 ms on a bad one: set-up 10 -> 45 ms).
        if reuse_streams_of is not None and capturable and getattr(reuse_streams_of, "stream", None) is not None:
            o = reuse_streams_of
            self.stream, self.side_stream, self.ch1_stream = o.stream, o.side_stream, o.ch1_stream
            self.dist_stream = getattr(o, "dist_stream", None)
            self.view_streams, self.group_streams = list(o.view_streams), list(o.group_streams)
            # ... and its background tensor: the camera batch (renderer.pipes._view_batch) is cached per (cameras, background
            # tensor), and with it the static bins, the depth hints and the sort states -- a new tensor per frame made every
            # frame start without depth hints and left 64 frames' worth of static bins (72 MB each) in the caches
            if getattr(o, "background", None) is not None and o.background.device == dev:
                self.background = o.background

    def _group_lists(self, mine):
        """The local views cut into `view_groups` consecutive groups (sizes; a remainder joins the last group)."""
        if not self.view_groups or len(mine) < 2:
            return [mine]
        out, at = [], 0
        for n in self.view_groups:
            if at >= len(mine):
                break
            out.append(mine[at:at + n])
            at += n
        if at < len(mine):
            out[-1] = out[-1] + mine[at:]
        return out

    def _mine(self, batch):
        """Views of the batch this rank renders: its round-robin share, or an explicit `view_subset` (single-process
        emulation of one rank's share of a larger run: the batch mean still divides by the whole batch)."""
        return list(self.view_subset) if self.view_subset is not None else shard_views(batch, self.rank, self.world)

    @torch.no_grad()
    def make_targets(self, shift=0.3):
        """Synthetic ground truth: the scene rendered with the hidden particles displaced by `shift`."""
        if self.stream is not None and torch.cuda.current_stream() != self.stream:
            with torch.cuda.stream(self.stream):
                self.make_targets(shift)
            torch.cuda.current_stream().wait_stream(self.stream)
            return
        gm = self.gm
        keep = gm._estimate_xyz_nn.data.clone()
        gm._estimate_xyz_nn.data += shift / gm.scale_factor
        for cam in self.cams:
            pkg = self.render_func(cam, gm, None, self.background, GRsetting=self.GRsetting, GRzer=self.GRzer,
                                   pos_type="guess_visual_nn", scale=True)
            cam.original_image = pkg["render"].detach().clamp(0, 1).clone()
            if self.dual_channel:
                from .renderer.pipes import render_fluid
                pkg1 = render_fluid(cam, gm, None, self.background, GRsetting=self.GRsetting1, GRzer=self.GRzer1,
                                    pos_type="guess_visual_nn", scale=True)
                cam.original_image_ch1 = pkg1["render"].detach().clamp(0, 1).clone()
        gm._estimate_xyz_nn.data.copy_(keep)

    def _image_loss(self, image, gt_image):
        c = self.cfg
        if self.image_loss == "fused":
            from .losses import fused_l1_dssim_grey
            l1_value, ssim_value = fused_l1_dssim_grey(image, gt_image)  # grey-mean fused in
        else:
            # grey-mean both images and replicate to 3 channels (tpp:356-360)
            gt = torch.cat([torch.mean(gt_image, dim=0, keepdim=True)] * 3, dim=0)
            im = torch.cat([torch.mean(image, dim=0, keepdim=True)] * 3, dim=0)
            l1_value = l1_loss(im, gt)
            ssim_value = 1.0 - ssim(im, gt)
        return ((1.0 - c["lambda_dssim"]) * l1_value * c["lambda_image"]
                + c["lambda_dssim"] * ssim_value * c["lambda_image"]), l1_value, ssim_value

    def _physics_loss(self):
        gm, c = self.gm, self.cfg
        if self.fused_physics:
            from .physics import physical_stage_loss
            return physical_stage_loss(gm, c["lambda_exyz"], c["lambda_gas_constraints"],
                                       c["lambda_next_gas_constraints"], gm.state_memo("physics_loss"))
        loss = 0.0
        if c["lambda_exyz"] > 0:
            loss = loss + c["lambda_exyz"] * l2_loss(gm._estimate_xyz_nn * gm.scale_factor, gm._estimate_xyz)
        if c["lambda_gas_constraints"] > 0:
            pr = gm.get_gas_constraints_from_exyz_nn()
            loss = loss + c["lambda_gas_constraints"] * l2_loss(pr, torch.ones_like(pr))
        if c["lambda_next_gas_constraints"] > 0:
            pn = gm.get_gas_constraints_from_vel_nn_guess()
            loss = loss + c["lambda_next_gas_constraints"] * l2_loss(pn, torch.ones_like(pn))
        return loss

    def _fresh_streams(self):
        """New streams for everything the loop launches on (after a failed capture, see capture())."""
        dev = self.gm._xyz.device
        if self.stream is not None:
            self.stream = torch.cuda.Stream(device=dev)
        for name in ("side_stream", "dist_stream", "ch1_stream"):
            if getattr(self, name, None) is not None:
                setattr(self, name, torch.cuda.Stream(device=dev))
        if getattr(self, "view_streams", None):
            self.view_streams = [torch.cuda.Stream(device=dev) for _ in self.view_streams]

    def capture(self, warmup=3, iterations=1, pool=None):
        """Record one whole iteration (all views forward + losses + backward + gradient mean + Adam) as a
        hipGraph; later iteration() calls replay it.  Needs the sync-free rasteriser mode with a seeded
        binning capacity (run a few eager iterations first) and capturable=True.  The launch sequence is
        frozen: the binning capacity and every cache decision are those of the recorded iteration;
        rasterizer.check_status() after replays still reports a capacity overflow.
        `pool`: a `torch.cuda.graph_pool_handle()` the capture allocates from instead of a private pool of its own.  A
        caller that re-captures frame after frame (bench.py --frames) passes ONE handle for all its captures: a destroyed
        graph's private pool is only handed back by `torch.cuda.empty_cache()` -- which this capture does not call, see
        _graph_capture -- so private pools pile up at ~1.1 GiB per frame of config 3 (28 segments from the driver per
        capture, 70 GiB reserved after 60 frames, and late frames' captures slow down to 20 ms); in a shared pool the
        previous frame's blocks are free again when its graph has gone.
        `iterations` > 1 records that many consecutive iterations in one graph (one launch gap per replay instead
        of one per iteration; the learning rate is constant in this stage, gm.update_learning_rate_current): every
        iteration() call then advances the optimisation by `iterations` steps (self.iterations_per_call)."""
        assert self.capturable, "HotLoop(capturable=True) is required for graph capture"
        from . import rasterizer
        assert not rasterizer._HOST_SYNC, "graph capture needs rasterizer.set_host_sync(False)"
        for _ in range(warmup):
            self.iteration()
        torch.cuda.synchronize()
        self.gm.invalidate_caches()
        # The fused step leaves the NEXT iteration's hidden-particle grid (and scaled positions) behind.  The recorded
        # graph's first iteration may take them over instead of rebuilding the grid with five launches of its own (~28 us
        # per replay, round 5): on every replay they are what the previous replay's last iteration -- or, the first time,
        # the warm-up's -- left.  Particles moved between replays by anything else need a new capture (or use_graph(False)).
        if self.fused_step and os.environ.get("FNX_CAPTURE_KEEP_GRID", "1") != "0":
            gm, est = self.gm, self.gm._estimate_xyz_nn
            key = (id(est), est._version)
            grid, scaled = getattr(gm, "_step_grid", None), getattr(gm, "_est_scaled", None)
            if grid is not None and scaled is not None and scaled[0] == key and grid.N == est.shape[0]:
                gm._grid_cache["est"] = (key, grid)
        rasterizer._pending_status.clear()
        g = torch.cuda.CUDAGraph()
        self.graph_finish = None
        if self.batched_views and (self.world > 1 or self.force_all_reduce):
            # Multi-GPU: the RCCL all-reduce stays OUTSIDE the captured graphs (local gradient | all-reduce |
            # batch mean + optimiser step) and works on a buffer allocated OUTSIDE the graphs' memory pool:
            # RCCL touching pool memory (as a host read does) makes later replays fault with
            # "write access to a read-only page".  A 0.3 MB all-reduce costs one launch either way.
            self._reduce_buf = torch.zeros_like(self.gm._estimate_xyz_nn.detach())
            torch.cuda.synchronize()
            if _graph_allreduce_wanted() and self.fused_step:
                try:
                    itr0, tot0 = self.itr, self.gm.total_iterations
                    _drain_collective_watchdog()
                    # thread_local: calls of other threads (the process group's watchdog, the allocator) do not
                    # invalidate this capture
                    with _graph_capture(g, self.stream, pool=pool, capture_error_mode="thread_local"):
                        for k in range(int(iterations)):
                            self._iteration_body_batched(phase="local")
                            dist.all_reduce(self._reduce_buf, op=dist.ReduceOp.SUM)
                            self._finish_step(len(self.cams), grad=self._reduce_buf)
                            if k == 0 and _TEST_GRAPH_AR_FAIL == "raise":  # tests: the fallback below
                                raise RuntimeError("injected failure inside the capture")
                            if k == 0 and _TEST_GRAPH_AR_FAIL == "sync":  # tests: a call that invalidates the capture
                                torch.cuda.synchronize()
                    self.itr, self.gm.total_iterations = itr0, tot0
                    self.graph, self.graph_iterations, self._replay, self.graph_finish = g, int(iterations), True, None
                    return g
                except Exception as e:  # the collective refused to be captured: the eager path below
                    print(f"[harness] all-reduce inside the graph failed ({type(e).__name__}: {e}); eager collective", flush=True)
                    self.itr, self.gm.total_iterations = itr0, tot0
                    # a graph object whose capture was invalidated must never be destroyed: its destructor raises (this
                    # torch: c10 check inside ~CUDAGraph -> std::terminate).  One leaked reference keeps it alive to the end.
                    _leak(g)
                    g = torch.cuda.CUDAGraph()
                    try:
                        torch.cuda.synchronize()
                    except Exception:  # the invalidated capture's error, reported once more
                        torch.cuda.synchronize()
                    # ... and on this HIP the streams that took part keep the status "invalidated" behind the failed
                    # hipStreamEndCapture: the loop goes on with streams of its own making
                    self._fresh_streams()
                    self.gm.invalidate_caches()
                    rasterizer._pending_status.clear()
            try:
                with _graph_capture(g, self.stream, pool=pool):
                    self._iteration_body_batched(phase="local")
            except Exception:
                _leak(g)
                self._fresh_streams()
                raise
            if self.fused_step:
                # batch mean + Adam step is ONE kernel (fnx_adam_step): launched eagerly behind the all-reduce, it
                # costs a kernel launch instead of the ~27 us start-up gap of a second graph
                self.graph_finish = "eager"
            else:
                g2 = torch.cuda.CUDAGraph()
                with _graph_capture(g2, self.stream, pool=g.pool()):
                    self._finish_step(len(self.cams), grad=self._reduce_buf)
                self.graph_finish = g2
            iterations = 1
        else:
            itr0, tot0 = self.itr, self.gm.total_iterations
            with _graph_capture(g, self.stream, pool=pool):
                for _ in range(int(iterations)):
                    self._iteration_body()
            self.itr, self.gm.total_iterations = itr0, tot0  # recording is not running
        self.graph = g
        self.graph_iterations = int(iterations)
        self._replay = True
        return g

    def release_graph(self):
        """Drop the captured graph(s) NOW (and with them the memory their captures allocated), without waiting for this
        object to be collected: a loop that is replaced frame after frame may sit in a reference cycle until the cyclic
        collector runs, and its graph's pool (~1.1 GiB for config 3) with it."""
        self.graph = None
        self.graph_finish = None
        self._replay = False

    @property
    def iterations_per_call(self):
        """Optimisation steps one iteration() call performs (> 1 only while replaying a multi-iteration graph)."""
        return self.graph_iterations if (self.graph is not None and self._replay) else 1

    def use_graph(self, enabled: bool):
        """Switch between replaying the captured graph and eager launches.  The graph (and the memory
        pool behind every tensor allocated while capturing) stays alive; state derived from the
        particle positions is forgotten because replays do not bump tensor version counters."""
        self._replay = bool(enabled) and self.graph is not None
        self.gm.invalidate_caches()

    def iteration(self):
        if self.stream is None:
            return self._iteration_body()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            if self.graph is not None and self._replay:
                self.itr += self.graph_iterations
                self.gm.total_iterations += self.graph_iterations
                self.graph.replay()
                if self.graph_finish is not None:
                    dist.all_reduce(self._reduce_buf, op=dist.ReduceOp.SUM)
                    if self.graph_finish == "eager":
                        self._finish_step(len(self.cams), grad=self._reduce_buf)
                    else:
                        self.graph_finish.replay()
            else:
                self._iteration_body()
        torch.cuda.current_stream().wait_stream(self.stream)

    def _iteration_body_parallel(self):
        """Same iteration with the views of the batch on separate streams (fork after the work every
        view shares, join before the gradient mean).  Inside a captured graph the views become parallel
        branches, so the tail of one view's blend kernels overlaps the next view's kernels.  The physics
        gradient is identical for every view: it is evaluated once and added once per local view."""
        gm = self.gm
        self.itr += 1
        gm.total_iterations += 1
        gm.update_learning_rate_current(self.itr)
        gm.zero_gradient_cache_current()
        batch = len(self.cams)
        mine = self._mine(batch)
        main = torch.cuda.current_stream()
        while len(self.view_streams) < len(mine):
            self.view_streams.append(torch.cuda.Stream(device=gm._xyz.device))
        with torch.no_grad():
            gm.get_visual_xyz_from_nn()  # fills the per-state memo (forward kernel runs once, here)
        if self.physics_per_view or self.rank == 0:
            gp, = torch.autograd.grad(self._physics_loss(), gm._estimate_xyz_nn)
            n_phys = len(mine) if self.physics_per_view else batch
        else:
            gp, n_phys = None, 0
        fork = torch.cuda.Event()
        fork.record(main)
        for k, v in enumerate(mine):
            s = self.view_streams[k]
            s.wait_event(fork)
            with torch.cuda.stream(s):
                cam = self.cams[v]
                pkg = self.render_func(cam, gm, None, self.background, GRsetting=self.GRsetting, GRzer=self.GRzer,
                                       pos_type="guess_visual_nn", scale=True)
                loss, _, _ = self._image_loss(pkg["render"], cam.original_image)
                if self.cfg.get("lambda_current_distance", 0.0) > 0:  # tpp:365-366
                    from .utils.loss_utils import distance_loss
                    loss = loss + self.cfg["lambda_current_distance"] * distance_loss(
                        pkg["render_xyz"], self.cfg["distance_threshold_visual"])
                torch.autograd.grad(loss, [gm._estimate_xyz_nn], allow_unused=True)  # -> deferred visual backward
            main.wait_stream(s)
        if gp is not None:
            gm._estimate_xyz_nn_grad += gp * float(n_phys)
        gm.flush_deferred_gradients()
        if self.world > 1 or self.force_all_reduce:
            dist.all_reduce(gm._estimate_xyz_nn_grad, op=dist.ReduceOp.SUM)
        gm.set_batch_gradient_current(batch)
        gm.optimizer.step()
        gm.optimizer.zero_grad()

    def _gt_stack(self, mine, attr="original_image", grey_mean=False):
        """Ground-truth images of this rank's views as one [V,C,H,W] tensor (stacked once; the entry keeps the
        images it was built from).  grey_mean: their grey means [V,1,H,W] instead (losses.grey_mean_target: what the
        physical stage compares, tpp:356-360; formed once per frame instead of per pixel and iteration)."""
        if self._gt_cache is None:
            self._gt_cache = {}
        imgs = [getattr(self.cams[v], attr) for v in mine]
        key = (attr, bool(grey_mean), tuple(mine))
        hit = self._gt_cache.get(key)
        if hit is None or len(hit[0]) != len(imgs) or any(a is not b for a, b in zip(hit[0], imgs)):
            stack = torch.stack(imgs).contiguous()
            if grey_mean:
                from .losses import grey_mean_target
                stack = grey_mean_target(stack)
            hit = self._gt_cache[key] = (imgs, stack)
        return hit[1]

    def _iteration_body_batched(self, phase="all"):
        """Same iteration with this rank's views rendered, compared and back-propagated by ONE launch
        sequence (rasteriser / loss kernels take the view as a grid dimension).  Mathematically the sum
        over the views of the per-view losses of `_iteration_body`; the physics gradient, identical for
        every view, is evaluated once and added once per local view."""
        from .losses import image_loss_value_and_grad
        from .renderer.pipes import render_dynamics_views
        gm, c = self.gm, self.cfg
        self.itr += 1
        gm.total_iterations += 1
        gm.update_learning_rate_current(self.itr)
        gm.zero_gradient_cache_current()
        batch = len(self.cams)
        mine = self._mine(batch)
        main = torch.cuda.current_stream()
        # hidden-particle grid + the one visual forward of this iteration (memoised), then the leaf of the rasteriser
        # positions [advected visual / scale_factor | background].  The main chain's next kernel is enqueued BEFORE the
        # side branches fork: a captured graph keeps the first successor of a node on the node's hardware queue, and a
        # hop of the critical path between queues costs ~10 us (rocprof timeline), a hop of a side branch nothing.
        if mine:  # the interpolation kernel writes the fluid rows of the leaf itself (positions / scale_factor)
            means3D = gm.render_means_from_nn()
        else:
            means3D = None
            with torch.no_grad():
                gm.get_visual_xyz_from_nn()
        # Side branches of the iteration.  Their fork POINTS are recorded where their inputs are ready, but their
        # kernels are enqueued behind the rasteriser forward: a captured graph keeps the first-enqueued successor of a
        # node on the node's hardware queue, and the critical path must not be the one that hops.
        #   physics terms (depend on the particle state only: ~12 small kernels under preprocess + sort + emit)
        #   distance_loss(render_xyz) (tpp:365-366; the same for every view: evaluated once by the radius-limited
        #     kernel, added once per local view).  It forks between the rasteriser's binning stage and its emit /
        #     blend stage: next to the throughput-bound blend kernels its 200k-point grid build costs its own few
        #     microseconds of work, next to the latency-bound depth sort it stretched the critical path by 100 us.
        if self.side_stream is None:
            self.side_stream = torch.cuda.Stream(device=gm._xyz.device)
            self.dist_stream = torch.cuda.Stream(device=gm._xyz.device)
        gp, n_phys, gd = None, 0, None
        fork = torch.cuda.Event()
        single_fork = _SINGLE_FORK and bool(mine) and not self.dual_channel and not _PHYSICS_EARLY
        if not single_fork:
            fork.record(main)
        # who evaluates the view-independent terms, and how many times their gradient counts (see __init__)
        shared = self.shared_terms_rank if self.shared_terms_rank is not None else (None if self.physics_per_view else 0)
        erank, eworld = self.emulated or (self.rank, self.world)
        lam_phys = (c["lambda_exyz"], c["lambda_gas_constraints"], c["lambda_next_gas_constraints"])
        if shared == "spread":
            own = spread_owners(batch, eworld)
            lam_phys = (lam_phys[0] if own["next"] == erank else 0.0, lam_phys[1] if own["gas"] == erank else 0.0,
                        lam_phys[2] if own["next"] == erank else 0.0)
            phys_here = any(l > 0 for l in lam_phys)
            n_phys_weight = batch
            dist_shared, dist_here, n_dist_weight = True, own["distance"] == erank, batch
        else:
            phys_here = shared is None or shared == erank
            n_phys_weight = len(mine) if shared is None else batch
            # the distance term needs a render on the rank that evaluates it (its gradient joins the rendered positions')
            dist_shared = shared is not None and len(shard_views(batch, shared, eworld)) > 0
            dist_here = (shared == erank) if dist_shared else True
            n_dist_weight = batch if dist_shared else len(mine)
        use_dist = bool(mine) and dist_here and c.get("lambda_current_distance", 0.0) > 0
        if use_dist:
            from .physics import prefer_distance_lists
            prefer_distance_lists(len(mine) >= 3)  # see there: lists beside a long blend forward, the grid beside a short one
        physics_launched = []

        def launch_physics():
            nonlocal gp, n_phys
            if not phys_here or physics_launched:  # once per iteration, however often the hook fires
                return
            physics_launched.append(1)
            if single_fork and not forked_d:  # the forward did not pass the hook: behind everything enqueued so far
                self.side_stream.wait_stream(main)
            else:
                self.side_stream.wait_event(fork_d if single_fork else fork)
            with torch.cuda.stream(self.side_stream):
                # work items of the hidden-particle grid for the cell-by-cell hidden<-visual backward at the end
                # of the iteration: two tiny kernels, off the critical path on this branch
                vmemo = gm._visual_memo[1]
                if "hgrid" in vmemo:
                    vmemo["hitems"] = vmemo["hgrid"].cell_items(refresh=True)
                if self.fused_physics and _PHYS_NOOP:  # developer probe: fork / join without the work
                    from . import _physics_lib as _PL
                    if getattr(self, "_gp_zero", None) is None:
                        self._gp_zero = torch.zeros_like(gm._estimate_xyz_nn)
                    _PL.check(_PL.physics().fnx_stream_delay(1.0, _lib_raw_stream()))
                    gp = self._gp_zero
                elif self.fused_physics:  # value and gradient straight from the fused stage (no autograd node)
                    from .physics import physical_stage_value_and_grad
                    _, gp = physical_stage_value_and_grad(gm, lam_phys[0], lam_phys[1], lam_phys[2],
                                                          gm.state_memo("physics_loss"))
                else:
                    gp, = torch.autograd.grad(self._physics_loss(), gm._estimate_xyz_nn)
            n_phys = n_phys_weight

        if _PHYSICS_EARLY:
            launch_physics()
        if mine:
            from . import rasterizer
            fork_d, forked_d = torch.cuda.Event(), []
            if use_dist and _DIST_AT == "early":
                fork_d.record(main)
                forked_d.append(1)
            if (use_dist and _DIST_AT == "hook") or _PHYSICS_AT_HOOK or single_fork:
                def _hook():
                    if _PHYSICS_AT_HOOK:
                        launch_physics()
                    if (use_dist and _DIST_AT == "hook") or single_fork:
                        fork_d.record(torch.cuda.current_stream())
                        forked_d.append(1)
                rasterizer.set_between_stages_hook(_hook)
            try:
                from .renderer import pipes as _pipes
                dual_fused = (self.dual_channel and _DUAL_FUSED and not _SCREEN_GRAD and _pipes._STATIC_SPLIT
                              and gm.get_gs_xyz.shape[0] > 0)
                # view groups: the first group takes the place of the one batch; the others follow on their own streams below
                groups = self._group_lists(mine) if not self.dual_channel else [mine]
                first = groups[0]
                if len(groups) > 1:
                    fork_g = torch.cuda.Event()
                    fork_g.record(main)
                pkg = render_dynamics_views([self.cams[v] for v in first], gm, None, self.background,
                                            GRsetting=self.GRsetting, GRzer=self.GRzer, pos_type="guess_visual_nn",
                                            scale=True, means3D=means3D, screen_grad=_SCREEN_GRAD,
                                            dual_bg=self.background[:1] if dual_fused else None)
            finally:
                rasterizer.set_between_stages_hook(None)
        if not _PHYSICS_EARLY and not (_PHYSICS_AT_HOOK and mine) and not (single_fork and use_dist):
            launch_physics()
        if mine:
            dimg_ready = None
            if use_dist and _DIST_AT == "loss":  # the image term first: the distance branch forks behind it
                loss, per_view, dimg = image_loss_value_and_grad(pkg["render"].detach(), self._gt_stack(first, grey_mean=_GT_GREY), c["lambda_dssim"],
                                                                 c["lambda_image"])
                dimg_ready = (loss, per_view, dimg)
                fork_d.record(main)
                forked_d.append(1)
            if use_dist:
                from .physics import distance_loss_value_and_grad
                if forked_d:
                    self.dist_stream.wait_event(fork_d)
                else:  # the forward did not pass the hook: order the branch behind everything enqueued so far
                    self.dist_stream.wait_stream(main)
                with torch.cuda.stream(self.dist_stream):
                    if _DIST_DELAY_US > 0:
                        from . import _physics_lib as _PL
                        _PL.check(_PL.physics().fnx_stream_delay(_DIST_DELAY_US, _lib_raw_stream()))
                    n_vis = gm._visual_xyz.shape[0]
                    if _DIST_NOOP:  # developer probe: the branch's fork / join without its work
                        from . import _physics_lib as _PL
                        if getattr(self, "_gd_zero", None) is None:
                            self._gd_zero = torch.zeros(n_vis, 3, device=means3D.device)
                        _PL.check(_PL.physics().fnx_stream_delay(1.0, _lib_raw_stream()))
                        self.last_distance, gd = None, self._gd_zero
                    else:
                        self.last_distance, gd = distance_loss_value_and_grad(means3D.detach()[:n_vis],
                                                                              c["distance_threshold_visual"])
            if single_fork and use_dist:  # behind the distance kernels, the order the two branches run in anyway
                launch_physics()
            # image term and its gradient with respect to the rendered batch (no autograd node for the loss:
            # that saves the ones seed and its copies), then back through the rasteriser
            if self.dual_channel and dual_fused:
                fork_img = torch.cuda.Event()
                fork_img.record(main)  # both images exist: the second one's image term forks here
            if dimg_ready is not None:
                loss, per_view, dimg = dimg_ready
            else:
                loss, per_view, dimg = image_loss_value_and_grad(pkg["render"].detach(), self._gt_stack(first, grey_mean=_GT_GREY), c["lambda_dssim"],
                                                                 c["lambda_image"])
            if self.log_scalars:
                self.last = dict(l1=per_view[-1, 0].item(), ssim=1.0 - per_view[-1, 1].item(), total=loss.item())
            outs, seeds = [pkg["render"]], [dimg]
            if self.dual_channel and dual_fused:
                # the 1-channel image of the fluid came out of the same pass (dual mode); its image term joins the backward
                # (on its own stream: the two image terms are independent, ~40 us each at 8 views)
                if self.ch1_stream is None:
                    self.ch1_stream = torch.cuda.Stream(device=gm._xyz.device)
                side = os.environ.get("FNX_DUAL_LOSS_STREAM", "0") == "1"
                if side:
                    self.ch1_stream.wait_event(fork_img)
                with torch.cuda.stream(self.ch1_stream if side else main):
                    _, _, dimg1 = image_loss_value_and_grad(pkg["render1"].detach(), self._gt_stack(mine, "original_image_ch1"),
                                                            c["lambda_dssim"], c["lambda_image"], grey=False)
                if side:
                    main.wait_stream(self.ch1_stream)
                outs.append(pkg["render1"])
                seeds.append(dimg1)
            elif self.dual_channel:
                # the 1-channel render of the fluid (config 5) and its image term on their own stream, next to the
                # 3-channel render: with few views per rank both blend forwards are bound by their deepest tiles' walks,
                # not by throughput, and overlap almost entirely; autograd runs each backward on its forward's stream
                from .renderer.pipes import render_fluid_views
                if self.ch1_stream is None:
                    self.ch1_stream = torch.cuda.Stream(device=gm._xyz.device)
                n_fluid = means3D.shape[0] - gm.get_gs_xyz.shape[0]
                self.ch1_stream.wait_event(fork)
                with torch.cuda.stream(self.ch1_stream):
                    pkg1 = render_fluid_views([self.cams[v] for v in mine], gm, None, self.background,
                                              GRsetting=self.GRsetting1, GRzer=self.GRzer1, pos_type="guess_visual_nn",
                                              scale=True, means3D=means3D[:n_fluid], screen_grad=_SCREEN_GRAD)
                    _, _, dimg1 = image_loss_value_and_grad(pkg1["render"].detach(), self._gt_stack(mine, "original_image_ch1"),
                                                            c["lambda_dssim"], c["lambda_image"], grey=False)
                main.wait_stream(self.ch1_stream)
                outs.append(pkg1["render"])
                seeds.append(dimg1)
            g_means, = torch.autograd.grad(outs, [means3D], grad_outputs=seeds)
            if len(groups) > 1:
                # the other groups' chains (render -> image term -> backward), each a view batch of its own (camera tensors,
                # static bins, sort states and depth hints are kept per camera set) on a stream of its own: parallel branches
                # of the captured graph from the point where the rendered positions exist to the sum of the gradients
                while len(self.group_streams) < len(groups) - 1:
                    self.group_streams.append(torch.cuda.Stream(device=gm._xyz.device))
                for gi, grp in enumerate(groups[1:]):
                    gs = self.group_streams[gi]
                    gs.wait_event(fork_g)
                    with torch.cuda.stream(gs):
                        pkg2 = render_dynamics_views([self.cams[v] for v in grp], gm, None, self.background,
                                                     GRsetting=self.GRsetting, GRzer=self.GRzer, pos_type="guess_visual_nn",
                                                     scale=True, means3D=means3D, screen_grad=_SCREEN_GRAD)
                        _, _, dimg2 = image_loss_value_and_grad(pkg2["render"].detach(), self._gt_stack(grp, grey_mean=_GT_GREY),
                                                                c["lambda_dssim"], c["lambda_image"])
                        g2, = torch.autograd.grad([pkg2["render"]], [means3D], grad_outputs=[dimg2])
                    main.wait_stream(gs)
                    g_means = g_means.add_(g2)
            extra = None
            if gd is not None:  # added inside the hidden<-visual backward instead of by a pass over g_means
                main.wait_stream(self.dist_stream)
                extra = (gd, float(c["lambda_current_distance"]) * n_dist_weight)
            gm.defer_render_means_gradient(g_means, extra)  # -> the one hidden<-visual backward of the iteration
        if gp is not None:
            main.wait_stream(self.side_stream)
        multi = self.world > 1 or self.force_all_reduce
        if self.fused_step and not multi:
            gm.fused_step_current(batch, [(gp, float(n_phys))] if gp is not None and n_phys else [])
            return
        if gp is not None and n_phys:
            gm.accumulate_gradient_current(gp, n_phys)
        gm.flush_deferred_gradients()
        if multi:
            if phase == "local":
                # the caller all-reduces self._reduce_buf outside the captured graph, then runs _finish_step
                self._reduce_buf.copy_(gm._estimate_xyz_nn_grad)
                return
            dist.all_reduce(gm._estimate_xyz_nn_grad, op=dist.ReduceOp.SUM)
        self._finish_step(batch)

    def _finish_step(self, batch, grad=None):
        """Batch mean of the (reduced) gradient cache + optimiser step."""
        gm = self.gm
        if grad is not None:
            gm._estimate_xyz_nn_grad = grad
        if self.fused_step:
            gm._grad_cache_used = True
            gm.fused_step_current(batch)
            return
        gm.set_batch_gradient_current(batch)
        gm.optimizer.step()
        gm.optimizer.zero_grad()

    def _iteration_body(self):
        if self.batched_views:
            return self._iteration_body_batched()
        if self.parallel_views:
            return self._iteration_body_parallel()
        gm = self.gm
        self.itr += 1
        gm.total_iterations += 1
        gm.update_learning_rate_current(self.itr)
        gm.zero_gradient_cache_current()
        batch = len(self.cams)  # the benchmark renders every view each iteration (BASELINE: 5 views/iter)
        mine = self._mine(batch)
        for n, v in enumerate(mine):
            cam = self.cams[v]
            pkg = self.render_func(cam, gm, None, self.background, GRsetting=self.GRsetting, GRzer=self.GRzer,
                                   pos_type="guess_visual_nn", scale=True)
            loss, l1_value, ssim_value = self._image_loss(pkg["render"], cam.original_image)
            if self.cfg.get("lambda_current_distance", 0.0) > 0:  # tpp:365-366
                from .utils.loss_utils import distance_loss
                loss = loss + self.cfg["lambda_current_distance"] * distance_loss(pkg["render_xyz"],
                                                                                 self.cfg["distance_threshold_visual"])
            if self.physics_per_view:
                loss = loss + self._physics_loss()          # as the reference: once per view (tpp:368-389)
            elif n == 0 and self.rank == 0:
                loss = loss + batch * self._physics_loss()  # same gradient after the 1/batch scaling
            if self.log_scalars:
                self.last = dict(l1=l1_value.item(), ssim=ssim_value.item(), total=loss.item())
            loss.backward()
            gm.cache_gradient_current()
            gm.optimizer.zero_grad()
        gm.flush_deferred_gradients()
        if self.world > 1 or self.force_all_reduce:
            dist.all_reduce(gm._estimate_xyz_nn_grad, op=dist.ReduceOp.SUM)
        gm.set_batch_gradient_current(batch)
        gm.optimizer.step()
        gm.optimizer.zero_grad()


class HotLoopLevelTwo:
    """Visual-particle stage of one frame (train_visual_particle.py:133-222): positions fixed, the
    visual particles' colour / opacity / scales / rotation are optimised against all views with
    L1 + D-SSIM on the RGB image plus consistency-to-previous-frame L2 terms; gradients are cached per
    view, averaged over the batch (gm_dynamics.py:474-503) and applied with Adam(eps = 1e-15)."""

    def __init__(self, gm, cams, rd_pipe="render_dynamics", rank=0, world=1, cfg=SMOKE_L2, image_loss="fused",
                 log_scalars=False, batched_views=False, capturable=False, force_all_reduce=False, fused_attributes=None):
        """batched_views: the views of the batch through ONE view-batched render / fused image loss / backward (all
        four attribute gradients out of the rasteriser's full backward, the static background binned once), the
        view-independent consistency terms evaluated once and counted once per view; capturable: device-side fused
        Adam over the four attribute groups, so that capture() can record whole iterations as a hipGraph;
        fused_attributes (default: with batched_views): activations, consistency / scale-ratio terms and the batch mean
        in two kernels (fnx_level2_activate / fnx_level2_backward) instead of ~90 elementwise launches."""
        from .utils.loss_utils import l2_loss_consistency
        self.gm, self.cams, self.rank, self.world, self.cfg = gm, cams, rank, world, dict(cfg)
        self.view_subset = None
        self.render_func, self.GRsetting, self.GRzer = get_render_pipe(rd_pipe)
        self.image_loss, self.log_scalars = image_loss, log_scalars
        self.batched_views, self.capturable, self.force_all_reduce = bool(batched_views), bool(capturable), force_all_reduce
        assert not capturable or batched_views, "graph capture is implemented for the view-batched level-two loop"
        self.fused_attributes = self.batched_views if fused_attributes is None else bool(fused_attributes)
        assert not self.fused_attributes or (batched_views and gm._visual_color.shape[1] == 1)
        self._attr, self._flat_grad = None, None
        self.background = torch.zeros(3, device=gm._visual_xyz.device)
        self.prev = {n: getattr(gm, f"_visual_{n}").detach().clone() for n in gm._L2}
        self._cons = l2_loss_consistency
        gm.training_setup_current_level_two(SimpleNamespace(**{k: cfg[k] for k in cfg if k.endswith("_lr")}),
                                            capturable=capturable)
        self.last = {}
        self.stream = torch.cuda.Stream(device=gm._visual_xyz.device) if capturable else None
        self.graph, self.graph_iterations, self._replay, self._gt, self._means = None, 1, False, None, None

    @property
    def multi(self):
        return self.world > 1 or self.force_all_reduce

    def _gt_stack(self, mine):
        imgs = [self.cams[v].original_image for v in mine]
        if self._gt is None or len(self._gt[0]) != len(imgs) or any(a is not b for a, b in zip(self._gt[0], imgs)):
            self._gt = (imgs, torch.stack(imgs).contiguous())
        return self._gt[1]

    def _render_means(self):
        """[visual positions / scale_factor | background positions]: fixed in this stage, built once."""
        gm = self.gm
        src = (gm._visual_xyz, gm._gs_xyz)
        if self._means is None or self._means[0][0] is not src[0] or self._means[0][1] is not src[1]:
            with torch.no_grad():
                self._means = (src, torch.cat([gm._visual_xyz / gm.scale_factor, gm._gs_xyz], dim=0).float().contiguous())
        return self._means[1]

    def _regularisers(self):
        """The view-independent terms of one view's loss (train_visual_particle.py:161-194): consistency with the
        previous frame per active attribute, and the scale-ratio regulariser."""
        gm, c = self.gm, self.cfg
        reg = 0.0
        for n in gm._l2_active():
            reg = reg + c[f"lambda_consistency_{n}"] * self._cons(getattr(gm, f"_visual_{n}"), self.prev[n])
        if "scales" in gm._l2_active() and c["lambda_reg_scaling"] > 0:
            sc = gm.get_visual_scaling
            ratio = torch.max(sc, dim=1).values / torch.min(sc, dim=1).values
            reg = reg + c["lambda_reg_scaling"] * torch.clamp_min(ratio - c["scaling_reg_ratio_threshold"], 0).mean()
        return reg

    def _attribute_arrays(self):
        """Persistent [fluid | background] attribute arrays for the fused path: the background rows (fixed in this
        stage) are written once, the fluid rows by fnx_level2_activate every iteration."""
        gm = self.gm
        src = (gm._gs_opacity, gm._gs_scales, gm._gs_rotation, gm._gs_color)
        key = tuple((t, t._version) for t in src) + (gm._visual_color.shape[0],)
        if self._attr is None or len(self._attr[0]) != len(key) or any(
                (a[0] is not b[0] or a[1] != b[1]) if isinstance(a, tuple) else a != b for a, b in zip(self._attr[0], key)):
            n = gm._visual_color.shape[0]
            with torch.no_grad():
                dev = gm._visual_color.device
                gs = dict(opacity=gm.get_gs_opacity, scales=gm.get_gs_scaling, rotation=gm.get_gs_rotation,
                          color=gm.get_gs_color)
                arrays = {}
                for k, t in gs.items():
                    a = torch.zeros((n + t.shape[0], t.shape[1]), device=dev, dtype=torch.float32)
                    a[n:] = t.float()
                    arrays[k] = a
            self._attr = (key, arrays)
        return self._attr[1]

    def _body_fused(self):
        from .losses import image_loss_value_and_grad, level2_activate, level2_backward
        from .renderer.pipes import render_dynamics_views
        gm, c = self.gm, self.cfg
        gm.total_iterations += 1
        batch = len(self.cams)
        mine = self._mine(batch)
        names = gm._l2_active()
        raw = {n: getattr(gm, f"_visual_{n}") for n in gm._L2}
        # the gradients of the fitted attributes live in ONE flat buffer (views of it are the parameters' .grad), so
        # that a multi-rank step needs one all-reduce ("one fused flat buffer per iteration")
        flat = self._flat_grad
        sizes = [raw[n].numel() for n in names]
        if flat is None or flat.numel() != sum(sizes) or any(raw[n].grad is None for n in names):
            flat = self._flat_grad = torch.zeros(sum(sizes), device=raw["color"].device, dtype=torch.float32)
            off = 0
            for n, k in zip(names, sizes):
                raw[n].grad = flat[off:off + k].view_as(raw[n])
                off += k
        out = {n: raw[n].grad for n in names}
        if mine:
            arrays = self._attribute_arrays()
            level2_activate({k: t.detach() for k, t in raw.items()}, arrays)
            leaves = {k: arrays[k].detach().requires_grad_() for k in gm._L2}
            pkg = render_dynamics_views([self.cams[v] for v in mine], gm, None, self.background, GRsetting=self.GRsetting,
                                        GRzer=self.GRzer, pos_type="visual", scale=True, means3D=self._render_means(),
                                        attributes=(leaves["opacity"], leaves["scales"], leaves["rotation"], leaves["color"]),
                                        screen_grad=False)  # positions are fixed: no 2D-mean gradient sums
            loss, per_view, dimg = image_loss_value_and_grad(pkg["render"].detach(), self._gt_stack(mine), c["lambda_dssim"],
                                                             c["lambda_image"], grey=False)
            if self.log_scalars:
                self.last = dict(l1=per_view[-1, 0].item(), ssim=1.0 - per_view[-1, 1].item(), total=loss.item())
            g = dict(zip(gm._L2, torch.autograd.grad([pkg["render"]], [leaves[k] for k in gm._L2], grad_outputs=[dimg])))
            with torch.no_grad():
                level2_backward({k: t.detach() for k, t in raw.items()}, self.prev, g, out,
                                {k: c[f"lambda_consistency_{k}"] for k in gm._L2},
                                c["lambda_reg_scaling"] if "scales" in names else 0.0, c["scaling_reg_ratio_threshold"],
                                float(len(mine)), 1.0 / batch)
        else:
            flat.zero_()
        if self.multi:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        # set_batch_gradient_current_level_two (gm_dynamics.py:494-503) happened in the kernel; the step on the torch
        # optimiser's own state, one launch per attribute group (torch's capturable fused Adam takes ~45 us per group)
        if self.capturable:
            from .physics import adam_step
            for n in names:
                adam_step(raw[n], gm.optimizer, [(out[n], 1.0)], 1.0)
        else:
            gm.optimizer.step()

    def _body_batched(self):
        if self.fused_attributes:
            return self._body_fused()
        from .losses import image_loss_value_and_grad
        from .renderer.pipes import render_dynamics_views
        gm, c = self.gm, self.cfg
        gm.total_iterations += 1
        batch = len(self.cams)
        mine = self._mine(batch)
        names = gm._l2_active()
        params = [getattr(gm, f"_visual_{n}") for n in names]
        grads = [None] * len(params)
        if mine:
            pkg = render_dynamics_views([self.cams[v] for v in mine], gm, None, self.background, GRsetting=self.GRsetting,
                                        GRzer=self.GRzer, pos_type="visual", scale=True, means3D=self._render_means())
            loss, per_view, dimg = image_loss_value_and_grad(pkg["render"].detach(), self._gt_stack(mine), c["lambda_dssim"],
                                                             c["lambda_image"], grey=False)
            if self.log_scalars:
                self.last = dict(l1=per_view[-1, 0].item(), ssim=1.0 - per_view[-1, 1].item(), total=loss.item())
            reg = self._regularisers()
            if torch.is_tensor(reg):  # one backward for the image term of all views + n_local x the per-view regularisers
                grads = torch.autograd.grad([pkg["render"], reg], params, grad_outputs=[dimg, torch.full_like(reg, float(len(mine)))],
                                            allow_unused=True)
            else:
                grads = torch.autograd.grad([pkg["render"]], params, grad_outputs=[dimg], allow_unused=True)
        for n, p, g in zip(names, params, grads):
            g = torch.zeros_like(p) if g is None else g
            if self.multi:
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
            p.grad = g * (1.0 / batch)  # set_batch_gradient_current_level_two (gm_dynamics.py:494-503)
        gm.optimizer.step()
        gm.optimizer.zero_grad()

    def capture(self, warmup=2, iterations=1):
        from . import rasterizer
        assert self.capturable and not rasterizer._HOST_SYNC and not self.multi
        for _ in range(warmup):
            self.iteration()
        torch.cuda.synchronize()
        rasterizer._pending_status.clear()
        g = torch.cuda.CUDAGraph()
        tot0 = self.gm.total_iterations
        with _graph_capture(g, self.stream):
            for _ in range(int(iterations)):
                self._body_batched()
        self.gm.total_iterations = tot0
        self.graph, self.graph_iterations, self._replay = g, int(iterations), True
        return g

    @property
    def iterations_per_call(self):
        return self.graph_iterations if (self.graph is not None and self._replay) else 1

    def use_graph(self, enabled):
        self._replay = bool(enabled) and self.graph is not None

    def _mine(self, batch):
        return list(self.view_subset) if self.view_subset is not None else shard_views(batch, self.rank, self.world)

    @torch.no_grad()
    def make_targets(self, noise=0.3, seed=0):
        """Synthetic ground truth: the scene rendered with perturbed visual colours / opacities."""
        gm = self.gm
        g = torch.Generator(device="cpu").manual_seed(seed)
        keep = {n: getattr(gm, f"_visual_{n}").data.clone() for n in ("color", "opacity")}
        for n in keep:
            t = getattr(gm, f"_visual_{n}")
            t.data += noise * torch.randn(t.shape, generator=g).to(t.device)
        gm._visual_color.data.clamp_(0.0, 1.0)
        for cam in self.cams:
            pkg = self.render_func(cam, gm, None, self.background, GRsetting=self.GRsetting, GRzer=self.GRzer,
                                   pos_type="visual", scale=True)
            cam.original_image = pkg["render"].detach().clamp(0, 1).clone()
        for n, v in keep.items():
            getattr(gm, f"_visual_{n}").data.copy_(v)

    def iteration(self):
        if self.batched_views:
            if self.stream is None:
                return self._body_batched()
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                if self.graph is not None and self._replay:
                    self.gm.total_iterations += self.graph_iterations
                    self.graph.replay()
                else:
                    self._body_batched()
            torch.cuda.current_stream().wait_stream(self.stream)
            return
        gm, c = self.gm, self.cfg
        gm.total_iterations += 1
        gm.zero_gradient_cache_current_level_two()
        batch = len(self.cams)
        for v in self._mine(batch):
            cam = self.cams[v]
            pkg = self.render_func(cam, gm, None, self.background, GRsetting=self.GRsetting, GRzer=self.GRzer,
                                   pos_type="visual", scale=True)
            image, gt = pkg["render"], cam.original_image
            if self.image_loss == "fused":
                from .losses import fused_l1_ssim
                l1_value, s = fused_l1_ssim(image, gt)
                ssim_value = 1.0 - s
            else:
                l1_value, ssim_value = l1_loss(image, gt), 1.0 - ssim(image, gt)
            loss = ((1.0 - c["lambda_dssim"]) * l1_value + c["lambda_dssim"] * ssim_value) * c["lambda_image"]
            for n in gm._l2_active():
                loss = loss + c[f"lambda_consistency_{n}"] * self._cons(getattr(gm, f"_visual_{n}"), self.prev[n])
            if "scales" in gm._l2_active() and c["lambda_reg_scaling"] > 0:
                sc = gm.get_visual_scaling
                ratio = torch.max(sc, dim=1).values / torch.min(sc, dim=1).values
                loss = loss + c["lambda_reg_scaling"] * torch.clamp_min(ratio - c["scaling_reg_ratio_threshold"], 0).mean()
            if self.log_scalars:
                self.last = dict(l1=l1_value.item(), ssim=ssim_value.item(), total=loss.item())
            loss.backward()
            gm.cache_gradient_current_level_two()
            gm.optimizer.zero_grad()
        if self.world > 1:
            for n in gm._l2_active():
                dist.all_reduce(gm._l2_grad[n], op=dist.ReduceOp.SUM)
        gm.set_batch_gradient_current_level_two(batch)
        gm.optimizer.step()
        gm.optimizer.zero_grad()


class FirstFrameLoop:
    """First-frame stage (entries_fluid_nexus/train_physical_particle.py:103-163 with render_dynamics + the grey-mean
    image term; entries_scalar_real/train_physical_particle.py:97-165 with render_fluid on the 1-channel image): the
    positions of the visual particles are the leaf, pos_type="visual", loss = L1 + D-SSIM (+ lambda_first_distance x
    distance_loss(visual_xyz, distance_threshold_visual)), gradient mean over the batch, Adam(eps = 1e-15).
    The views of the batch go through one view-batched launch sequence; `capture()` records k iterations as one
    hipGraph.  Multi-GPU: views sharded, one all-reduce of the position gradient."""

    def __init__(self, gm, cams, rd_pipe="render_fluid", rank=0, world=1, cfg=SCALAR_REAL, capturable=True,
                 force_all_reduce=False, log_scalars=False):
        self.gm, self.cams, self.rank, self.world, self.cfg = gm, cams, rank, world, dict(cfg)
        self.view_subset = None
        self.rd_pipe = rd_pipe
        self.render_func, self.GRsetting, self.GRzer = get_render_pipe(rd_pipe)
        assert rd_pipe in ("render_fluid", "render_dynamics")
        self.grey = rd_pipe == "render_dynamics"  # tpp:131-135 compares grey-mean images; ScalarReal is single-channel
        self.force_all_reduce, self.log_scalars = force_all_reduce, log_scalars
        self.background = torch.zeros(3, device=gm._visual_xyz.device)
        args = SimpleNamespace(**{k: cfg[k] for k in ("position_lr_init", "position_lr_final", "position_lr_delay_mult",
                                                      "position_lr_max_steps")})
        gm.training_setup_first_visual(args, capturable=capturable)
        self.capturable = capturable
        self.stream = torch.cuda.Stream(device=gm._visual_xyz.device) if capturable else None
        self.graph, self.graph_iterations, self._replay, self._reduce_buf = None, 1, False, None
        self.itr, self.last, self._gt, self.dist_stream = 0, {}, None, None

    @property
    def multi(self):
        return self.world > 1 or self.force_all_reduce

    def _mine(self, batch):
        return list(self.view_subset) if self.view_subset is not None else shard_views(batch, self.rank, self.world)

    @torch.no_grad()
    def make_targets(self, shift=(0.004, 0.002, 0.0)):
        """Synthetic ground truth: the scene rendered with the visual particles displaced by `shift` (world units)."""
        gm = self.gm
        keep = gm._visual_xyz.data.clone()
        gm._visual_xyz.data += torch.tensor(shift, device=keep.device)
        for cam in self.cams:
            pkg = self.render_func(cam, gm, None, self.background, GRsetting=self.GRsetting, GRzer=self.GRzer,
                                   pos_type="visual")
            cam.original_image = pkg["render"].detach().clamp(0, 1).clone()
        gm._visual_xyz.data.copy_(keep)
        self._gt = None

    def _gt_stack(self, mine):
        imgs = [self.cams[v].original_image for v in mine]
        if self._gt is None or len(self._gt[0]) != len(imgs) or any(a is not b for a, b in zip(self._gt[0], imgs)):
            self._gt = (imgs, torch.stack(imgs).contiguous())
        return self._gt[1]

    def _local_gradient(self):
        """Sum over this rank's views of d loss_v / d visual_xyz as a list of (tensor, scale) terms."""
        from .losses import image_loss_value_and_grad
        from .renderer.pipes import render_dynamics_views, render_fluid_views
        gm, c = self.gm, self.cfg
        mine = self._mine(len(self.cams))
        param = gm._visual_xyz
        V = param.shape[0]
        terms = []
        if mine:
            # distance_loss(visual positions) is the same for every view (tpp:141-144): evaluated once, on its own
            # stream next to the rasteriser (it needs the leaf only), added once per local view
            from . import rasterizer
            gd, main = None, torch.cuda.current_stream()
            use_dist = c.get("lambda_first_distance", 0.0) > 0
            # the distance branch forks between the rasteriser's binning stage and its emit / blend stage (see HotLoop):
            # next to the latency-bound depth sort its grid build stretches the critical path, next to the blend it is free
            fork, forked = torch.cuda.Event(), []
            if use_dist:
                rasterizer.set_between_stages_hook(lambda: (fork.record(torch.cuda.current_stream()), forked.append(1)))
            cams = [self.cams[v] for v in mine]
            try:
                if self.rd_pipe == "render_fluid":
                    means = param
                    pkg = render_fluid_views(cams, gm, None, self.background, GRsetting=self.GRsetting, GRzer=self.GRzer,
                                             pos_type="visual", means3D=means, screen_grad=_SCREEN_GRAD)
                else:
                    means = gm.render_means_from_visual()
                    pkg = render_dynamics_views(cams, gm, None, self.background, GRsetting=self.GRsetting,
                                                GRzer=self.GRzer, pos_type="visual", scale=False, means3D=means,
                                                screen_grad=_SCREEN_GRAD)
            finally:
                rasterizer.set_between_stages_hook(None)
            loss, per_view, dimg = image_loss_value_and_grad(pkg["render"].detach(), self._gt_stack(mine),
                                                             c["lambda_dssim"], 1.0, grey=self.grey)
            if self.log_scalars:
                self.last = dict(l1=per_view[-1, 0].item(), ssim=1.0 - per_view[-1, 1].item(), total=loss.item())
            g, = torch.autograd.grad([pkg["render"]], [means], grad_outputs=[dimg])
            terms.append((g[:V], 1.0))
            if use_dist:
                # enqueued AFTER the rasteriser chain (a captured graph keeps a node's first-enqueued successor on its
                # hardware queue: the critical path must not be the branch that hops)
                from .physics import distance_loss_value_and_grad
                if self.dist_stream is None:
                    self.dist_stream = torch.cuda.Stream(device=param.device)
                if forked:
                    self.dist_stream.wait_event(fork)
                else:  # the forward did not pass the hook: order the branch behind everything enqueued so far
                    self.dist_stream.wait_stream(main)
                with torch.cuda.stream(self.dist_stream):
                    _, gd = distance_loss_value_and_grad(param.detach(), c["distance_threshold_visual"])
                main.wait_stream(self.dist_stream)
                terms.append((gd, float(c["lambda_first_distance"]) * len(mine)))
        return terms

    def _body(self, phase="all"):
        from . import physics
        gm = self.gm
        self.itr += 1
        gm.total_iterations += 1
        gm.update_learning_rate_first_visual(self.itr)
        batch = len(self.cams)
        terms = self._local_gradient()
        if self.multi:
            buf = self._reduce_buf
            buf.zero_()
            for t, sc in terms:
                buf.add_(t, alpha=sc)
            if phase == "local":
                return
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            terms = [(buf, 1.0)]
        physics.adam_step(gm._visual_xyz, gm.optimizer, terms, batch)

    def _finish(self):
        from . import physics
        physics.adam_step(self.gm._visual_xyz, self.gm.optimizer, [(self._reduce_buf, 1.0)], len(self.cams))

    def capture(self, warmup=2, iterations=1):
        from . import rasterizer
        assert self.capturable and not rasterizer._HOST_SYNC
        if self.multi and self._reduce_buf is None:
            self._reduce_buf = torch.zeros_like(self.gm._visual_xyz.detach())
        for _ in range(warmup):
            self.iteration()
        torch.cuda.synchronize()
        rasterizer._pending_status.clear()
        g = torch.cuda.CUDAGraph()
        itr0, tot0 = self.itr, self.gm.total_iterations
        with _graph_capture(g, self.stream):
            if self.multi:  # local gradient | all-reduce outside the graph | one-kernel step launched eagerly
                self._body(phase="local")
                iterations = 1
            else:
                for _ in range(int(iterations)):
                    self._body()
        self.itr, self.gm.total_iterations = itr0, tot0
        self.graph, self.graph_iterations, self._replay = g, int(iterations), True
        return g

    @property
    def iterations_per_call(self):
        return self.graph_iterations if (self.graph is not None and self._replay) else 1

    def use_graph(self, enabled):
        self._replay = bool(enabled) and self.graph is not None

    def iteration(self):
        if self.multi and self._reduce_buf is None:
            self._reduce_buf = torch.zeros_like(self.gm._visual_xyz.detach())
        if self.stream is None:
            return self._body()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            if self.graph is not None and self._replay:
                self.itr += self.graph_iterations
                self.gm.total_iterations += self.graph_iterations
                self.graph.replay()
                if self.multi:
                    dist.all_reduce(self._reduce_buf, op=dist.ReduceOp.SUM)
                    self._finish()
            else:
                self._body()
        torch.cuda.current_stream().wait_stream(self.stream)
