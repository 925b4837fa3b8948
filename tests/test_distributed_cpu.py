"""world_size-2 gloo test of the multi-GPU gradient semantics: views sharded round-robin,
one all-reduce(sum) of the cached leaf gradient, then the reference's 1/batch scaling
(gm_dynamics.py:461-472) -- must equal the single-process loop over all views."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fluidnexus_amd.gaussian_splatting.gm_dynamics import GaussianModel
from fluidnexus_amd.harness import shard_views


def _toy_view_loss(x, v):
    """A differentiable stand-in for 'render view v and compare': depends on the view index."""
    w = torch.linspace(0.5, 1.5, x.numel()).reshape(x.shape) * (v + 1)
    return ((x * w).sin() ** 2).sum() + 0.1 * (x ** 2).sum()


def _accumulate(gm, views):
    gm.zero_gradient_cache_current()
    for v in views:
        _toy_view_loss(gm._estimate_xyz_nn, v).backward()
        gm.cache_gradient_current()
        gm._estimate_xyz_nn.grad = None


def _model():
    torch.manual_seed(0)
    gm = GaussianModel()
    gm._estimate_xyz_nn = torch.nn.Parameter(torch.randn(50, 3))
    return gm


def _worker(rank, world, port, n_views, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gm = _model()
    _accumulate(gm, shard_views(n_views, rank, world))
    dist.all_reduce(gm._estimate_xyz_nn_grad, op=dist.ReduceOp.SUM)
    gm.set_batch_gradient_current(n_views)
    q.put((rank, gm._estimate_xyz_nn.grad.numpy().copy()))  # plain data: the worker may exit before it is read
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gradient_equals_serial():
    n_views, world = 5, 2
    ref = _model()
    _accumulate(ref, range(n_views))
    ref.set_batch_gradient_current(n_views)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_views, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert torch.allclose(torch.from_numpy(got[r]), ref._estimate_xyz_nn.grad, rtol=1e-6, atol=1e-7)
    assert (got[0] == got[1]).all()  # replicated optimiser sees identical gradients


# ---- the real HotLoop bookkeeping (shard -> per-view backward -> gradient cache -> all-reduce -> 1/batch -> Adam) ----
def _stub_render(cam, gm, pipe_args, bg, **kw):
    """Stands in for render_dynamics on the host: a smooth image-valued function of the optimised positions that
    depends on the camera (cam.uid).  Everything else the loop does is its own code."""
    x = gm._estimate_xyz_nn
    gen = torch.Generator().manual_seed(100 + cam.uid)
    A = torch.randn(x.numel(), 3 * 16 * 16, generator=gen) * 0.3
    return {"render": torch.sigmoid(x.reshape(1, -1) @ A).reshape(3, 16, 16), "render_xyz": x}


def _host_loop(rank, world, n_views):
    from types import SimpleNamespace

    from fluidnexus_amd.harness import SMOKE, HotLoop
    torch.manual_seed(0)
    gm = GaussianModel(device="cpu")
    gm.setup_constants()
    N = 40
    gm._xyz = torch.randn(N, 3)
    gm._estimate_xyz = gm._xyz + 0.1 * torch.randn(N, 3)
    cams = [SimpleNamespace(uid=v, original_image=torch.rand(3, 16, 16, generator=torch.Generator().manual_seed(v)))
            for v in range(n_views)]
    cfg = dict(SMOKE, lambda_exyz=0.0, lambda_gas_constraints=0.0, lambda_next_gas_constraints=0.0,
               lambda_current_distance=0.0)  # the particle terms need the HIP kernels; the image term drives this test
    loop = HotLoop(gm, cams, rank=rank, world=world, cfg=cfg, image_loss="torch")
    loop.render_func = _stub_render
    return gm, loop


def _loop_worker(rank, world, port, n_views, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gm, loop = _host_loop(rank, world, n_views)
    assert loop._mine(n_views) == [v for v in range(n_views) if v % world == rank]
    for _ in range(steps):
        loop.iteration()
    q.put((rank, gm._estimate_xyz_nn.detach().numpy().copy()))  # plain data: the worker may exit before it is read
    dist.barrier()
    dist.destroy_process_group()


def test_hot_loop_sharded_equals_single_process():
    """HotLoop itself (harness.py) with 5 views over 2 ranks (3 / 2) against the same loop in one process: same
    particle positions after a few optimiser steps, identical on both ranks."""
    n_views, world, steps = 5, 2, 3
    ref_gm, ref_loop = _host_loop(0, 1, n_views)
    x0 = ref_gm._estimate_xyz_nn.detach().clone()
    for _ in range(steps):
        ref_loop.iteration()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, n_views, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = ref_gm._estimate_xyz_nn.detach()
    assert float((want - x0).abs().max()) > 1e-4  # the optimiser moved the particles
    assert (got[0] == got[1]).all()
    assert torch.allclose(torch.from_numpy(got[0]), want, rtol=0, atol=2e-6)


# ---- the view-batched multi-rank path every real N-GPU run takes --------------------------------------------------
# _iteration_body_batched(phase="local") -> all-reduce of HotLoop._reduce_buf -> _finish_step (harness.py).  On the host
# the HIP-backed pieces are replaced by linear / smooth stand-ins (the hidden -> visual interpolation by a fixed matrix,
# the view-batched render by a smooth map of the positions that depends on the camera, the fused image loss by its
# torch expression, the fused physics stage by a quadratic) and the HIP stream objects by no-ops; the sharding, the
# once-per-local-view physics term, the reduce buffer, the all-reduce, the 1/batch mean and the optimiser step are
# the loop's own code.
class _NoStream:
    def wait_event(self, e):
        pass

    def wait_stream(self, s):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _NoEvent:
    def record(self, s=None):
        pass


def _batched_host_loop(rank, world, n_views, shared_rank=None):
    from types import SimpleNamespace

    import fluidnexus_amd.losses as losses
    import fluidnexus_amd.physics as physics
    import fluidnexus_amd.renderer.pipes as pipes
    from fluidnexus_amd.harness import SMOKE, HotLoop

    torch.cuda.current_stream = lambda *a, **k: _NoStream()
    torch.cuda.Stream = lambda *a, **k: _NoStream()
    torch.cuda.Event = lambda *a, **k: _NoEvent()
    torch.cuda.stream = lambda s: _NoStream()

    torch.manual_seed(0)
    gm = GaussianModel(device="cpu")
    gm.setup_constants()
    N, V = 40, 24
    gm._xyz = torch.randn(N, 3)
    gm._estimate_xyz = gm._xyz + 0.1 * torch.randn(N, 3)
    gm._visual_xyz = torch.randn(V, 3)
    gm._gs_xyz = torch.randn(6, 3)
    A = torch.randn(V, N, generator=torch.Generator().manual_seed(5)) * 0.2  # visual <- hidden interpolation
    state = {}

    def render_means_from_nn():
        leaf = torch.cat([A @ gm._estimate_xyz_nn.detach(), gm._gs_xyz]).requires_grad_(True)
        state["leaf"] = leaf
        return leaf

    def defer_render_means_gradient(g, extra=None):
        state["g"] = g[:V] if extra is None else g[:V] + float(extra[1]) * extra[0]

    def flush_deferred_gradients():
        g = state.pop("g", None)
        if g is not None:
            gm._estimate_xyz_nn_grad += A.t() @ g
            gm._grad_cache_used = True

    gm.render_means_from_nn = render_means_from_nn
    gm.defer_render_means_gradient = defer_render_means_gradient
    gm.flush_deferred_gradients = flush_deferred_gradients

    def render_dynamics_views(cams, gm_, pipe_args, bg, means3D=None, **kw):
        imgs = []
        for cam in cams:
            B = torch.randn(means3D.numel(), 3 * 8 * 8, generator=torch.Generator().manual_seed(100 + cam.uid)) * 0.3
            imgs.append(torch.sigmoid(means3D.reshape(1, -1) @ B).reshape(3, 8, 8))
        return {"render": torch.stack(imgs)}

    def image_loss_value_and_grad(images, gts, lambda_dssim, lambda_image, grey=True):
        x = images.detach().clone().requires_grad_(True)
        per = ((x - gts) ** 2).mean(dim=(1, 2, 3))  # a smooth per-view image term (the fused kernel's is L1 + D-SSIM)
        loss = per.sum() * lambda_image
        g, = torch.autograd.grad(loss, x)
        return loss.detach(), torch.stack([per.detach(), per.detach()], 1), g

    full = (SMOKE["lambda_exyz"], SMOKE["lambda_gas_constraints"], SMOKE["lambda_next_gas_constraints"])

    def physical_stage_value_and_grad(gm_, l1, l2, l3, memo):
        # the same on every rank, like the real (view-independent) terms; three terms, each switched by its own weight
        # (a rank of a "spread" run passes zeros for the terms it does not own -- fnx_physical_stage skips those)
        x = gm_._estimate_xyz_nn.detach()
        k = 0.004 * l1 / full[0] + 0.005 * l2 / full[1] + 0.001 * l3 / full[2]
        return 0.5 * k * (x ** 2).sum(), k * x

    def distance_loss_value_and_grad(xyz, thr):  # a smooth view-independent term of the rendered positions
        return 0.5 * (xyz ** 2).sum(), xyz.clone()

    pipes.render_dynamics_views = render_dynamics_views
    losses.image_loss_value_and_grad = image_loss_value_and_grad
    physics.physical_stage_value_and_grad = physical_stage_value_and_grad
    physics.distance_loss_value_and_grad = distance_loss_value_and_grad

    cams = [SimpleNamespace(uid=v, original_image=torch.rand(3, 8, 8, generator=torch.Generator().manual_seed(v)))
            for v in range(n_views)]
    cfg = dict(SMOKE, lambda_current_distance=0.03)
    loop = HotLoop(gm, cams, rank=rank, world=world, cfg=cfg, image_loss="fused", fused_physics=True,
                   defer_visual_backward=True, batched_views=True, force_all_reduce=world == 1,
                   shared_terms_rank=shared_rank)
    return gm, loop


def _batched_step(loop, world):
    """What HotLoop.iteration() does per replay in a multi-rank run (harness.py: graph replay of the local phase, the
    all-reduce outside the graph, then the finish step)."""
    if loop._reduce_buf is None:
        loop._reduce_buf = torch.zeros_like(loop.gm._estimate_xyz_nn.detach())
    loop._iteration_body_batched(phase="local")
    dist.all_reduce(loop._reduce_buf, op=dist.ReduceOp.SUM)
    loop._finish_step(len(loop.cams), grad=loop._reduce_buf)


def _batched_worker(rank, world, port, n_views, steps, q, shared_rank=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gm, loop = _batched_host_loop(rank, world, n_views, shared_rank)
    for _ in range(steps):
        _batched_step(loop, world)
    q.put((rank, gm._estimate_xyz_nn.detach().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def _serial_reference_worker(port, n_views, steps, q):
    """One rank, the un-phased batched body (phase="all": in-line all-reduce over a 1-rank group)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    gm, loop = _batched_host_loop(0, 1, n_views)
    x0 = gm._estimate_xyz_nn.detach().numpy().copy()
    for _ in range(steps):
        loop._iteration_body_batched(phase="all")
    q.put(("ref", (x0, gm._estimate_xyz_nn.detach().numpy().copy())))
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("shared_rank", [None, 1, "spread"])
def test_batched_local_phase_all_reduce_finish_step_equals_single_rank(shared_rank):
    """5 views over 2 ranks (3 / 2) through `_iteration_body_batched(phase="local")` + all-reduce + `_finish_step`
    against one rank running all 5 views through the un-phased body: same positions after three optimiser steps,
    identical on both ranks.  shared_rank None: the physics and distance terms are added once per LOCAL view on every rank
    (3 + 2 = 5 = the single rank's count); 1: only rank 1 (the one with fewer views) evaluates them and adds them 5 times
    (bench.py --shared-terms last-rank); "spread": each of the three terms is evaluated by ONE rank (harness.spread_owners)
    and added 5 times by it -- what bench.py does in a multi-rank run since round 6."""
    n_views, world, steps = 5, 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    ref_p = ctx.Process(target=_serial_reference_worker, args=(port + 1, n_views, steps, q))
    ref_p.start()
    procs = [ctx.Process(target=_batched_worker, args=(r, world, port, n_views, steps, q, shared_rank)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world + 1))
    for p in procs + [ref_p]:
        p.join(timeout=60)
        assert p.exitcode == 0
    x0, want = got["ref"]
    assert float(abs(want - x0).max()) > 1e-4
    assert (got[0] == got[1]).all()
    assert abs(got[0] - want).max() <= 2e-6
