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
