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
    q.put((rank, gm._estimate_xyz_nn.grad.clone()))
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
        assert torch.allclose(got[r], ref._estimate_xyz_nn.grad, rtol=1e-6, atol=1e-7)
    assert torch.equal(got[0], got[1])  # replicated optimiser sees identical gradients
