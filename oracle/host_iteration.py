"""ONE iteration of the physical-particle loop composed on the HOST -- TEST INFRASTRUCTURE ONLY (tests/ and bench.py's
cpu_baseline leg; no product path imports this file).

Reference: entries_fluid_nexus/train_physical_particle.py:329-432 (zero cache -> per view: render_dynamics
(guess_visual_nn, scale) -> grey-mean L1 + D-SSIM -> distance_loss -> exyz / gas / next-gas terms -> backward -> cache) ->
batch mean (gm_dynamics.py:461-472) -> Adam step.  Pieces: oracle/raster_oracle.c per view (parity unpinned, see its
header), oracle/physics_oracle.py (pinned by tests/golden/physics.npz), fluidnexus_amd.utils.loss_utils on the host
(pinned by tests/golden/loss_utils.npz), torch.optim.Adam.

Two neighbour searches: the oracle's own brute force (what tests/test_bench_iteration_parity_gpu.py compares the HIP path
with), and -- for timing the whole iteration at BASELINE's sizes -- a k-d tree (scipy.spatial.cKDTree) that yields the same
edge rule (distance < r in float64): `KdPhysicsOracle`, `distance_loss_kdtree`.  The k-d tree forms are checked against the
brute-force ones in tests/test_host_iteration.py."""
from __future__ import annotations

import math

import numpy as np
import torch

from .physics_oracle import PhysicsOracle, distance_loss_oracle


class KdPhysicsOracle(PhysicsOracle):
    """PhysicsOracle with its edge lists from a k-d tree: every ordered pair (query row, point col) with distance < r."""

    @staticmethod
    def _edges(y, x, r):
        from scipy.spatial import cKDTree
        yn, xn = y.detach().double().numpy(), x.detach().double().numpy()
        # per query the points within r, all host cores (sparse_distance_matrix is single-threaded: 5 s for config 3's
        # 200 k x 24.8 k interpolation search against 1 s)
        import itertools
        lists = cKDTree(xn).query_ball_point(yn, float(r), workers=-1, return_sorted=True)
        cnt = np.fromiter((len(l) for l in lists), np.int64, len(lists))
        col = np.fromiter(itertools.chain.from_iterable(lists), np.int64, int(cnt.sum()))
        row = np.repeat(np.arange(yn.shape[0], dtype=np.int64), cnt)
        # the tree keeps d <= r (pairs at distance 0 -- the self pairs, coincident points -- are listed too); the rule is
        # d < r on the float64 difference
        d2 = ((yn[row] - xn[col]) ** 2).sum(1)
        keep = d2 < float(r) ** 2
        row, col = row[keep], col[keep]
        return torch.from_numpy(row), torch.from_numpy(col)  # row-major, columns ascending: the brute-force form's order


def distance_loss_kdtree(positions, threshold):
    """distance_loss_oracle with the pairs from a k-d tree: same sums (float64), O(N log N)."""
    from scipy.spatial import cKDTree
    x = np.asarray(positions, dtype=np.float64)
    thr = float(np.float32(threshold))
    pairs = cKDTree(x).query_pairs(thr, output_type="ndarray")  # i < j, d <= thr
    if pairs.shape[0] == 0:
        return 0.0, np.zeros_like(x)
    i, j = pairs[:, 0], pairs[:, 1]
    diff = x[i] - x[j]
    d = np.sqrt((diff ** 2).sum(1))
    m = d < thr
    i, j, diff, d = i[m], j[m], diff[m], d[m]
    t = thr - d
    loss = 2.0 * float((t ** 2).sum())  # ordered pairs
    k = np.where(d > 0, -4.0 * t / np.where(d > 0, d, 1.0), 0.0)
    grad = np.zeros_like(x)
    np.add.at(grad, i, k[:, None] * diff)
    np.add.at(grad, j, -k[:, None] * diff)
    return loss, grad


def host_iteration(oracle, po, st, x, cams, cfg, V, distance=distance_loss_oracle, image_dtype=torch.float64):
    """d (sum over the views of the per-view loss) / d x on the host, divided by V (the batch mean); x: float64 leaf [N, 3]
    (world units).  st: the frame's frozen state, cams: per view dict(view, proj, campos, tan, gt) -- `frame_state`."""
    from fluidnexus_amd.utils.loss_utils import l1_loss, l2_loss, ssim
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
        img = torch.tensor(f["color"], dtype=image_dtype, requires_grad=True)
        gt = cam["gt"].to(image_dtype)
        gt3 = torch.cat([torch.mean(gt, dim=0, keepdim=True)] * 3, dim=0)              # tpp:356-360
        im3 = torch.cat([torch.mean(img, dim=0, keepdim=True)] * 3, dim=0)
        loss = ((1.0 - cfg["lambda_dssim"]) * l1_loss(im3, gt3) + cfg["lambda_dssim"] * (1.0 - ssim(im3, gt3))) * cfg["lambda_image"]
        dimg, = torch.autograd.grad(loss, img)
        g = oracle.backward(f, dimg.numpy().astype(np.float32))
        g_means += g["dL_dmeans3D"][:n_fluid].astype(np.float64)
    # the view-independent terms, once per view (tpp:365-404): distance loss on the rendered positions, physics terms on x
    _, gd = distance(render_xyz.detach().numpy(), cfg["distance_threshold_visual"])
    g_means += V * cfg["lambda_current_distance"] * gd
    phys = cfg["lambda_exyz"] * l2_loss(x * sf, st["estimate_xyz"])
    pr = po.gas_constraints_from_exyz_nn(x, st["imass"])
    phys = phys + cfg["lambda_gas_constraints"] * l2_loss(pr, torch.ones_like(pr))
    pn = po.gas_constraints_from_vel_nn_guess(x, st["x_prev"], st["imass"], st["buoyancy"], st["force"])
    phys = phys + cfg["lambda_next_gas_constraints"] * l2_loss(pn, torch.ones_like(pn))
    total = V * phys + (render_xyz * torch.from_numpy(g_means)).sum()   # the rasteriser's gradient enters as a cotangent
    gx, = torch.autograd.grad(total, x)
    return gx / V  # set_batch_gradient_current (gm_dynamics.py:461-472)


def frame_state(gm, cams, size):
    """The frozen state of a harness frame (fluidnexus_amd.harness.build_smoke_frame) as host arrays for host_iteration."""
    from fluidnexus_amd.renderer.pipes import _static_attributes
    cpu = lambda t: t.detach().double().cpu()  # noqa: E731
    opac, scales, rots, cols = (t.detach().float().cpu().numpy() for t in _static_attributes(gm, "guess_visual_nn", False))
    st = dict(x_prev=cpu(gm._xyz), visual_xyz=cpu(gm._visual_xyz), estimate_xyz=cpu(gm._estimate_xyz), imass=cpu(gm._imass),
              buoyancy=cpu(gm._buoyancy), force=cpu(gm._force), gs_xyz=gm._gs_xyz.detach().float().cpu().numpy(),
              opacity=opac, scales=scales, rotations=rots, colors=cols, bg=np.zeros(3, np.float32), W=size, H=size)
    hc = [dict(view=c.world_view_transform.cpu().numpy(), proj=c.full_proj_transform.cpu().numpy(),
               campos=c.camera_center.cpu().numpy(), tan=math.tan(c.FoVx * 0.5), gt=c.original_image.double().cpu())
          for c in cams]
    return st, hc
