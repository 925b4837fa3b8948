"""Autograd wrappers of the fused particle-physics kernels (include/fnx_physics.h).

`density_ratio`  == the arithmetic of GaussianModel.get_gas_constraints_from_exyz_nn after the
                    scaling (gm_dynamics.py:1276-1292): radius_graph + poly6 + index_add_ + 2 divisions;
`visual_from_hidden` == GaussianModel.get_visual_xyz_from_nn after the scaling (gm_dynamics.py:1463-1496).
Both fuse the neighbour search; see the header for the neighbour rule; `knn_k=` selects the max_num_neighbors mode.
"""
from __future__ import annotations

import os

import torch

from ._lib import raw_stream as _lib_raw_stream

from . import _physics_lib as PL


def _stream():
    return _lib_raw_stream()


def _req(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("fluidnexus_amd physics: tensors must be on a HIP device (no CPU path)")
    t = t.float() if t.dtype != torch.float32 else t
    return t if t.is_contiguous() else t.contiguous()


class HashGrid:
    """Opaque uniform hash grid over a point cloud (cell edge = H)."""

    def __init__(self, xyz: torch.Tensor, cell: float, build: bool = True, zeroed: bool = False):
        lib = PL.physics()
        xyz = _req(xyz.detach())
        self.N = xyz.shape[0]
        self.cell = float(cell)
        # zeroed=True: storage for fnx_adam_step_grid (its contract: bucket counts zero on entry, left zero)
        self.blob = (torch.zeros if zeroed else torch.empty)(lib.fnx_grid_bytes(self.N), dtype=torch.uint8, device=xyz.device)
        self.velocity_of = None  # (hidden_prev.data_ptr(), secs) whose per-slot velocities the payload holds, if any
        if build:  # build=False: storage only, a fused entry point fills it (fnx_physical_stage)
            PL.check(lib.fnx_grid_build(xyz.data_ptr() if self.N else None, self.N, self.cell, self.blob.data_ptr(),
                                        _stream()))
        self._items = None

    def cell_items(self, refresh: bool = False) -> torch.Tensor:
        """Per-cell work items of the (built) grid for the cell-by-cell kernels; made on first use, and again with
        `refresh` (a grid object that is rebuilt in place over moved points keeps its buffers)."""
        if self._items is None or refresh:
            lib = PL.physics()
            if self._items is None:
                self._items = torch.empty(lib.fnx_grid_cell_items_bytes(self.N), dtype=torch.uint8, device=self.blob.device)
            PL.check(lib.fnx_grid_cell_items(self.blob.data_ptr(), self.N, self._items.data_ptr(), _stream()))
        return self._items


class _DensityRatio(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, imass, H, p0, grid):
        lib = PL.physics()
        xyz, imass = _req(xyz), _req(imass)
        N = xyz.shape[0]
        if grid is None:
            grid = HashGrid(xyz, H)
        out = torch.empty(N, 1, dtype=torch.float32, device=xyz.device)
        PL.check(lib.fnx_density_forward(xyz.data_ptr(), N, imass.data_ptr(), H, p0, grid.blob.data_ptr(),
                                         out.data_ptr(), _stream()))
        ctx.save_for_backward(xyz, imass, grid.blob)
        ctx.consts = (H, p0)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = PL.physics()
        xyz, imass, blob = ctx.saved_tensors
        H, p0 = ctx.consts
        N = xyz.shape[0]
        g = _req(g)
        dx = torch.empty_like(xyz)
        PL.check(lib.fnx_density_backward(xyz.data_ptr(), N, imass.data_ptr(), H, p0, blob.data_ptr(), g.data_ptr(),
                                          dx.data_ptr(), _stream()))
        return dx, None, None, None, None


def knn_cut(queries, points_grid, H, K):
    """int32 [Nq] holding uint32 bit patterns: the K-th smallest index among the points of `points_grid` within H of each
    query, 0xFFFFFFFF (-1) where a query has at most K of them -- torch_cluster's max_num_neighbors rule on CUDA (the
    first K hits in index order; include/fnx_physics.h, fnx_knn_cut)."""
    lib = PL.physics()
    queries = _req(queries.detach())
    cut = torch.empty(queries.shape[0], dtype=torch.int32, device=queries.device)
    PL.check(lib.fnx_knn_cut(queries.data_ptr() if queries.shape[0] else None, queries.shape[0], points_grid.N, float(H),
                             int(K), points_grid.blob.data_ptr(), cut.data_ptr(), _stream()))
    return cut


class _DensityRatioCapped(torch.autograd.Function):
    """_DensityRatio on the edge set radius_graph(loop=True, max_num_neighbors=K) keeps (gm_dynamics.py:1276-1290)."""

    @staticmethod
    def forward(ctx, xyz, imass, H, p0, grid, K):
        lib = PL.physics()
        xyz, imass = _req(xyz), _req(imass)
        N = xyz.shape[0]
        if grid is None:
            grid = HashGrid(xyz, H)
        cut = knn_cut(xyz, grid, H, K)
        out = torch.empty(N, 1, dtype=torch.float32, device=xyz.device)
        PL.check(lib.fnx_density_forward_kcap(xyz.data_ptr(), N, imass.data_ptr(), H, p0, grid.blob.data_ptr(),
                                              cut.data_ptr(), out.data_ptr(), _stream()))
        ctx.save_for_backward(xyz, imass, grid.blob, cut)
        ctx.consts = (H, p0)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = PL.physics()
        xyz, imass, blob, cut = ctx.saved_tensors
        H, p0 = ctx.consts
        g = _req(g)
        dx = torch.empty_like(xyz)
        PL.check(lib.fnx_density_backward_kcap(xyz.data_ptr(), xyz.shape[0], imass.data_ptr(), H, p0, blob.data_ptr(),
                                               cut.data_ptr(), g.data_ptr(), dx.data_ptr(), _stream()))
        return dx, None, None, None, None, None


def density_ratio(xyz, imass, H, p0, grid=None, knn_k=None):
    """p_ratio [N,1] of positions xyz [N,3] (scaled units), inverse masses imass [N,1].
    `grid`: an up-to-date HashGrid over xyz (cell = H) to reuse; built here when None.
    `knn_k`: reproduce max_num_neighbors = knn_k (per-particle kernels, not the fused stage); None = all pairs."""
    if knn_k is not None:
        return _DensityRatioCapped.apply(xyz, imass, float(H), float(p0), grid, int(knn_k))
    return _DensityRatio.apply(xyz, imass, float(H), float(p0), grid)


class _VisualFromHidden(torch.autograd.Function):
    @staticmethod
    def forward(ctx, visual, hidden, hidden_prev, H, secs, eps, visual_grid, hgrid, memo):
        lib = PL.physics()
        visual, hidden, hidden_prev = _req(visual), _req(hidden), _req(hidden_prev)
        V, N = visual.shape[0], hidden.shape[0]
        if memo is not None and "out" in memo:
            # same inputs as an earlier call (caller's guarantee): reuse the forward results
            out, sum_w, wvel = memo["out"], memo["sum_w"], memo["wvel"]
        else:
            if hgrid is None:
                hgrid = HashGrid(hidden, H)
            out = torch.empty_like(visual)
            sum_w = torch.empty(V, dtype=torch.float32, device=visual.device)
            wvel = torch.empty(V, 3, dtype=torch.float32, device=visual.device)
            if visual_grid is None:
                visual_grid = HashGrid(visual, H)
            # memo["out_div"] = (tensor [V,3], divisor): the caller also wants out / divisor (the rasteriser's units),
            # written by the same kernel; memo["out_div_done"] tells it that it was
            div = memo.get("out_div") if memo is not None else None
            # the fused optimiser step (fnx_adam_step_grid) leaves the per-slot velocities next to the grid it builds
            ready = hgrid.velocity_of == (hidden_prev.data_ptr(), float(secs))
            PL.check(lib.fnx_visual_interp_forward_cells_vel(
                visual.data_ptr(), V, hidden.data_ptr(), hidden_prev.data_ptr(), N, H, secs, eps, hgrid.blob.data_ptr(),
                visual_grid.blob.data_ptr(), visual_grid.cell_items().data_ptr(), out.data_ptr(), sum_w.data_ptr(),
                wvel.data_ptr(), div[0].data_ptr() if div is not None else None, float(div[1]) if div is not None else 1.0,
                1 if ready else 0, _stream()))
            hgrid.velocity_of = (hidden_prev.data_ptr(), float(secs))  # slot_velocity has run for this grid now
            if memo is not None:
                memo.update(out=out, sum_w=sum_w, wvel=wvel, out_div_done=div is not None)
        if visual_grid is None:
            visual_grid = HashGrid(visual, H)
        ctx.save_for_backward(visual, hidden, hidden_prev, sum_w, wvel, visual_grid.blob)
        ctx.consts = (H, secs, eps)
        ctx.memo = memo
        if memo is not None:
            if hgrid is not None:
                memo["hgrid"] = hgrid  # the backward walks the hidden particles cell by cell over this grid
            if memo.get("defer"):
                memo["saved"] = (visual, hidden, hidden_prev, sum_w, wvel, visual_grid.blob, (H, secs, eps))
        ctx.hgrid = memo.get("hgrid") if memo is not None else hgrid
        return out

    @staticmethod
    def backward(ctx, g):
        lib = PL.physics()
        visual, hidden, hidden_prev, sum_w, wvel, vblob = ctx.saved_tensors
        H, secs, eps = ctx.consts
        g = _req(g)
        memo = ctx.memo
        if memo is not None and memo.get("defer"):
            # the map g -> dL/dhidden is linear: sum the upstream gradients of all views that share this
            # forward and run the kernel once (flush_deferred_visual_backward)
            # (kept as a list: the views may run on different streams; they are summed after the join)
            memo.setdefault("g_list", []).append(g)
            return None, None, None, None, None, None, None, None, None
        dh = _visual_backward(visual, hidden, hidden_prev, H, secs, eps, vblob, ctx.hgrid, None, sum_w, wvel, g)
        return None, dh, None, None, None, None, None, None, None


def _visual_backward(visual, hidden, hidden_prev, H, secs, eps, vblob, hgrid, hitems, sum_w, wvel, g, extra=None):
    """dL/dhidden [N,3]: cell by cell over the hidden grid when one is at hand (`hitems`: its work items, built here
    when None), else a wave per hidden particle.  `extra` = (g2, scale2): the upstream gradient is g + scale2 * g2."""
    lib = PL.physics()
    dh = torch.empty_like(hidden)
    if hgrid is not None and hgrid.N == hidden.shape[0]:
        if hitems is None:
            hitems = hgrid.cell_items(refresh=True)
        g2 = _req(extra[0]) if extra is not None else None
        PL.check(lib.fnx_visual_interp_backward_cells_sum(
            visual.data_ptr(), visual.shape[0], hidden.data_ptr(), hidden_prev.data_ptr(), hidden.shape[0], H, secs, eps,
            vblob.data_ptr(), hgrid.blob.data_ptr(), hitems.data_ptr(), sum_w.data_ptr(), wvel.data_ptr(), g.data_ptr(),
            g2.data_ptr() if g2 is not None else None, float(extra[1]) if extra is not None else 0.0,
            dh.data_ptr(), _stream()))
    else:
        if extra is not None:
            g = g + extra[0] * float(extra[1])
        PL.check(lib.fnx_visual_interp_backward(visual.data_ptr(), visual.shape[0], hidden.data_ptr(),
                                                hidden_prev.data_ptr(), hidden.shape[0], H, secs, eps, vblob.data_ptr(),
                                                sum_w.data_ptr(), wvel.data_ptr(), g.data_ptr(), dh.data_ptr(),
                                                _stream()))
    return dh


def flush_deferred_visual_backward(memo):
    """dL/dhidden [N,3] for the gradients accumulated in `memo` by deferred backward calls (or None)."""
    if memo is None or not memo.get("g_list"):
        return None
    lib = PL.physics()
    visual, hidden, hidden_prev, sum_w, wvel, vblob, (H, secs, eps) = memo["saved"]
    gl = memo.pop("g_list")
    g = gl[0] if len(gl) == 1 else torch.stack(gl).sum(dim=0)
    g = g.contiguous()
    return _visual_backward(visual, hidden, hidden_prev, H, secs, eps, vblob, memo.get("hgrid"), memo.get("hitems"),
                            sum_w, wvel, g, memo.pop("g_extra", None))


class _VisualFromHiddenCapped(torch.autograd.Function):
    """_VisualFromHidden where a visual particle keeps max_num_neighbors = K hidden particles (gm_dynamics.py:1463-1468);
    no memo, no deferral: the mode exists for parity with capped reference runs."""

    @staticmethod
    def forward(ctx, visual, hidden, hidden_prev, H, secs, eps, visual_grid, hgrid, K):
        lib = PL.physics()
        visual, hidden, hidden_prev = _req(visual), _req(hidden), _req(hidden_prev)
        V, N = visual.shape[0], hidden.shape[0]
        if hgrid is None:
            hgrid = HashGrid(hidden, H)
        if visual_grid is None:
            visual_grid = HashGrid(visual, H)
        cutv = knn_cut(visual, hgrid, H, K)
        out = torch.empty_like(visual)
        sum_w = torch.empty(V, dtype=torch.float32, device=visual.device)
        wvel = torch.empty(V, 3, dtype=torch.float32, device=visual.device)
        PL.check(lib.fnx_visual_interp_forward_kcap(visual.data_ptr(), V, hidden.data_ptr(), hidden_prev.data_ptr(), N, H,
                                                    secs, eps, hgrid.blob.data_ptr(), cutv.data_ptr(), out.data_ptr(),
                                                    sum_w.data_ptr(), wvel.data_ptr(), _stream()))
        hgrid.velocity_of = (hidden_prev.data_ptr(), float(secs))
        ctx.save_for_backward(visual, hidden, hidden_prev, sum_w, wvel, visual_grid.blob, cutv)
        ctx.consts = (H, secs, eps)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = PL.physics()
        visual, hidden, hidden_prev, sum_w, wvel, vblob, cutv = ctx.saved_tensors
        H, secs, eps = ctx.consts
        g = _req(g)
        dh = torch.empty_like(hidden)
        PL.check(lib.fnx_visual_interp_backward_kcap(visual.data_ptr(), visual.shape[0], hidden.data_ptr(),
                                                     hidden_prev.data_ptr(), hidden.shape[0], H, secs, eps, vblob.data_ptr(),
                                                     cutv.data_ptr(), sum_w.data_ptr(), wvel.data_ptr(), g.data_ptr(),
                                                     dh.data_ptr(), _stream()))
        return None, dh, None, None, None, None, None, None, None


def visual_from_hidden(visual, hidden, hidden_prev, H, secs, eps=1e-8, visual_grid=None, hidden_grid=None,
                       memo=None, share_output=False, knn_k=None):
    """visual [V,3] (constant), hidden [N,3] (differentiable), hidden_prev [N,3] -> advected visual [V,3].
    `visual_grid`: a HashGrid over `visual` to reuse across iterations (visual is fixed within a frame);
    `hidden_grid`: an up-to-date HashGrid over `hidden`; `memo`: a dict the caller keeps for as long as
    the inputs are unchanged -- the forward kernel then runs once and later calls only add an autograd
    node (the views of one iteration all see the same particle state)."""
    if knn_k is not None:  # max_num_neighbors = knn_k: per-particle kernels, evaluated on every call
        return _VisualFromHiddenCapped.apply(visual, hidden, hidden_prev, float(H), float(secs), float(eps), visual_grid,
                                             hidden_grid, int(knn_k))
    out = _VisualFromHidden.apply(visual, hidden, hidden_prev, float(H), float(secs), float(eps), visual_grid,
                                  hidden_grid, memo)
    # memoised results are handed out as copies unless the caller promises not to modify them in place
    return out.clone() if memo is not None and not share_output else out



class _PhysicalStageLoss(torch.autograd.Function):
    """lambda_exyz * l2(x_nn * sf, x_est) + lambda_gas * l2(p_ratio(x), 1) + lambda_next * l2(p_ratio(x'), 1)
    (train_physical_particle.py:368-404) as ONE autograd node on top of fnx_physical_stage: value and
    analytic gradient (density kernel backward + the affine Jacobian of the one-tick advection,
    gm_dynamics.py:1014-1030) come out of one launch sequence of ~12 kernels."""

    @staticmethod
    def run(x_nn, gm, lam_e, lam_g, lam_n):
        """The launch sequence: returns (terms [3], loss [], grad [N,3]) -- views of one device buffer."""
        lib = PL.physics()
        if getattr(gm, "knn_cap", False):
            raise RuntimeError("the fused physical stage takes all pairs within H; with gm.knn_cap (max_num_neighbors mode) "
                               "use the per-term methods (HotLoop(fused_physics=False))")
        x = _req(x_nn.detach())
        N = x.shape[0]
        dev = x.device
        est, build_est = gm._grid_slot("est", x)
        guess, _ = gm._grid_slot("guess", x)
        out = torch.empty(4 + 3 * N, dtype=torch.float32, device=dev)  # terms[3] | loss | grad[N,3]
        scratch = torch.empty(15 * N + 64, dtype=torch.float32, device=dev)
        base = out.data_ptr()
        PL.check(lib.fnx_physical_stage(
            x.data_ptr(), N, float(gm.scale_factor), _req(gm._estimate_xyz).data_ptr(), _req(gm._xyz).data_ptr(),
            _req(gm._imass).data_ptr(), _req(gm._buoyancy).data_ptr(), _req(gm._force).data_ptr(),
            float(gm.buoyancy_max_y), float(gm.H), float(gm.p0), float(gm._secs), lam_e, lam_g, lam_n,
            est.blob.data_ptr(), int(build_est), guess.blob.data_ptr(), scratch.data_ptr(), base, base + 12, base + 16,
            _stream()))
        return out[:3], out[3].view(()), out[4:].view(N, 3)

    @staticmethod
    def forward(ctx, x_nn, gm, lam_e, lam_g, lam_n, memo):
        if memo is not None and "loss" in memo:  # same particle state as an earlier call of this iteration
            ctx.grad = memo["grad"]
            return memo["loss"].clone()
        terms, loss, grad = _PhysicalStageLoss.run(x_nn, gm, lam_e, lam_g, lam_n)
        ctx.grad = grad
        if memo is not None:
            memo.update(loss=loss, grad=grad, terms=terms)
            return loss.clone()
        return loss

    @staticmethod
    def backward(ctx, g):
        return ctx.grad * g, None, None, None, None, None


def physical_stage_value_and_grad(gm, lam_exyz, lam_gas, lam_next, memo=None):
    """(loss, d loss / d x_nn [N,3]) of physical_stage_loss without going through autograd: the hot loop adds the
    gradient to its batch gradient itself, which saves the scalar clone, the ones seed and the grad * 1 kernels."""
    if memo is not None and "loss" in memo:
        return memo["loss"], memo["grad"]
    terms, loss, grad = _PhysicalStageLoss.run(gm._estimate_xyz_nn, gm, float(lam_exyz), float(lam_gas), float(lam_next))
    if memo is not None:
        memo.update(loss=loss, grad=grad, terms=terms)
    return loss, grad


def physical_stage_loss(gm, lam_exyz, lam_gas, lam_next, memo=None):
    """Weighted physics terms of the physical-particle stage for GaussianModel `gm` (one autograd node).
    `memo`: dict kept by the caller while the particle state is unchanged; the terms are view-independent,
    so the views of one iteration share one evaluation (value and gradient are reused bit for bit)."""
    return _PhysicalStageLoss.apply(gm._estimate_xyz_nn, gm, float(lam_exyz), float(lam_gas), float(lam_next), memo)


_DIST_MEMO = [None]  # (state key, threshold, loss, grad) of the latest tagged evaluation


class _DistanceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, positions, threshold):
        # The reference evaluates the term once per VIEW on the same rendered positions (train_physical_particle.py:365-366).
        # The automated per-view pipe tags its positions with the particle state they were computed from (`_fnx_state`):
        # a second evaluation for the same state and threshold reuses value and gradient instead of searching again.
        key = getattr(positions, "_fnx_state", None)
        hit = _DIST_MEMO[0]
        if key is not None and hit is not None and hit[0] == key and hit[1] == float(threshold) and hit[3].shape == positions.shape:
            loss, grad = hit[2], hit[3]
        else:
            loss, grad = distance_loss_value_and_grad(positions.detach(), threshold)
            if key is not None:
                _DIST_MEMO[0] = (key, float(threshold), loss, grad)
        ctx.save_for_backward(grad)
        return loss.clone() if key is not None else loss

    @staticmethod
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        return grad * g, None


_DIST_LISTS = os.environ.get("FNX_DIST_GRID", "0") != "1"  # FNX_DIST_GRID=1: the counted / scanned / filled grid version
_DIST_FORCED = "FNX_DIST_GRID" in os.environ
# Verlet pair lists on top of the linked-list form (fnx_distance_loss_verlet): a call whose lists are still valid touches
# no hash table.  OPT-IN (FNX_DIST_VERLET=1 / set_distance_verlet): exact and graph-capturable (tests/test_distance_verlet_gpu.py),
# but measured SLOWER inside the config-3 loop (round 5, A/B in one gpurun call: 959 / 958 against 970 / 968 it/s): alone
# a valid-lists call costs 36 us of kernels against 65, but a rebuild (27 cells) costs 115, the loop's first ~100
# iterations rebuild every 4th call (the cloud drifts by lr per step) and 5 % of the later ones do (fringe particles that
# jump), and what the branch costs the iteration is mostly being there beside emit, not its own traffic.
# skin = threshold is the widest the 27-cell rebuild supports.
_DIST_VERLET = os.environ.get("FNX_DIST_VERLET", "0") == "1"
DIST_VERLET_K = 16  # list slots per point (config 3's plume: 2.7 neighbours on average within 2 thresholds)


def prefer_distance_lists(enabled: bool):
    """Which implementation distance_loss_value_and_grad uses (both exact, tests/test_physics_gpu.py): the linked-list one
    (two launches, one thread per point: few waves next to a long blend forward -- config 3, five views: +14 it/s) or the
    counted grid (five launches, eight lanes per point: shorter end to end, which wins when the branch itself is the
    critical path -- one or two views per rank: 917 against 882 it/s on a rank's share of config 5).  The loops choose by
    their view count; FNX_DIST_GRID in the environment pins it."""
    global _DIST_LISTS
    if not _DIST_FORCED:
        _DIST_LISTS = bool(enabled)


def set_distance_verlet(enabled: bool):
    """Verlet pair lists for the linked-list form (default off, see _DIST_VERLET)."""
    global _DIST_VERLET
    _DIST_VERLET = bool(enabled)


class _DistanceBuffers:
    """Persistent, zero-filled-once device state of the linked-list / Verlet distance loss, one entry per (device, stream,
    N): the stamped bucket table and the pair-list state.  The kernels receive RAW pointers into these tensors, and a
    captured hipGraph keeps such a pointer without a tensor reference: entries handed out while a stream is
    capturing are pinned until release_captured(); otherwise the least recently used entries go once more than `keep`
    exist (a sequence run changes N every frame)."""

    def __init__(self, keep=4):
        self.keep, self.entries, self.pinned, self.tick = keep, {}, set(), 0

    def get(self, dev, N, K):
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, int(N), int(K))
        capturing = torch.cuda.is_current_stream_capturing()
        e = self.entries.get(key)
        if e is None:
            if capturing:
                return None  # never allocate inside a capture: the warm-up iterations created the entry
            lib = PL.physics()
            e = self.entries[key] = dict(
                table=torch.zeros(lib.fnx_distance_table_bytes(int(N)), dtype=torch.uint8, device=dev),
                state=torch.zeros(lib.fnx_distance_verlet_bytes(int(N), int(K)), dtype=torch.uint8, device=dev), used=0)
            for k in sorted((k for k in self.entries if k not in self.pinned and k != key),
                            key=lambda k: self.entries[k]["used"])[:max(0, len(self.entries) - len(self.pinned) - self.keep)]:
                del self.entries[k]
        self.tick += 1
        e["used"] = self.tick
        if capturing:
            self.pinned.add(key)
        return e

    def release_captured(self):
        self.pinned.clear()

    def counters(self):
        """[(N, valid, calls, rebuilds, overflowed points)] of every live pair-list state (host read)."""
        out = []
        for (_, _, N, _), e in self.entries.items():
            st = e["state"]
            off = (-st.data_ptr()) % 256
            h = st[off:off + 64].view(torch.int32).cpu().tolist()
            out.append((N, h[0], h[2], h[3], h[4]))
        return out


_DIST_BUFFERS = _DistanceBuffers()


def distance_verlet_counters():
    return _DIST_BUFFERS.counters()


def release_captured_distance_state():
    """Un-pin the distance-loss buffers a destroyed hipGraph referred to."""
    _DIST_BUFFERS.release_captured()


def distance_loss_value_and_grad(positions, threshold, need_grad=True, skin=None):
    """(loss, d loss / d positions [N,3]) of utils/loss_utils.distance_loss(positions, threshold)
    (loss_utils.py:98-121: every pair closer than `threshold` pays (threshold - distance)^2, counted in both
    orders) on a hash grid with cell = threshold instead of the reference's dense N x N torch.cdist -- the same
    sum, but O(N) memory, so it stays usable at 10^5 particles.  `skin`: margin of the Verlet pair lists (default:
    `threshold`, the widest supported)."""
    lib = PL.physics()
    x = _req(positions.detach())
    if x.dim() != 2 or x.shape[1] != 3:
        raise RuntimeError("positions must have dimensions (num_points, 3)")
    N = x.shape[0]
    if N == 0:
        z = torch.zeros((), dtype=torch.float32, device=x.device)
        return z, torch.zeros_like(x)
    grad = torch.empty_like(x) if need_grad else None
    buf = _DIST_BUFFERS.get(x.device, N, DIST_VERLET_K) if _DIST_LISTS else None
    if buf is not None:  # one thread per point (csrc/physics.hip fnx_distance_loss_lists / _verlet)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        if _DIST_VERLET:
            PL.check(lib.fnx_distance_loss_verlet(x.data_ptr(), N, float(threshold), float(threshold if skin is None else skin),
                                                  buf["table"].data_ptr(), buf["state"].data_ptr(), DIST_VERLET_K,
                                                  grad.data_ptr() if need_grad else None, loss.data_ptr(), _stream()))
        else:
            PL.check(lib.fnx_distance_loss_lists(x.data_ptr(), N, float(threshold), buf["table"].data_ptr(),
                                                 grad.data_ptr() if need_grad else None, loss.data_ptr(), _stream()))
        return loss[0], grad
    grid = torch.empty(lib.fnx_grid_bytes(N), dtype=torch.uint8, device=x.device)
    partials = torch.empty(lib.fnx_distance_loss_partials(N), dtype=torch.float32, device=x.device)
    PL.check(lib.fnx_distance_loss(x.data_ptr(), N, float(threshold), grid.data_ptr(), partials.data_ptr(),
                                   grad.data_ptr() if need_grad else None, _stream()))
    return partials.sum(), grad


def distance_loss(positions, threshold):
    """Differentiable drop-in for utils/loss_utils.distance_loss on device tensors (one autograd node)."""
    return _DistanceLoss.apply(positions, float(threshold))


def adam_step(param, optimizer, terms, batch_size, grad_out=None, scaled_out=None, scale=1.0, grid=None, prev=None,
              secs=None):
    """Gradient mean + Adam step of `param` in one kernel (fnx_adam_step), on the state of `optimizer`
    (a torch.optim.Adam with amsgrad = False, weight_decay = 0; the hyper-parameters are those of the group that
    holds `param`).
    terms: up to three (tensor, scale) pairs; the gradient is sum(tensor * scale) / batch_size.
    scaled_out: optional tensor like `param` that receives the updated param * scale.
    grid: a HashGrid(..., build=False, zeroed=True) over [N,3] points: the step also builds it over scaled_out
    (fnx_adam_step_grid: two launches for step + grid build); prev / secs: leave the per-slot velocities
    (scaled_out - prev) / secs in the grid for the hidden -> visual interpolation."""
    lib = PL.physics()
    group = next((g for g in optimizer.param_groups if any(q is param for q in g["params"])), None)
    if group is None:
        raise RuntimeError("adam_step: the parameter does not belong to the optimiser")
    if group.get("amsgrad") or group.get("weight_decay") or group.get("maximize"):
        raise RuntimeError("adam_step: amsgrad / weight_decay / maximize are not supported")
    st = optimizer.state[param]
    if len(st) == 0:  # same lazy initialisation as torch.optim.Adam._init_group (capturable layout)
        st["step"] = torch.zeros((), dtype=torch.float32, device=param.device)
        st["exp_avg"] = torch.zeros_like(param, memory_format=torch.preserve_format)
        st["exp_avg_sq"] = torch.zeros_like(param, memory_format=torch.preserve_format)
    if not st["step"].is_cuda:
        raise RuntimeError("adam_step needs the optimiser state on the device (capturable=True)")
    if "fnx_arrived" not in st:  # arrival counter of the kernel's "last workgroup advances the step count" protocol
        st["fnx_arrived"] = torch.zeros(1, dtype=torch.int32, device=param.device)
    terms = [(t, float(sc)) for t, sc in terms if t is not None]
    if not 1 <= len(terms) <= 3:
        raise RuntimeError("adam_step takes one to three gradient terms")
    ts = [_req(t.detach()) for t, _ in terms] + [None] * (3 - len(terms))
    sc = [sc for _, sc in terms] + [0.0] * (3 - len(terms))
    b1, b2 = group["betas"]
    x = param.data
    assert x.is_contiguous() and x.dtype == torch.float32
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    if grid is not None:
        if x.dim() != 2 or x.shape[1] != 3 or grid.N != x.shape[0] or scaled_out is None or x.shape[0] == 0:
            raise RuntimeError("adam_step(grid=...): param must be [N,3] with N = grid.N > 0, and scaled_out given")
        if prev is not None:
            prev = _req(prev.detach())
        PL.check(lib.fnx_adam_step_grid(x.data_ptr(), x.shape[0], ptr(ts[0]), sc[0], ptr(ts[1]), sc[1], ptr(ts[2]), sc[2],
                                        1.0 / float(batch_size), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                        st["step"].data_ptr(), float(group["lr"]), float(b1), float(b2),
                                        float(group["eps"]), ptr(grad_out), ptr(scaled_out), float(scale),
                                        st["fnx_arrived"].data_ptr(), float(grid.cell), grid.blob.data_ptr(), ptr(prev),
                                        float(secs) if secs is not None else 1.0, _stream()))
        grid.velocity_of = (prev.data_ptr(), float(secs)) if prev is not None else None
        _raw_write_done(param)
        return
    PL.check(lib.fnx_adam_step(x.data_ptr(), x.numel(), ptr(ts[0]), sc[0], ptr(ts[1]), sc[1], ptr(ts[2]), sc[2],
                               1.0 / float(batch_size), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                               st["step"].data_ptr(), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                               ptr(grad_out), ptr(scaled_out), float(scale), st["fnx_arrived"].data_ptr(),
                               _stream()))
    _raw_write_done(param)


def _raw_write_done(param):
    """The library's optimiser steps write `param.data` through raw pointers: autograd's version counter -- what every
    per-state memo in this package is keyed on -- does not move by itself (the distance-loss memo then served
    the previous iteration's value and gradient).  Bump it like an in-place torch op would, and drop the module-level memo."""
    _DIST_MEMO[0] = None
    try:
        torch._C._autograd._unsafe_set_version_counter((param,), (param._version + 1,))
    except Exception:  # an older torch without the hook: the memos are cleared explicitly above / by invalidate_caches()
        pass


def knn_mean_dist2(points):
    """simple_knn._C.distCUDA2 (submodules/simple-knn/simple_knn.cu): mean squared distance of every point of
    `points` [N,3] to its 3 nearest other points, fp32 [N].  One host sync to size the search grid from the
    bounding box (the reference copies min / max to the host as well, simple_knn.cu:175-180)."""
    lib = PL.physics()
    pts = _req(points.detach())
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    N = pts.shape[0]
    out = torch.empty(N, dtype=torch.float32, device=pts.device)
    if N == 0:
        return out
    ext = (pts.max(dim=0).values - pts.min(dim=0).values).clamp_min(1e-12).double().cpu()
    vol = float(ext[0] * ext[1] * ext[2])
    cell = max((vol / N) ** (1.0 / 3.0) * 1.5, 1e-6 * float(ext.max()), 1e-30)
    grid = HashGrid(pts, cell, build=False)
    PL.check(lib.fnx_knn_mean_dist2(pts.data_ptr(), N, float(cell), grid.blob.data_ptr(), out.data_ptr(), _stream()))
    return out
