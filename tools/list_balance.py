"""How evenly the per-block lists of the blend kernels fill their lanes (bench scene, forward of all views):
wave-iterations of the present layout (wave = quadrant, four lists per wave, 256-entry batches) against a layout in
which one wave walks all 16 lists of a tile over 64-entry sub-batches.  Run on the GPU box."""
import sys

import torch

sys.path.insert(0, ".")
from fluidnexus_amd import _lib, harness as Hn  # noqa: E402
from fluidnexus_amd.renderer.pipes import render_dynamics_views, render_fluid_views  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
SIZE = 512
if cfg == 2:
    gm, cams = Hn.build_scalar_real_frame(100_000, n_views=5, size=SIZE, seed=0, device=dev)
    loop = Hn.FirstFrameLoop(gm, cams, rd_pipe="render_fluid", cfg=dict(Hn.SCALAR_REAL), capturable=False)
    pkg = render_fluid_views(cams, gm, None, loop.background, GRsetting=loop.GRsetting, GRzer=loop.GRzer, pos_type="visual",
                             scale=False, means3D=gm._visual_xyz.detach().clone().requires_grad_())
else:
    gm, cams = Hn.build_smoke_frame(200_000, 100_000, (20, 62, 20), n_views=5, size=SIZE, seed=0, device=dev)
    loop = Hn.HotLoop(gm, cams, cfg=dict(Hn.SMOKE), batched_views=True, capturable=False, fused_physics=True,
                      defer_visual_backward=True, image_loss="fused")
    means3D = gm.render_means_from_nn()
    pkg = render_dynamics_views(cams, gm, None, loop.background, GRsetting=loop.GRsetting, GRzer=loop.GRzer,
                                pos_type="guess_visual_nn", scale=True, means3D=means3D)
im = pkg["render"]
saved = im.grad_fn.saved_tensors
binning, img = saved[-2], saved[-1]
torch.cuda.synchronize()
V, H, W = im.shape[0], SIZE, SIZE
gx = gy = SIZE // 16
T = gx * gy
lib = _lib.raster()
IL = _lib.image_layout(W, H)
ib = lib.fnx_image_bytes(W, H)
cap = binning.numel() // V


def blob(t, off, n, dt):
    return t[off:off + n * dt.itemsize].view(dt)


fn = im.grad_fn
capacity = int(fn.capacity)
sb = getattr(fn, "static_bin", None)
BL = _lib.binning_layout(capacity, None if sb is None else int(sb.R_cap))
k_of = torch.empty(16, dtype=torch.long)
for by in range(4):
    for bx in range(4):
        k_of[by * 4 + bx] = 4 * ((by >> 1) * 2 + (bx >> 1)) + ((by & 1) * 2 + (bx & 1))
k_of = k_of.to(dev)
tot = dict(entries=0, block_entries=0, present=0, present_wg=0, new64=0, new128=0, new256=0)
GROUP = 4
for v in range(V):
    iv, bv = img[v * ib:(v + 1) * ib], binning[v * cap:(v + 1) * cap]
    ranges = blob(iv, IL.ranges, 2 * T, torch.int32).long().view(T, 2)
    ncon = blob(iv, IL.n_contrib, H * W, torch.int32).long().view(H, W)
    blk = ncon.view(gy, 4, 4, gx, 4, 4).permute(0, 3, 1, 4, 2, 5).amax(dim=(4, 5)).reshape(T, 16)  # [tile, by * 4 + bx]
    m = torch.zeros(T, 16, dtype=torch.long, device=dev)
    m[:, k_of] = blk  # bit 4 q + b <-> quadrant q = (q & 1, q >> 1), block b = (b & 1, b >> 1)
    r0, r1 = ranges[:, 0], ranges[:, 1]
    used = torch.minimum((r1 - r0).clamp_min(0), m.amax(dim=1))
    tile = torch.repeat_interleave(torch.arange(T, device=dev), used)
    start = torch.cumsum(used, 0) - used
    q = torch.arange(int(used.sum()), device=dev) - start[tile]
    masks = blob(bv, BL.block_masks, (min(int(BL.total), bv.numel()) - int(BL.block_masks)) // 2, torch.int16).long() & 0xFFFF
    mk = masks[r0[tile] + q]
    bits = ((mk[:, None] >> torch.arange(16, device=dev)[None, :]) & 1).bool() & (q[:, None] < m[tile])
    tot["entries"] += int(used.sum())
    tot["block_entries"] += int(bits.sum())
    nb = int(used.max().item() + 255) // 256 + 1

    def iters(sub, lists_per_wave, whole_wg=False):
        per = 256 // sub
        idx = (tile * (nb * per) + q // sub)
        cnt = torch.zeros(T * nb * per, 16, dtype=torch.long, device=dev)
        cnt.index_add_(0, idx, bits.long())
        g = cnt.view(-1, 16 // lists_per_wave, lists_per_wave).amax(dim=2)  # per wave: its longest list
        g = (g + GROUP - 1) // GROUP * GROUP
        if whole_wg:  # every wave of the workgroup stays until the slowest one reaches the batch's barrier
            return int(g.amax(dim=1).sum()) * g.shape[1]
        return int(g.sum())

    tot["present"] += iters(256, 4)  # wave-iterations of 64 pixel-entries
    tot["present_wg"] += iters(256, 4, whole_wg=True)
    tot["new64"] += iters(64, 16)    # wave-iterations of 256 pixel-entries
    tot["new128"] += iters(128, 16)
    tot["new256"] += iters(256, 16)
print(tot)
be = tot["block_entries"]
print("present: lane efficiency", be / (tot["present"] * 4.0), " wave-slot efficiency with the per-batch barrier:",
      tot["present"] / tot["present_wg"])
for k in ("new64", "new128", "new256"):
    print(k, "lane efficiency", be / (tot[k] * 16.0), " wave-iterations x4 vs present:", tot[k] * 4.0 / tot["present"])
