"""Developer tool: distribution of instances per depth-rank block and per (block, tile) slice for the bench scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidnexus_amd.harness import build_smoke_frame
from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
gm, cams = build_smoke_frame(n_views=5, size=512)
gm.training_setup_current(__import__("types").SimpleNamespace(position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                                                              position_lr_delay_mult=0.01, position_lr_max_steps=30000))
fn, GRsetting, GRzer = get_render_pipe("render_dynamics")
bg = torch.zeros(3, device="cuda")
from fluidnexus_amd import _lib
from tests.hip_harness import _view
for cam in cams[:2]:
    with torch.no_grad():
        pkg = fn(cam, gm, None, bg, GRsetting=GRsetting, GRzer=GRzer, pos_type="guess_visual_nn", scale=True)
    radii = pkg["radii"].long()
    # recompute rects from the library's formula on the host side (approximate: uses screen-space means via a second pass)
    P = radii.shape[0]
    # use the autograd ctx-free path: rerun through hip_harness-like access is heavy; approximate rect by radius only
    tiles = ((2 * radii + 15) // 16 + 1).clamp(max=32) ** 2 * (radii > 0)
    means = pkg["means3D"]
    cam_z = (torch.cat([means, torch.ones(P, 1, device="cuda")], 1) @ cam.world_view_transform)[:, 2]
    order = torch.argsort(torch.where(radii > 0, cam_z, torch.full_like(cam_z, 1e9)))
    t_sorted = tiles[order]
    nb = (P + 1023) // 1024
    pad = torch.zeros(nb * 1024 - P, dtype=t_sorted.dtype, device="cuda")
    per_block = torch.cat([t_sorted, pad]).view(nb, 1024).sum(1)
    print("instances(approx)", int(tiles.sum()), "blocks", nb, "per block mean", float(per_block.float().mean()),
          "max", int(per_block.max()), "p99", float(per_block.float().quantile(0.99)), "max tiles/splat", int(tiles.max()))
    print("top blocks", sorted(per_block.tolist())[-10:])
