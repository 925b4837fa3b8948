"""Developer probe (round 5): how far the rendered visual positions move between iterations of the config-3 loop, i.e. how
often a Verlet pair list of the distance loss with a given skin would have to be rebuilt.  usage: python tools/dist_disp_probe.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidnexus_amd import rasterizer, harness as Hn

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
gm, cams = Hn.build_smoke_frame()
loop = Hn.HotLoop(gm, cams, image_loss="fused", fused_physics=True, defer_visual_backward=True, capturable=True,
                  batched_views=True, fused_step=True)
rasterizer.set_host_sync(False)
rasterizer.set_blend_math("fast")
loop.make_targets()
limits = [0.00025, 0.0005, 0.001, 0.002]
ref = {L: None for L in limits}
rebuilds = {L: [] for L in limits}
steps = []
prev = None
for it in range(iters):
    loop.iteration()
    with torch.no_grad():
        x = (gm.get_visual_xyz_from_nn() / gm.scale_factor).clone()
    if prev is not None:
        steps.append((x - prev).norm(dim=1).max().item())
    prev = x
    for L in limits:
        if ref[L] is None or (x - ref[L]).norm(dim=1).max().item() > L:
            ref[L] = x
            rebuilds[L].append(it)
torch.cuda.synchronize()
import numpy as np
s = np.array(steps)
print("max per-iteration displacement (world units): first 10", np.round(s[:10], 6), "median", np.median(s), "p99", np.percentile(s, 99))
for L in limits:
    r = rebuilds[L]
    late = [i for i in r if i >= 100]
    print(f"limit {L}: {len(r)} rebuilds in {iters} iterations, {len(late)} after iteration 100; first {r[:12]}")
