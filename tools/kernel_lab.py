"""Developer tool: run the bench scene's view-batched rasteriser forward+backward K times (eager), so that
`rocprofv3 --kernel-trace --stats` gives clean per-kernel durations.  FNX_RASTER_LIB selects a library
variant built by tools/build_variant.py."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidnexus_amd import rasterizer
from fluidnexus_amd.harness import build_smoke_frame
from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
from fluidnexus_amd.renderer.pipes import render_dynamics_views

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--no-backward", action="store_true")
a = ap.parse_args()
gm, cams = build_smoke_frame(n_views=a.views, size=a.size)
gm.training_setup_current(__import__("types").SimpleNamespace(position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                                                              position_lr_delay_mult=0.01, position_lr_max_steps=30000))
_, GRsetting, GRzer = get_render_pipe("render_dynamics")
bg = torch.zeros(3, device="cuda")
rasterizer.set_host_sync(False)
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
for it in range(a.iters + 2):
    if it == 2:
        torch.cuda.synchronize()
        e0.record()
    pkg = render_dynamics_views(cams, gm, None, bg, GRsetting=GRsetting, GRzer=GRzer, pos_type="guess_visual_nn", scale=True)
    if not a.no_backward:
        pkg["render"].sum().backward()
        gm.optimizer.zero_grad()
e1.record()
torch.cuda.synchronize()
rasterizer.check_status()
print(f"{e0.elapsed_time(e1) / a.iters:.3f} ms per batched pass; num_rendered(last view) {rasterizer.last_num_rendered}")
