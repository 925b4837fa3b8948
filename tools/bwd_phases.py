"""Developer tool: per-phase cycle totals of the blend backward (library built with -DFNX_EXP_BCLK, wave 0 of every
workgroup), config 3 scene, positions-only mode."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidnexus_amd import _lib, rasterizer
from fluidnexus_amd.harness import build_smoke_frame
from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
from fluidnexus_amd.renderer.pipes import render_dynamics_views
gm, cams = build_smoke_frame()
gm.training_setup_current(__import__("types").SimpleNamespace(position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01, position_lr_max_steps=30000))
_, S_, Z_ = get_render_pipe("render_dynamics")
bg = torch.zeros(3, device="cuda")
rasterizer.set_host_sync(False)
rasterizer.set_blend_math("fast")
lib = _lib.raster()
buf = (C.c_ulonglong * 16)()
for it in range(4):
    if it == 3:
        torch.cuda.synchronize()
        lib.fnx_debug_bwd_clock(buf, 1)
    pkg = render_dynamics_views(cams, gm, None, bg, GRsetting=S_, GRzer=Z_, pos_type="guess_visual_nn", scale=True, screen_grad=False)
    pkg["render"].sum().backward()
    gm.optimizer.zero_grad()
torch.cuda.synchronize()
lib.fnx_debug_bwd_clock(buf, 0)
names = ["flush (to loop top)", "item head (pixel inputs, maxima)", "wait barrier A", "staging writes", "wait barrier B", "list build",
         "walk", "prefetch + wait barrier C"]
tot = sum(int(buf[i]) for i in range(8))
for i, n in enumerate(names):
    print(f"{n:36s} {int(buf[i]) / 1024 / 2100:8.1f} us per workgroup  {100 * int(buf[i]) / tot:5.1f} %")
print(f"total {tot / 1024 / 2100:.1f} us per workgroup (1024 workgroups, 2.1 GHz assumed)")
