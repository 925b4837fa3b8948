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
buf = (C.c_ulonglong * 32)()
for it in range(4):
    if it == 3:
        torch.cuda.synchronize()
        lib.fnx_debug_bwd_clock(buf, 1)
    pkg = render_dynamics_views(cams, gm, None, bg, GRsetting=S_, GRzer=Z_, pos_type="guess_visual_nn", scale=True, screen_grad=False)
    pkg["render"].sum().backward()
    gm.optimizer.zero_grad()
torch.cuda.synchronize()
lib.fnx_debug_bwd_clock(buf, 0)
if "--lanes" in sys.argv:  # the entries-as-lanes form (FNX_BWD_FORM=1): lane 0 of every wave
    names = ["flush: global atomics (to loop top)", "staging", "wait barrier B", "walk (list builds + chunks)", "prefetch requests",
             "wait barrier C", "flush: sums -> gradients"]
    tot = sum(int(buf[i]) for i in range(7))
    for i, n in enumerate(names):
        print(f"{n:36s} {int(buf[i]) / 4096 / 100:8.1f} us per wave  {100 * int(buf[i]) / tot:5.1f} %")
    for i, n in ((16, "  ticket wait (wave 0)"), (17, "  wait for the next item's ids"), (18, "  record requests"), (19, "  pixel requests"),
                 (20, "  mean / covariance gather requests"), (21, "  next-next item descriptor"), (22, "  sums -> gradients (+ gather wait)"),
                 (23, "  wait for the next item's records / pixels")):
        print(f"{n:46s} {100 * int(buf[i]) / tot:5.1f} %")
    items, ents, ln, ch, nb, em = (int(buf[i]) for i in range(8, 14))
    print(f"total {tot / 4096 / 100:.1f} us per wave if the counter runs at 100 MHz (4096 waves)")
    print(f"items {items}, staged entries {ents} ({ents / max(items, 1):.1f} per item); walked blocks {nb} ({nb / max(items, 1):.2f} per item), "
          f"list entries {ln} ({ln / max(nb, 1):.1f} per walked block, {ln / max(ents, 1):.2f} blocks per staged entry), chunks {ch} "
          f"(lane fill {ln / max(16 * ch, 1):.3f}), chunks with a gradient {em} ({em / max(ch, 1):.3f}); flushed entries {int(buf[14])} ({int(buf[14]) / max(ents, 1):.3f} of the staged)")
    sys.exit(0)
names = ["flush (to loop top)", "item head (pixel inputs, maxima)", "wait barrier A", "staging writes", "wait barrier B", "list build",
         "walk", "prefetch + wait barrier C"]
tot = sum(int(buf[i]) for i in range(8))
for i, n in enumerate(names):
    print(f"{n:36s} {int(buf[i]) / 1024 / 2100:8.1f} us per workgroup  {100 * int(buf[i]) / tot:5.1f} %")
print(f"total {tot / 1024 / 2100:.1f} us per workgroup (1024 workgroups, 2.1 GHz assumed)")
