"""Developer tool: time the view-batched forward of config 3 with per-kernel breakdown (eager, events around the call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidnexus_amd import rasterizer
from fluidnexus_amd.harness import build_smoke_frame, build_scalar_real_frame
from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
from fluidnexus_amd.renderer.pipes import render_dynamics_views, render_fluid_views
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if cfg == 3:
    gm, cams = build_smoke_frame()
    gm.training_setup_current(__import__("types").SimpleNamespace(position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01, position_lr_max_steps=30000))
    _, S_, Z_ = get_render_pipe("render_dynamics")
    f = lambda: render_dynamics_views(cams, gm, None, bg, GRsetting=S_, GRzer=Z_, pos_type="guess_visual_nn", scale=True)
else:
    gm, cams = build_scalar_real_frame()
    _, S_, Z_ = get_render_pipe("render_fluid")
    f = lambda: render_fluid_views(cams, gm, None, bg, GRsetting=S_, GRzer=Z_, pos_type="visual")
bg = torch.zeros(3, device="cuda")
rasterizer.set_host_sync(False)
rasterizer.set_blend_math(os.environ.get("FNX_MATH", "exact"))
if os.environ.get("FNX_DEEP_KERNEL"):
    rasterizer.set_deep_kernel(int(os.environ["FNX_DEEP_KERNEL"]))
with torch.no_grad():
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
print("forward ms", e0.elapsed_time(e1) / 10)
import ctypes as C
from fluidnexus_amd import _lib
lib = _lib.raster()
if hasattr(lib, "fnx_debug_fwd_clock"):
    buf = (C.c_ulonglong * 64)()
    lib.fnx_debug_fwd_clock(buf)
    for w in range(4):
        v = [int(x) for x in buf[16 * w:16 * w + 16]]
        print("wave", w, dict(loop_top=v[0], all_done_barrier=v[1], masks_and_stores=v[4], barrier_b=v[5], list_build=v[6], merge_search=v[7], barrier_c=v[10], fetch_next=v[2], blend_loop=v[3], sum_n_w=v[8], batches=v[9], list_length=v[15]))
if hasattr(lib, "fnx_debug_ws_clock") and os.environ.get("FNX_DEEP_KERNEL", "0") in ("3", "4"):
    buf = (C.c_ulonglong * 128)()
    lib.fnx_debug_ws_clock(buf)
    for w in range(8):
        v = [int(x) for x in buf[16 * w:16 * w + 16]]
        if w < 4:
            print("walker", w, dict(loop_top=v[0], wait_for_batch=v[1], walk=v[2], batches=v[9], list_length=v[15]))
        else:
            print("stager", w - 4, dict(loop_top=v[0], wait_for_buffer=v[1], records_masks_stores=v[2], sbar1=v[3], lists=v[4],
                                        merge=v[5], sbar2=v[6], advance_requests=v[7]))
if hasattr(lib, "fnx_debug_fwd_stats"):
    buf = (C.c_ulonglong * 8)()
    lib.fnx_debug_fwd_stats(buf, 1)
    with torch.no_grad():
        f()
    torch.cuda.synchronize()
    lib.fnx_debug_fwd_stats(buf, 0)
    e, hits, any_hit, blocks, live, halves, rows_hit = [int(x) for x in buf[:7]]
    print(f"wave-entries {e}  lanes hit/entry {hits / e:.1f} of {live / e:.1f} alive  entries with a hit {any_hit / e:.3f}  "
          f"rows with a list entry per step {blocks / e:.2f} of 4; rows hit {rows_hit / e:.2f}; 4x2 half-rows hit {halves / e:.2f} of 8")
if hasattr(lib, "fnx_debug_fwd_wg"):
    import numpy as np
    n = 5 * 1024
    buf = (C.c_ulonglong * (4 * n))()
    lib.fnx_debug_fwd_wg(buf, 4 * n)
    a = np.array(buf[:], dtype=np.uint64).reshape(n, 4)
    t0, t1 = a[:, 0].astype(np.float64), a[:, 1].astype(np.float64)
    depth, length = (a[:, 2] >> np.uint64(32)).astype(np.int64), (a[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    base = t0.min()
    tick = 0.01  # us per tick of the 100 MHz wall clock
    s, e = (t0 - base) * tick, (t1 - base) * tick
    print(f"launch span {e.max():.1f} us; workgroups {n}; last start {s.max():.1f} us")
    order = np.argsort(-e)[:8]
    for i in order:
        print(f"  wg {i % 1024:4d} view {i // 1024}: start {s[i]:7.1f} end {e[i]:7.1f} dur {e[i] - s[i]:7.1f} us  consumed {depth[i]:5d} of {length[i]:5d}")
    for thr in (256, 1024, 2048, 4096):
        m = depth < thr
        print(f"  tiles consuming < {thr}: {m.sum():5d}, all ended by {e[m].max():7.1f} us")
    dur = e - s
    print(f"  sum of durations {dur.sum() / 1e3:.1f} ms = {dur.sum() / e.max():.0f} workgroups busy on average")
