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
ap.add_argument("--math", default="exact", choices=["exact", "fast"])
ap.add_argument("--no-deep", action="store_true")
ap.add_argument("--deep", type=int, default=None)
ap.add_argument("--deep-min", type=int, default=None)
ap.add_argument("--no-split", action="store_true")
ap.add_argument("--mode3", action="store_true", help="positions-only backward (screen_grad=False), as the bench runs it")
a = ap.parse_args()
gm, cams = build_smoke_frame(n_views=a.views, size=a.size)
gm.training_setup_current(__import__("types").SimpleNamespace(position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                                                              position_lr_delay_mult=0.01, position_lr_max_steps=30000))
_, GRsetting, GRzer = get_render_pipe("render_dynamics")
bg = torch.zeros(3, device="cuda")
rasterizer.set_host_sync(False)
rasterizer.set_blend_math(a.math)
if a.no_split:
    from fluidnexus_amd.renderer import pipes as _pp
    _pp.set_static_split(False)
if a.deep is not None:
    from fluidnexus_amd import _lib as _l2
    rasterizer.set_deep_kernel(a.deep)
if a.deep_min is not None:
    rasterizer.set_deep_variant(True, a.deep_min)
if a.no_deep:
    from fluidnexus_amd import _lib as _l
    rasterizer.set_deep_kernel(0)
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
for it in range(a.iters + 2):
    if it == 2:
        torch.cuda.synchronize()
        e0.record()
    pkg = render_dynamics_views(cams, gm, None, bg, GRsetting=GRsetting, GRzer=GRzer, pos_type="guess_visual_nn", scale=True,
                                **({"screen_grad": False} if a.mode3 else {}))
    if not a.no_backward:
        pkg["render"].sum().backward()
        gm.optimizer.zero_grad()
e1.record()
torch.cuda.synchronize()
rasterizer.check_status()
print(f"{e0.elapsed_time(e1) / a.iters:.3f} ms per batched pass; num_rendered(last view) {rasterizer.last_num_rendered}")

import ctypes as C, numpy as np
from fluidnexus_amd import _lib
lib = _lib.raster()
if hasattr(lib, "fnx_debug_emit_clock"):
    nb = (gm._visual_xyz.shape[0] + gm._gs_xyz.shape[0] + 1023) // 1024
    n = nb * a.views
    buf = (C.c_ulonglong * (4 * n))()
    lib.fnx_debug_emit_clock(buf, 4 * n)
    arr = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
    t0 = arr[:, 0].min()
    st, en = (arr[:, 0] - t0) / 100.0, (arr[:, 1] - t0) / 100.0  # wall_clock64: 100 MHz -> microseconds
    dur = en - st
    print("emit WGs", n, "kernel span us", en.max(), "dur mean", dur.mean(), "median", np.median(dur), "max", dur.max(),
          "start max", st.max())
    order = np.argsort(-dur)[:8]
    for i in order:
        print("  wg", i, "view", i // nb, "block", i % nb, "start", st[i], "dur", dur[i], "sub-batches", arr[i, 2], "instances", arr[i, 3])
    print("  sub-batches total", arr[:, 2].sum(), "instances", arr[:, 3].sum())
    # time per sub-batch / per instance regression
    A = np.stack([np.ones(n), arr[:, 2], arr[:, 3]], 1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
    print("  fit dur = %.2f + %.2f * sub_batches + %.5f * instances (us)" % tuple(coef))
    big = (C.c_ulonglong * (4 * 16384))()
    lib.fnx_debug_emit_clock(big, 4 * 16384)
    names = ["setup loads", "prefix", "search", "walk", "atomicOr+barrier", "position+store+barrier", "cursor update", "-"]
    for off, label in ((0, "wg 0 (heavy)"), (8, "wg 200")):
        ph = [big[4 * 16000 + off + i] for i in range(8)]
        print(" ", label, "phase cycles:", {n: int(v) for n, v in zip(names, ph)}, "total", sum(ph))
if hasattr(lib, "fnx_debug_fwd_wg"):  # -DFNX_EXP_CLOCK builds: per-workgroup timeline of the LAST blend forward
    n = a.views * 1024
    buf = (C.c_ulonglong * (4 * n))()
    lib.fnx_debug_fwd_wg(buf, 4 * n)
    arr = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4)
    t0, t1 = arr[:, 0].astype(np.float64), arr[:, 1].astype(np.float64)
    depth, length = (arr[:, 2] >> np.uint64(32)).astype(np.int64), (arr[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    nbwd, staged = (arr[:, 3] >> np.uint64(32)).astype(np.int64), (arr[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    ok = t1 > 0
    s, e = (t0 - t0[ok].min()) * 0.01, (t1 - t0[ok].min()) * 0.01
    dur = e - s
    print(f"fwd: launch span {e[ok].max():.1f} us, last start {s[ok].max():.1f} us, sum of durations {dur[ok].sum() / 1e3:.2f} ms "
          f"= {dur[ok].sum() / e[ok].max():.0f} workgroups busy on average")
    print(f"fwd: list entries {length.sum()}, staged {staged.sum()}, deepest contributor sum {depth.sum()}, "
          f"batches staged {((staged + 255) // 256).sum()}, batches holding a contributor {((depth + 255) // 256).sum()}, "
          f"backward items {nbwd.sum()}")
    for q in (0.5, 0.8, 0.9, 0.95, 0.99, 1.0):
        print(f"   {q:4.2f} of the workgroups ended by {np.quantile(e[ok], q):7.1f} us")
    for lo, hi in ((0, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 4096), (4096, 1 << 30)):
        m = ok & (staged >= lo) & (staged < hi)
        if m.sum():
            print(f"   staged in [{lo}, {hi}): {m.sum():5d} tiles, mean duration {dur[m].mean():7.1f} us, per staged batch "
                  f"{dur[m].sum() / np.maximum(1, ((staged[m] + 255) // 256).sum()):6.2f} us, total {dur[m].sum() / 1e3:6.2f} ms")
    for i in np.argsort(-e)[:6]:
        print(f"   wg {i % 1024:4d} view {i // 1024}: start {s[i]:7.1f} end {e[i]:7.1f} dur {dur[i]:7.1f} us, staged {staged[i]:5d} "
              f"of {length[i]:5d}, deepest contributor {depth[i]:5d}")
