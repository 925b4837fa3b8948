"""Developer tool: the segmented blend forward (rasterizer.set_segmented_forward) against the one-list walk on the bench
scene -- images, depth, the positions' gradient -- and the two forwards' times (eager, events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidnexus_amd import rasterizer
from fluidnexus_amd.harness import build_smoke_frame
from fluidnexus_amd.helpers.helper_pipe import get_render_pipe
from fluidnexus_amd.renderer.pipes import render_dynamics_views

views = int(sys.argv[1]) if len(sys.argv) > 1 else 5
gm, cams = build_smoke_frame(n_views=views, size=512)
gm.training_setup_current(__import__("types").SimpleNamespace(position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                                                              position_lr_delay_mult=0.01, position_lr_max_steps=30000))
_, GRsetting, GRzer = get_render_pipe("render_dynamics")
bg = torch.zeros(3, device="cuda")
rasterizer.set_host_sync(False)
rasterizer.set_blend_math("fast")
torch.manual_seed(0)
dimg = None


def run(seg, backward=True):
    rasterizer.set_segmented_forward(seg)
    gm.invalidate_caches()
    pkg = render_dynamics_views(cams, gm, None, bg, GRsetting=GRsetting, GRzer=GRzer, pos_type="guess_visual_nn", scale=True,
                                screen_grad=False)
    g = None
    if backward:
        global dimg
        if dimg is None:
            dimg = torch.randn_like(pkg["render"]) * 1e-3
        gm.optimizer.zero_grad()
        leaf = gm._estimate_xyz_nn
        if leaf.grad is not None:
            leaf.grad = None
        pkg["render"].backward(dimg)
        g = leaf.grad.detach().clone()
    return pkg["render"].detach().clone(), pkg["depth"].detach().clone(), g


for it in range(8):
    a = run(False)
    b = run(True)
    gm.optimizer.step()  # the particles move between the calls, as in the loop
    torch.cuda.synchronize()
    rasterizer.check_status()
    dc, dd = (a[0] - b[0]).abs(), (a[1] - b[1]).abs()
    gg = (a[2] - b[2]).abs()
    print(f"call {it}: colour max diff {dc.max().item():.3e} (> 2e-5: {(dc > 2e-5).sum().item()} of {dc.numel()}), "
          f"depth max diff {dd.max().item():.3e} (differing: {(dd > 0).sum().item()}), gradient max diff {gg.max().item():.3e} "
          f"of max {a[2].abs().max().item():.3e}")
    print("   counters", rasterizer.segmented_forward_counters())
for seg in (False, True, False, True):
    with torch.no_grad():
        for _ in range(3):
            run(seg, backward=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10):
            run(seg, backward=False)
        e1.record()
        torch.cuda.synchronize()
    print(f"segments {seg}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per forward pass (all kernels of the view batch)")
print("   counters", rasterizer.segmented_forward_counters())
import ctypes as C, numpy as np
from fluidnexus_amd import _lib
lib = _lib.raster()
if hasattr(lib, "fnx_debug_fwd_wg"):  # -DFNX_EXP_CLOCK builds: per-workgroup timeline of the LAST blend forward (segments on)
    n = 16384
    torch.cuda.synchronize()
    lib.fnx_debug_fwd_wg_reset()
    with torch.no_grad():
        run(True, backward=False)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (4 * n))()
    lib.fnx_debug_fwd_wg(buf, 4 * n)
    b2 = (C.c_ulonglong * n)()
    lib.fnx_debug_fwd_wg2(b2, n)
    arr = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4)
    comb = np.frombuffer(b2, dtype=np.uint64).astype(np.float64)
    t0, t1 = arr[:, 0].astype(np.float64), arr[:, 1].astype(np.float64)
    ok = t1 > 0
    base = t0[ok].min()
    s, e, cs = (t0 - base) * 0.01, (t1 - base) * 0.01, (comb - base) * 0.01
    role = (arr[:, 3] >> np.uint64(48)).astype(np.int64)
    seg, nseg = role & 255, role >> 8
    staged = (arr[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    dur = e - s
    print(f"fwd: span {e[ok].max():.1f} us, workgroups {ok.sum()}, sum of durations {dur[ok].sum() / 1e3:.2f} ms")
    m = ok & (nseg == 0)
    print(f"   whole tiles: {m.sum()}, mean {dur[m].mean():.1f} us, total {dur[m].sum() / 1e3:.2f} ms, last end {e[m].max():.1f}")
    m = ok & (nseg > 0) & (comb == 0)
    print(f"   segments that only arrived: {m.sum()}, mean {dur[m].mean():.1f} us, max {dur[m].max():.1f}, total {dur[m].sum() / 1e3:.2f} ms, last end {e[m].max():.1f}")
    m = ok & (nseg > 0) & (comb > 0)
    print(f"   segments that went on: {m.sum()}, first round mean {(cs[m] - s[m]).mean():.1f} us, second round mean {(e[m] - cs[m]).mean():.1f} us "
          f"max {(e[m] - cs[m]).max():.1f}, total {dur[m].sum() / 1e3:.2f} ms, putting together starts at mean {cs[m].mean():.1f} max {cs[m].max():.1f}")
    for k in range(0, 9):
        mk = ok & (nseg > 0) & (seg == k)
        if mk.sum():
            print(f"      segment {k}: {mk.sum()} workgroups, start mean {s[mk].mean():.1f}, duration mean {dur[mk & (comb == 0)].mean() if (mk & (comb == 0)).sum() else 0:.1f} us, staged mean {staged[mk].mean():.0f}")
