"""Developer tool: find which part of the iteration breaks hipGraph capture."""
import math, sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidnexus_amd import rasterizer
from fluidnexus_amd.harness import HotLoop, build_smoke_frame

which = sys.argv[1]
rasterizer.set_host_sync(False)
gm, cams = build_smoke_frame(P_fluid=20000, P_background=5000, hidden_dims=(8, 20, 8), n_views=2, size=128)
loop = HotLoop(gm, cams, fused_physics=True, defer_visual_backward=True, image_loss="fused", capturable=True)
loop.make_targets()
for _ in range(0 if os.environ.get("NOITER") == "1" else 3):
    loop.iteration()
rasterizer.check_status()
torch.cuda.synchronize()


def body():
    cam = cams[0]
    if which == "adam":
        gm._estimate_xyz_nn.grad = torch.ones_like(gm._estimate_xyz_nn)
        gm.optimizer.step()
        return
    if which == "physics":
        from fluidnexus_amd.physics import physical_stage_loss
        l = physical_stage_loss(gm, 0.1, 1.0, 0.1, None)
        l.backward()
        return
    if which == "density_fwd":
        from fluidnexus_amd import physics
        with torch.no_grad():
            x = gm._estimate_xyz_nn * gm.scale_factor
            physics.density_ratio(x, gm._imass, gm.H, gm.p0, gm._cached_grid("est", x))
        return
    if which == "density_fwd_nogrid":
        from fluidnexus_amd import physics
        with torch.no_grad():
            x = gm._estimate_xyz_nn * gm.scale_factor
            physics.density_ratio(x, gm._imass, gm.H, gm.p0, GRID[0])
        return
    if which == "density_bwd_direct":
        from fluidnexus_amd import physics, _physics_lib as PL
        x = (gm._estimate_xyz_nn * gm.scale_factor).detach().contiguous()
        up = torch.ones(x.shape[0], 1, device=x.device)
        dx = torch.empty_like(x)
        PL.check(PL.physics().fnx_density_backward(x.data_ptr(), x.shape[0], gm._imass.data_ptr(), gm.H, gm.p0,
                                                   GRID[0].blob.data_ptr(), up.data_ptr(), dx.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream))
        return
    if which == "density_pregrid":
        from fluidnexus_amd import physics
        x = gm._estimate_xyz_nn * gm.scale_factor
        physics.density_ratio(x, gm._imass, gm.H, gm.p0, GRID[0]).sum().backward()
        return
    if which == "density_ones":
        from fluidnexus_amd import physics
        x = gm._estimate_xyz_nn * gm.scale_factor
        w = torch.ones(x.shape[0], 1, device=x.device)
        (physics.density_ratio(x, gm._imass, gm.H, gm.p0, GRID[0]) * w).sum().backward()
        return
    if which == "puretorch":
        (gm._estimate_xyz_nn * 2.0).sum().backward()
        return
    if which == "fresh":
        (FRESH[0] * 2.0).sum().backward()
        return
    if which == "puretorch_pregrad":
        (gm._estimate_xyz_nn * 2.0).sum().backward()
        return
    if which == "grid":
        from fluidnexus_amd import physics
        x = (gm._estimate_xyz_nn * gm.scale_factor).detach()
        physics.HashGrid(x, gm.H)
        return
    if which == "density":
        from fluidnexus_amd import physics
        x = gm._estimate_xyz_nn * gm.scale_factor
        physics.density_ratio(x, gm._imass, gm.H, gm.p0, gm._cached_grid("est", x)).sum().backward()
        return
    if which == "visual":
        v = gm.get_visual_xyz_from_nn()
        v.sum().backward()
        gm.flush_deferred_gradients() if hasattr(gm, "_estimate_xyz_nn_grad") else None
        return
    with torch.set_grad_enabled(which != "fwd"):
        pkg = loop.render_func(cam, gm, None, loop.background, GRsetting=loop.GRsetting, GRzer=loop.GRzer,
                               pos_type="visual" if which in ("fwd", "fwdbwd_novis") else "guess_visual_nn", scale=True)
    if which == "fwd":
        return
    if which in ("fwdbwd", "fwdbwd_novis"):
        pkg["render"].sum().backward()
        return
    if which == "loss":
        loss, _, _ = loop._image_loss(pkg["render"], cam.original_image)
        loss.backward()


from fluidnexus_amd import physics as _ph
FRESH = [torch.nn.Parameter(torch.randn(1000, 3, device='cuda'))]
GRID = [_ph.HashGrid((gm._estimate_xyz_nn * gm.scale_factor).detach(), gm.H)]
gm.invalidate_caches()
gm.zero_gradient_cache_current()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
gm.invalidate_caches()
if which != "puretorch_pregrad":
    gm._estimate_xyz_nn.grad = None
FRESH[0].grad = None
if os.environ.get("NOITER") == "1":
    pass
g = torch.cuda.CUDAGraph()
mode = sys.argv[2] if len(sys.argv) > 2 else "global"
with torch.cuda.graph(g, capture_error_mode=mode):
    body()
g.replay()
torch.cuda.synchronize()
print("OK", which)
