import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidnexus_amd import rasterizer
from fluidnexus_amd.harness import HotLoop, build_smoke_frame
pf, pb = int(sys.argv[1]), int(sys.argv[2])
size = int(sys.argv[3])
rasterizer.set_host_sync(False)
gm, cams = build_smoke_frame(P_fluid=pf, P_background=pb, hidden_dims=(20, 62, 20), n_views=5, size=size)
loop = HotLoop(gm, cams, fused_physics=True, defer_visual_backward=True, image_loss="fused", capturable=True)
loop.make_targets()
for _ in range(3):
    loop.iteration()
rasterizer.check_status()
print("eager ok", file=sys.stderr)
loop.capture(warmup=1)
torch.cuda.synchronize()
print("capture ok", file=sys.stderr)
for i in range(5):
    loop.iteration()
    torch.cuda.synchronize()
    print("replay", i, "ok", file=sys.stderr)
rasterizer.check_status()
print("status ok, R =", rasterizer.last_num_rendered, file=sys.stderr)
mode = sys.argv[4] if len(sys.argv) > 4 else "b2b"
import time
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(30):
        loop.iteration()
        if mode == "sync5" and i % 5 == 4:
            torch.cuda.synchronize()
        if mode == "sync1":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(mode, "30 replays ok, ms/iter", (time.perf_counter() - t0) / 30 * 1e3, file=sys.stderr)
    rasterizer.check_status()
