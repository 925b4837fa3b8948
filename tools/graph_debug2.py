import sys, os, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidnexus_amd import rasterizer
from fluidnexus_amd.harness import HotLoop, build_smoke_frame
mode = sys.argv[1]
rasterizer.set_host_sync(False)
gm, cams = build_smoke_frame(P_fluid=200000, P_background=100000, hidden_dims=(20, 62, 20), n_views=5, size=512)
loop = HotLoop(gm, cams, fused_physics=True, defer_visual_backward=True, image_loss="fused", capturable=True,
               parallel_views=(mode == "parallel"))
loop.make_targets()
for _ in range(3):
    loop.iteration()
rasterizer.check_status()
loop.capture(warmup=1)
torch.cuda.synchronize()
print("capture ok", file=sys.stderr)
if mode == "status_first":
    rasterizer.check_status()
    print("status done", file=sys.stderr)
if mode == "clear_only":
    rasterizer._pending_status.clear()
    print("cleared pending (tensors freed)", file=sys.stderr)
import time
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(30):
        loop.iteration()
    torch.cuda.synchronize()
    print(mode, "ms/iter", (time.perf_counter() - t0) / 30 * 1e3, file=sys.stderr)
rasterizer.check_status()
