"""Developer tool: step through the forced-distributed (1 rank, RCCL) graph path of HotLoop with syncs and prints."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/fnx_rccl_debug_%h_%p.log")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "dist"
if "nodist" not in mode:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from fluidnexus_amd import rasterizer
from fluidnexus_amd.harness import HotLoop, build_smoke_frame
cfg = os.environ.get("DBG_CFG", "small")
if cfg == "small":
    gm, cams = build_smoke_frame(P_fluid=20000, P_background=5000, hidden_dims=(8, 20, 8), n_views=2, size=128)
elif cfg == "mid":
    gm, cams = build_smoke_frame(P_fluid=100000, P_background=50000, hidden_dims=(14, 40, 14), n_views=3, size=256)
else:
    gm, cams = build_smoke_frame()
loop = HotLoop(gm, cams, force_all_reduce=("nodist" not in mode), image_loss="fused", fused_physics=True, defer_visual_backward=True,
               capturable=True, batched_views=True, fused_step=os.environ.get("DBG_FUSED", "1") == "1")
if mode == "noop":
    dist.all_reduce = lambda *a, **k: None
def step(msg, fn):
    fn(); torch.cuda.synchronize(); print("ok:", msg, flush=True)
step("targets", loop.make_targets)
rasterizer.set_host_sync(False)
for i in range(3):
    step(f"eager {i}", loop.iteration)
step("check_status", rasterizer.check_status)
step("capture", lambda: loop.capture(warmup=1))
for i in range(4):
    step(f"replay {i}", loop.iteration)
if "plaind2h" in mode:
    step("plain d2h", lambda: print(torch.zeros(4, device="cuda").cpu().sum().item()))
elif "cloned2h" in mode:
    step("clone d2h", lambda: print(rasterizer._status_ring[0].clone().cpu().sum().item()))
elif "ringd2h" in mode:
    step("ring d2h", lambda: print(rasterizer._status_ring[0].cpu().sum().item()))
elif "nocheck" not in mode:
    step("check_status", rasterizer.check_status)
for i in range(4):
    step(f"replay b{i}", loop.iteration)
print("done", float(gm._estimate_xyz_nn.abs().sum()))
