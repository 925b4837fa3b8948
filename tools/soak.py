"""Developer tool: soak the captured physical-stage loop of config 3 (thousands of graph replays): parameters stay
finite, the image error goes down, the status ring keeps reporting, device memory does not grow."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidnexus_amd import harness as Hn, rasterizer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
gm, cams = Hn.build_smoke_frame()
loop = Hn.HotLoop(gm, cams, image_loss="fused", fused_physics=True, defer_visual_backward=True, capturable=True,
                  batched_views=True, fused_step=True, cfg=dict(Hn.SMOKE))
loop.make_targets()
rasterizer.set_host_sync(False)
rasterizer.set_blend_math("exact" if "exact" in sys.argv[2:] else "fast")  # the bench's settings
rasterizer.set_lean_geometry(True)
rasterizer.set_coherent_sort("radix" not in sys.argv[2:])  # the bench's depth sort
for _ in range(3):
    loop.iteration()
rasterizer.check_status()
if 0 < rasterizer.max_sort_span_bits <= 25:  # the bench's margin: two bits below what three passes order
    rasterizer.set_sort_narrow(True)
loop.log_scalars = True
loop.iteration()
first = dict(loop.last)
loop.log_scalars = False
loop.capture(warmup=1, iterations=5)
torch.cuda.synchronize()
mem0 = torch.cuda.memory_allocated()
t0 = time.time()
done = 0
while done < steps:
    loop.iteration()
    done += loop.iterations_per_call
    if done % 5000 == 0:
        rasterizer.check_status()
torch.cuda.synchronize()
dt = time.time() - t0
rasterizer.check_status()
loop.use_graph(False)
loop.log_scalars = True
loop.iteration()
x = gm._estimate_xyz_nn.detach()
print(f"{done} iterations in {dt:.1f} s = {done / dt:.0f} it/s; finite {bool(torch.isfinite(x).all())}; "
      f"loss {first['total']:.6f} -> {loop.last['total']:.6f}; l1 {first['l1']:.6f} -> {loop.last['l1']:.6f}; "
      f"memory {mem0 / 2**20:.0f} -> {torch.cuda.memory_allocated() / 2**20:.0f} MiB; step {float(gm.optimizer.state[gm._estimate_xyz_nn]['step']):.0f}")
for vb in list(rasterizer._VIEW_BATCHES or ()):
    for key in list(vb._sort_state):
        print("coherent sort, per view (calls, in-launch full sorts, why, outliers taken):", vb.sort_counters(key[0], key[1], why=True, outliers=True, sort_key=key[2]))
