"""Developer tool: where does a frame of bench.py --frames go?  Times eager / replayed iterations after a frame boundary and
prints the coherent sort's fallback counters."""
import sys, time, types
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import bench
from fluidnexus_amd import rasterizer
from fluidnexus_amd.renderer import pipes

a = types.SimpleNamespace(no_graph=False, host_sync=False, scene="backdrop", stage="physical", no_distance=False, views="batched",
                          unfused_physics=False, image_loss="fused", emulate_world=0, shared_terms="per-view", physics_once=False,
                          torch_adam=False, frames=2, iters_per_frame=60, graph_iters=5, sort="radix" if "--radix" in sys.argv else "coherent")
dev = torch.device("cuda", 0)
rasterizer.set_blend_math("fast"); rasterizer.set_lean_geometry(True); rasterizer.set_coherent_sort("--radix" not in sys.argv)
rasterizer.set_host_sync(False)
orig = bench.Hn_HotLoop_iteration = None
from fluidnexus_amd import harness as Hn
it0 = Hn.HotLoop.iteration
times = []
def timed(self):
    torch.cuda.synchronize(); t = time.perf_counter(); it0(self); torch.cuda.synchronize()
    times.append(((time.perf_counter() - t) * 1e3 / self.iterations_per_call, self.iterations_per_call))
Hn.HotLoop.iteration = timed
out = bench.sequence_timing(a, dev, 3, 0, 1, False, 1.04)
print({k: v for k, v in out.items() if k != "note"})
print("per-iteration ms (last 40 calls):", [f"{t:.2f}x{k}" for t, k in times[-40:]])
for vb, _, _ in pipes._VIEW_BATCH_CACHE.values():
    for key in list(vb._sort_state):
        print("sort state", key, vb.sort_counters(key[0], key[1], why=True, sort_key=key[2]))
