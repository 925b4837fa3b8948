"""Developer probe: host-side profile (cProfile) of the automated drop-in loop."""
import sys, os, types, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, fluidnexus_amd
from fluidnexus_amd import rasterizer
from fluidnexus_amd.renderer import pipes
a = types.SimpleNamespace(no_graph=True, host_sync=True, scene="backdrop", stage="physical", no_distance=False, views="serial",
                          unfused_physics=True, image_loss="torch", emulate_world=0, shared_terms="per-view", physics_once=False,
                          torch_adam=True, graph_iters=5, sort="coherent")
dev = torch.device("cuda", 0)
rasterizer.set_blend_math("exact"); pipes.set_static_split(True); fluidnexus_amd.set_auto(True)
torch.backends.cudnn.enabled = False
gm, cams, loop = bench.build_workload(3, 5, dev, 0, 1, a, False)
loop.make_targets()
for _ in range(3):
    loop.iteration()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    loop.iteration()
pr.disable()
torch.cuda.synchronize()
print("ms per iteration (profiled):", (time.perf_counter() - t0) / 10 * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
