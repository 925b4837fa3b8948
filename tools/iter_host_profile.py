"""Developer probe: host-side cost (cProfile) of one eager batched iteration of the config-3 loop."""
import sys, os, types, cProfile, pstats, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fluidnexus_amd import rasterizer
a = types.SimpleNamespace(no_graph=False, host_sync=False, scene="backdrop", stage="physical", no_distance=False, views="batched",
                          unfused_physics=False, image_loss="fused", emulate_world=0, shared_terms="per-view", physics_once=False,
                          torch_adam=False, graph_iters=5, sort="coherent")
dev = torch.device("cuda", 0)
rasterizer.set_blend_math("fast"); rasterizer.set_lean_geometry(True); rasterizer.set_coherent_sort(True); rasterizer.set_host_sync(False)
gm, cams, loop = bench.build_workload(3, 5, dev, 0, 1, a, False)
loop.make_targets()
for _ in range(5):
    loop.iteration()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(20):
    loop.iteration()
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host ms per eager iteration (enqueue only):", (t1 - t0) / 20 * 1e3)
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
