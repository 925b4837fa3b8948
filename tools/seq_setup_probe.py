"""Developer tool: what the per-frame set-up of bench.py --frames is made of (loop object, eager iterations, capture)."""
import sys, time, types, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fluidnexus_amd import rasterizer, harness as Hn

a = types.SimpleNamespace(no_graph=False, host_sync=False, scene="backdrop", stage="physical", no_distance=False, views="batched",
                          unfused_physics=False, image_loss="fused", emulate_world=0, shared_terms="per-view", physics_once=False,
                          torch_adam=False, graph_iters=5, sort="coherent")
dev = torch.device("cuda", 0)
rasterizer.set_blend_math("fast"); rasterizer.set_lean_geometry(True); rasterizer.set_coherent_sort(True); rasterizer.set_host_sync(False)
gm, cams, loop = bench.build_workload(3, 5, dev, 0, 1, a, False)
loop.make_targets()
def tick():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = tick()
    gm.prepare_visual_particles_for_rendering()
    rasterizer.release_captured_status()
    loop = Hn.HotLoop(gm, cams, image_loss="fused", fused_physics=True, defer_visual_backward=True, capturable=True, cfg=loop.cfg,
                      batched_views=True, fused_step=True)
    t1 = tick()
    loop.iteration(); t2 = tick()
    loop.iteration(); t3 = tick()
    rasterizer.check_status(); t4 = tick()
    for gi in (5,):
        loop.capture(warmup=1, iterations=gi)
    t5 = tick()
    loop.iteration(); t6 = tick()
    print(f"loop object {1e3*(t1-t0):.1f} ms, eager it 1 {1e3*(t2-t1):.1f}, eager it 2 {1e3*(t3-t2):.1f}, check_status {1e3*(t4-t3):.1f}, "
          f"capture(warmup 1, 5 iterations) {1e3*(t5-t4):.1f}, first replay {1e3*(t6-t5):.1f}")
