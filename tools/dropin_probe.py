"""Developer probe (round 5): the drop-in loop with the seam's automation on -- wall time per iteration, to be run under
rocprofv3 --kernel-trace --stats for the kernels' share.  usage: python tools/dropin_probe.py [0|1 auto] [steps]"""
import sys, os, json, types, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
a = types.SimpleNamespace(no_graph=False, host_sync=False, scene="backdrop", stage="physical", no_distance=False, views="batched",
                          unfused_physics=False, image_loss="fused", emulate_world=0, shared_terms="per-view", physics_once=False,
                          torch_adam=False, graph_iters=5, sort="coherent")
dev = torch.device("cuda", 0)
auto = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
print(json.dumps(bench.drop_in_timing(a, dev, 3, steps=steps, auto=auto))[:110])
