"""Developer tool: by how many depth ranks do the fluid splats move between two consecutive iterations of a LATER frame of
bench.py --frames (where the coherent sort's repair window is exceeded)?  Prints, per view, the displacement histogram."""
import sys, types
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import bench
from fluidnexus_amd import rasterizer, harness as Hn

a = types.SimpleNamespace(no_graph=True, host_sync=False, scene="backdrop", stage="physical", no_distance=False, views="batched",
                          unfused_physics=False, image_loss="fused", emulate_world=0, shared_terms="per-view", physics_once=False,
                          torch_adam=False, frames=2, iters_per_frame=int(sys.argv[1]) if len(sys.argv) > 1 else 10, graph_iters=5,
                          sort="radix")
dev = torch.device("cuda", 0)
rasterizer.set_blend_math("fast"); rasterizer.set_lean_geometry(True); rasterizer.set_coherent_sort(False)
rasterizer.set_host_sync(False)
it0 = Hn.HotLoop.iteration
state = {"prev": None, "calls": 0}
def ranks(self):
    gm = self.gm
    with torch.no_grad():
        x = (gm._estimate_xyz_nn.detach() * gm.scale_factor)
        from fluidnexus_amd import physics
        vis = physics.visual_from_hidden(gm._visual_xyz.detach(), x, gm._xyz, gm.H, gm._secs, gm.EPSILON) / gm.scale_factor
    out = []
    for c in self.cams:
        W = c.world_view_transform
        z = vis @ W[:3, 2] + W[3, 2]
        out.append(torch.argsort(torch.argsort(z)))
    return out, vis
def timed(self):
    it0(self)
    state["calls"] += 1
    r, vis = ranks(self)
    p = state["prev"]
    if p is not None and p[0][0].shape == r[0].shape:
        line = []
        for v in range(len(r)):
            d = (r[v] - p[0][v]).abs()
            line.append(f"v{v}: max {int(d.max())} >256:{int((d > 256).sum())} >768:{int((d > 768).sum())} >1024:{int((d > 1024).sum())} >4096:{int((d > 4096).sum())}")
        mv = (vis - p[1]).norm(dim=1)
        print(f"call {state['calls']} N {r[0].shape[0]} | " + " | ".join(line) + f" | moved max {float(mv.max()):.4f} >0.01: {int((mv > 0.01).sum())}")
    state["prev"] = (r, vis)
Hn.HotLoop.iteration = timed
out = bench.sequence_timing(a, dev, 3, 0, 1, False, 1.04)
