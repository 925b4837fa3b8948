"""Developer tool: time one PBF frame step (guess -> 10 x project_gas_constraints -> confirm -> update_visual_particles)
on the benchmark frame (24.8 k hidden, 200 k visual particles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidnexus_amd.harness import build_smoke_frame
gm, _ = build_smoke_frame(n_views=1, size=64)
gm.setup_solver_constants(alpha=0.0, buoyancy_decay_rate=0.0, min_neighbors=1)
gm._counts = torch.zeros(gm._xyz.shape[0], 1, device="cuda")
def frame():
    gm.guess_hidden_particles()
    for _ in range(10):
        gm.update_solver_counts()
    for _ in range(10):
        gm.project_gas_constraints()
    gm.confirm_guess_hidden_particles()
    gm.update_visual_particles()
for _ in range(3):
    frame()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
K = 20
for _ in range(K):
    frame()
e1.record()
torch.cuda.synchronize()
print(f"PBF frame step (N = {gm._xyz.shape[0]}, V = {gm._visual_xyz.shape[0]}, 10 solver iterations): "
      f"{e0.elapsed_time(e1) / K:.3f} ms (eager launches)")
