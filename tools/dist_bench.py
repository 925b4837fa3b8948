"""Developer tool (round 5): the distance loss's kernels alone on config 3's rendered positions.
usage: rocprofv3 --kernel-trace --stats -- python tools/dist_bench.py lists|verlet|rebuild"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fluidnexus_amd import physics, synthetic as S
mode = sys.argv[1] if len(sys.argv) > 1 else "verlet"
x = torch.tensor(S.plume_gaussians(200_000, seed=0, channels=1)["means3D"], device="cuda")
xs = [x, (x + 0.0011).contiguous()]
thr = 0.002
physics.set_distance_verlet(mode != "lists")
for it in range(23):
    if it == 3:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    physics.distance_loss_value_and_grad(xs[it & 1] if mode == "rebuild" else x, thr)
torch.cuda.synchronize()
print(mode, "%.1f us per call (host clock, eager launches)" % ((time.perf_counter() - t0) / 20 * 1e6), physics.distance_verlet_counters())
