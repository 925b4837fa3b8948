"""Quick per-view timing of the rasteriser (forward / backward) on a BASELINE-shaped scene."""
import argparse
import math
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fluidnexus_amd import synthetic as S
from fluidnexus_amd import rasterizer as R

ap = argparse.ArgumentParser()
ap.add_argument("--fluid", type=int, default=200000)
ap.add_argument("--bgd", type=int, default=100000)
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--channels", type=int, default=3)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--nosync", action="store_true")
a = ap.parse_args()

g = S.to_torch(S.smoke_scene(a.fluid, a.bgd, channels=a.channels))
cams = S.arc_cameras(a.views, a.size, a.size)
bg = torch.zeros(3, device="cuda")
tan = math.tan(0.4)
if a.nosync:
    R.set_host_sync(False, 8_000_000)
leaves = {k: v.clone().requires_grad_(True) for k, v in g.items()}


def view_pass(cam, backward=True):
    rs = R.GaussianRasterizationSettings(a.size, a.size, tan, tan, bg, 1.0, cam.world_view_transform,
                                         cam.full_proj_transform, 0, cam.camera_center, False)
    rast = R.GaussianRasterizer(rs, a.channels)
    m2d = torch.zeros_like(leaves["means3D"], requires_grad=True)
    color, radii, depth = rast(leaves["means3D"], m2d, leaves["opacities"], colors_precomp=leaves["colors"],
                               scales=leaves["scales"], rotations=leaves["rotations"])
    if backward:
        color.sum().backward()
    return color


for cam in cams:
    view_pass(cam)
torch.cuda.synchronize()
for mode in ("fwd", "fwd+bwd"):
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(a.iters):
        for cam in cams:
            view_pass(cam, mode != "fwd")
    e1.record()
    torch.cuda.synchronize()
    print(f"{mode}: {e0.elapsed_time(e1) / a.iters / len(cams):.3f} ms per view")
from fluidnexus_amd import _lib
import ctypes as C
