"""Experiment: latency of the blend forward kernel on single tiles (FNX_ONLY_TILE build) vs the full grid."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fluidnexus_amd import synthetic as S
from tests.hip_harness import HipRun, scene_kwargs
from fluidnexus_amd import _lib
g = S.smoke_scene(200000, 100000)
cam = S.arc_cameras(5, 512, 512, device="cpu")[0]
kw = scene_kwargs(g, cam, 512, 512)
bg = np.zeros(3, np.float32)
h = HipRun(bg=bg, colors_precomp=g["colors"], scales=g["scales"], rotations=g["rotations"], **kw)
it = h.intermediates()
nc = it["n_contrib"].reshape(32, 16, 32, 16).transpose(0, 2, 1, 3).reshape(1024, 256)
E = nc.max(1)
order = np.argsort(-E)
print("heaviest tiles", [(int(t), int(E[t])) for t in order[:3]], "median tile", int(order[512]), int(E[order[512]]))
lib = _lib.raster()
s = torch.cuda.current_stream().cuda_stream
def time_stage2(n=20):
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    _lib.profile_enable(True)
    for _ in range(n):
        _lib.check(lib.fnx_forward_stage2(3, h.geom.data_ptr(), h.binning.data_ptr(), h.cap, h.img.data_ptr(), h.P, 512, 512,
                                          h.bg.data_ptr(), h.colors.data_ptr(), h.radii.data_ptr(), h.color.data_ptr(), h.depth.data_ptr(), s))
    torch.cuda.synchronize()
    ms, k = _lib.profile_read(0)
    _lib.profile_enable(False)
    return ms / k * 1e3
for t in [None, int(order[0]), int(order[1]), int(order[512]), int(order[900])]:
    if t is None:
        os.environ.pop("FNX_ONLY_TILE", None)
    else:
        os.environ["FNX_ONLY_TILE"] = str(t)
    print("tile", t, "E", None if t is None else int(E[t]), "blend forward us", round(time_stage2(), 1))
