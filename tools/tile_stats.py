"""Developer tool: per-tile list statistics of one view of a benchmark scene (what the blend kernels have to walk)."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidnexus_amd import synthetic as S
from tests.hip_harness import HipRun, scene_kwargs

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
view = int(sys.argv[2]) if len(sys.argv) > 2 else 2
SIZE = 512
if cfg == 2:
    g, C, cams = S.plume_gaussians(100_000, seed=0, channels=1), 1, S.arc_cameras(5, SIZE, SIZE, device="cpu")
elif cfg == 3:
    g, C, cams = S.smoke_scene(200_000, 100_000, seed=0, channels=3), 3, S.arc_cameras(5, SIZE, SIZE, device="cpu")
else:
    g, C, cams = S.smoke_scene(350_000, 150_000, seed=0, channels=3, ring=True), 3, S.ring_cameras(8, SIZE, SIZE, device="cpu")
h = HipRun(bg=np.zeros(3, np.float32), colors_precomp=g["colors"], scales=g["scales"], rotations=g["rotations"], channels=C,
           **scene_kwargs(g, cams[view], SIZE, SIZE, 0.8))
it = h.intermediates()
gx = gy = SIZE // 16
T = gx * gy
lens = (it["ranges"][:, 1] - it["ranges"][:, 0]).astype(np.int64)
ncon = it["n_contrib"].astype(np.int64).reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(T, 256)
fT = it["final_T"].reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(T, 256)
mx, mean = ncon.max(1), ncon.mean(1)
print("R", h.R, "tiles with entries", int((lens > 0).sum()), "list length: mean %.0f max %d" % (lens.mean(), lens.max()))
print("consumed (max n_contrib per tile): sum %d  = %.1f%% of R; batches of 256: %d" % (mx.sum(), 100.0 * mx.sum() / h.R, np.ceil(mx / 256).sum()))
order = np.argsort(-mx)
print("deepest tiles: (tile, list length, max n_contrib, mean n_contrib, min final_T)")
for t in order[:12]:
    print("  ", t, lens[t], mx[t], round(mean[t], 1), float(fT[t].min()))
for thr in (256, 512, 1024, 2048, 4096):
    sel = mx > thr
    print("tiles with consumed > %d: %d, holding %.1f%% of the consumed entries" % (thr, sel.sum(), 100.0 * mx[sel].sum() / max(mx.sum(), 1)))
# pixels hit per entry inside the consumed prefix of the deepest tile (how sparse are the splats?)
t = order[0]
ids = it["point_list"][it["ranges"][t, 0]: it["ranges"][t, 0] + mx[t]]
r = it["radii"][ids]
print("deepest tile: radii of its consumed entries: mean %.1f max %d; fluid share %.2f" % (r.mean(), r.max(), (ids < (100_000 if cfg == 2 else 200_000 if cfg == 3 else 350_000)).mean()))
