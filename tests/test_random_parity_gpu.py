"""-m gpu: seeded random scenes through the single-view C entry points against the oracle: odd image sizes (partial
tiles), few and many splats, small and screen-filling footprints, low and high opacities (early saturation and
deep, non-saturating lists), both channel counts, SH colours.  Forward bit-exact (integer state, colour, depth,
final_T), backward within the fp32 summation tolerance."""
import os

import numpy as np
import pytest

from fluidnexus_amd import synthetic as S
from tests.hip_harness import HipRun, scene_kwargs

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.RandomState(1000 + seed)
    W, H = int(rng.randint(17, 200)), int(rng.randint(17, 200))
    P = int(rng.choice([1, 7, 300, 2500, 9000]))
    channels = int(rng.choice([1, 3]))
    lo = rng.uniform(-6.0, -3.5)
    g = S.random_gaussians(P, seed=seed, box=float(rng.uniform(0.15, 0.6)), log_scale=(lo, lo + rng.uniform(0.3, 2.5)),
                           channels=channels)
    mode = rng.choice(["thin", "mixed", "opaque"])
    if mode == "thin":      # semi-transparent: nothing saturates, every list is walked to its end
        g["opacities"] = rng.uniform(0.004, 0.05, size=g["opacities"].shape).astype(np.float32)
    elif mode == "opaque":  # early saturation
        g["opacities"] = rng.uniform(0.6, 1.0, size=g["opacities"].shape).astype(np.float32)
    sh = None
    if channels == 3 and rng.rand() < 0.3:
        sh = (int(rng.randint(0, 4)), rng.normal(scale=0.4, size=(P, 16, 3)).astype(np.float32))
    return W, H, P, channels, g, sh, mode


@pytest.mark.parametrize("seed", range(int(os.environ.get("FNX_RANDOM_CASES", "14"))))  # more cases for a one-off sweep
def test_random_scene_matches_oracle(oracle, seed):
    W, H, P, channels, g, sh, mode = _case(seed)
    cam = S.front_camera(W, H, device="cpu")
    bg = np.array([0.3, 0.1, 0.6], np.float32)
    kw = scene_kwargs(g, cam, W, H, 0.8)
    extra = dict(scales=g["scales"], rotations=g["rotations"])
    if sh is not None:
        extra.update(shs=sh[1], sh_degree=sh[0])
    else:
        extra.update(colors_precomp=g["colors"])
    f = oracle.forward(kw["means3D"], kw["opacities"], bg, kw["view"], kw["proj"], kw["campos"], W, H, kw["tanx"],
                       kw["tany"], channels=channels, **extra)
    h = HipRun(bg=bg, channels=channels, **kw, **extra)
    it = h.intermediates()
    assert h.R == f["num_rendered"], (seed, mode)
    for k in ("radii", "tiles_touched", "ranges", "point_list", "n_contrib"):
        assert (it[k].astype(np.int64) == f[k].astype(np.int64)).all(), (k, seed, W, H, P, channels, mode)
    for k in ("color", "depth", "final_T"):
        assert (it[k].view(np.uint32) == f[k].view(np.uint32)).all(), (k, seed, W, H, P, channels, mode)
    dL = np.random.RandomState(seed).normal(size=(channels, H, W)).astype(np.float32)
    go, gh = oracle.backward(f, dL), h.backward(dL)
    for k, ref in go.items():
        if ref.size == 0 or np.abs(ref).max() == 0:
            continue
        err = np.abs(gh[k].reshape(ref.shape).astype(np.float64) - ref.astype(np.float64)).max() / np.abs(ref).max()
        assert err < 2e-4, (k, err, seed, W, H, P, channels, mode)
