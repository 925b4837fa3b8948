"""CPU: what can be pinned of the rasteriser oracle (oracle/raster_oracle.c, "parity unpinned" for the CUDA kernels)
with data produced by the reference itself:

  * its SH -> RGB stage and the SH / view-direction gradients of its backward against the reference's own
    utils/sh_utils.eval_sh evaluated as renderer/pipe.py:74-82 does (tests/golden/sh_positions.npz: colours, d colour /
    d SH and d colour / d position by autograd of the reference function) -- pins the SH half of forward.cu:20-67 and
    backward.cu:20-132 (+ the direction normalisation, auxiliary.h:95-118);
  * the public C headers compile as C (the drop-in boundary is a C ABI).
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "sh_positions.npz"))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_oracle_sh_forward_and_backward_match_reference_eval_sh(oracle, deg):
    O = oracle
    pos, campos, sh = G["pos"], G["campos"], G["sh"]
    P = pos.shape[0]
    M = (deg + 1) ** 2
    # a camera that sees every point (they lie within 1.0 of campos): looking down +z from 4 units behind them; the
    # SH direction uses `campos`, which the oracle takes as an argument of its own, like the reference kernels
    view = np.eye(4, dtype=np.float32)
    view[3, :3] = -campos + np.array([0.0, 0.0, 4.0], np.float32)  # row-vector convention: translation in the last row
    tan = 0.5
    W = H = 64
    proj = view @ np.array([[1 / tan, 0, 0, 0], [0, 1 / tan, 0, 0], [0, 0, 1.0, 1.0], [0, 0, -0.02, 0]], np.float32)
    f = O.forward(pos, np.full((P, 1), 0.5, np.float32), np.zeros(3, np.float32), view, proj, campos, W, H, tan, tan,
                  shs=sh[:, :M].copy(), sh_degree=deg, scales=np.full((P, 3), 0.01, np.float32),
                  rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1)))
    assert (f["radii"] > 0).all(), "every golden point must be rendered for its colour to be computed"
    want = G[f"rgb{deg}"]
    assert np.abs(f["rgb"] - want).max() <= 2e-6
    assert ((want == 0) == (f["clamped"] != 0)).all() or deg == 0  # the clamp flags are the zeros of the golden colours
    # backward of the SH stage alone: feed dL/dcolour per splat straight into the per-Gaussian backward
    L = O.lib()
    g = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
             dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcolors=G[f"w{deg}"].astype(np.float32).copy(),
             dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
             dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32))
    i = f["_inputs"]
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    L.fnx_oracle_preprocess_backward(
        C.c_int(P), C.c_int(deg), C.c_int(M), p(i["means3D"]), p(f["radii"]), p(i["shs"]), p(f["clamped"]), p(i["scales"]),
        p(i["rotations"]), C.c_float(1.0), p(f["cov3D"]), p(i["viewmatrix"]), p(i["projmatrix"]), C.c_int(W), C.c_int(H),
        C.c_float(tan), C.c_float(tan), p(i["campos"]), p(g["dL_dmeans2D"]), p(g["dL_dconic"]), p(g["dL_dmeans3D"]),
        p(g["dL_dcolors"]), p(g["dL_dcov3D"]), p(g["dL_dsh"]), p(g["dL_dscales"]), p(g["dL_drotations"]))
    ref_sh = G[f"dsh{deg}"][:, :M]
    assert np.abs(g["dL_dsh"] - ref_sh).max() <= 1e-5 * max(np.abs(ref_sh).max(), 1.0)
    ref_pos = G[f"dpos{deg}"]
    # with zero screen-space gradients the mean gradient is the view-direction term alone (backward.cu:125-131)
    assert np.abs(g["dL_dmeans3D"] - ref_pos).max() <= 2e-5 * max(np.abs(ref_pos).max(), 1.0)
    if deg > 0:
        assert np.abs(ref_pos).max() > 0


def test_public_headers_compile_as_c():
    """include/*.h are the C ABI: a C translation unit that includes them and takes the address of every declared
    entry point must compile with a C compiler in pedantic C11 mode."""
    from fluidnexus_amd import _lib, _physics_lib, losses
    names = list(_lib.SYMBOLS) + list(_physics_lib.SYMBOLS) + list(losses.SYMBOLS)
    src = "#include \"fnx_raster.h\"\n#include \"fnx_physics.h\"\n#include \"fnx_losses.h\"\n" \
          "typedef void (*fn)(void);\nfn table[] = {\n" + "".join(f"    (fn){n},\n" for n in names) + "};\n" \
          "int main(void) { fnx_geom_layout_t g; fnx_image_layout_t i; fnx_binning_layout_t b; fnx_static_layout_t s;\n" \
          "    (void)g; (void)i; (void)b; (void)s; return (int)(sizeof(table) / sizeof(table[0])) == 0; }\n"
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "abi.c")
        open(path, "w").write(src)
        r = subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Werror", "-Wno-pedantic", "-I", os.path.join(ROOT, "include"),
                            "-c", path, "-o", os.path.join(d, "abi.o")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
