"""ctypes/numpy front-end of oracle/raster_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It restates Rasterizer::forward/backward of
FluidDynamics/submodules/gaussian_rasterization_ch3/cuda_rasterizer/rasterizer_impl.cu:184-414
stage by stage and exposes every intermediate the reference keeps in its three scratch buffers.
PARITY STATUS: parity unpinned for the CUDA kernels (see the header of raster_oracle.c).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "raster_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def _load(so):
    L = C.CDLL(so)
    L.fnx_oracle_preprocess.restype = C.c_int64
    L.fnx_oracle_expf.restype = C.c_float
    L.fnx_oracle_expf.argtypes = [C.c_float]
    L.fnx_oracle_max_threads.restype = C.c_int
    return L


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = _load(so)
    return _LIB


def use_variant(name=None):
    """Switch every later call of this module to another build of the same source: "fma" = liboracle_fma.so
    (-ffp-contract=fast -mfma, built on demand; oracle/fp_contract_report.py), None = the parity checker."""
    global _LIB
    if name is None:
        _LIB = None
        return lib()
    so = os.path.join(_HERE, f"liboracle_{name}.so")
    src = os.path.join(_HERE, "raster_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(so)], stdout=subprocess.DEVNULL)
    _LIB = _load(so)
    return _LIB


def set_threads(n: int):
    lib().fnx_oracle_set_threads(C.c_int(int(n)))


def max_threads() -> int:
    return int(lib().fnx_oracle_max_threads())


def expf(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32)
    out = np.empty_like(x)
    f = lib().fnx_oracle_expf
    flat_in, flat_out = x.ravel(), out.ravel()
    for i in range(flat_in.size):
        flat_out[i] = f(C.c_float(float(flat_in[i])))
    return out


def _p(a):
    """numpy array (or None) -> void* (NULL for None: the reference's nullptr for 'not provided')."""
    if a is None:
        return C.c_void_p(0)
    return a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if a.size == 0:
        return None
    if shape is not None:
        a = a.reshape(shape)
    return a


def forward(means3D, opacities, bg, viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, *, colors_precomp=None,
            shs=None, sh_degree=0, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0, channels=3):
    """Rasterizer::forward (rasterizer_impl.cu:184-319).  Returns a dict with the outputs
    (color [C,H,W], depth [1,H,W], radii [P]) and every intermediate."""
    L = lib()
    means3D = _f32(means3D)
    P = 0 if means3D is None else means3D.shape[0]
    Cn = int(channels)
    colors_precomp = _f32(colors_precomp)
    shs = _f32(shs)
    scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    opacities = _f32(opacities)
    bg = _f32(bg)
    viewmatrix, projmatrix, campos = _f32(viewmatrix), _f32(projmatrix), _f32(campos)
    if Cn != 3 and colors_precomp is None:
        # rasterizer_impl.cu:226-228
        raise RuntimeError("For non-RGB, provide precomputed Gaussian colors!")
    M = 0 if shs is None else shs.shape[1]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    out = dict(P=P, W=W, H=H, C=Cn, M=M, D=int(sh_degree))
    out["color"] = np.zeros((Cn, H, W), np.float32)
    out["depth"] = np.zeros((1, H, W), np.float32)
    out["radii"] = np.zeros((P,), np.int32)
    out["means2D"] = np.zeros((P, 2), np.float32)
    out["depths"] = np.zeros((P,), np.float32)
    out["cov3D"] = np.zeros((P, 6), np.float32)
    out["rgb"] = np.zeros((P, 3), np.float32)
    out["conic_opacity"] = np.zeros((P, 4), np.float32)
    out["clamped"] = np.zeros((P, 3), np.uint8)
    out["tiles_touched"] = np.zeros((P,), np.uint32)
    out["point_offsets"] = np.zeros((P,), np.uint32)
    out["ranges"] = np.zeros((T, 2), np.uint32)
    out["final_T"] = np.zeros((H, W), np.float32)
    out["n_contrib"] = np.zeros((H, W), np.uint32)
    out["examined"] = np.zeros((H, W), np.uint32)
    if P == 0:  # rasterize_points.cu:81
        out["num_rendered"] = 0
        out["keys_sorted"] = np.zeros((0,), np.uint64)
        out["point_list"] = np.zeros((0,), np.uint32)
        return out
    R = L.fnx_oracle_preprocess(
        C.c_int(Cn), C.c_int(P), C.c_int(int(sh_degree)), C.c_int(M), _p(means3D), _p(scales),
        C.c_float(scale_modifier), _p(rotations), _p(opacities), _p(shs), _p(cov3D_precomp), _p(colors_precomp),
        _p(viewmatrix), _p(projmatrix), _p(campos), C.c_int(W), C.c_int(H), C.c_float(tan_fovx), C.c_float(tan_fovy),
        _p(out["radii"]), _p(out["means2D"]), _p(out["depths"]), _p(out["cov3D"]), _p(out["rgb"]),
        _p(out["conic_opacity"]), _p(out["clamped"]), _p(out["tiles_touched"]), _p(out["point_offsets"]))
    R = int(R)
    out["num_rendered"] = R
    out["keys_sorted"] = np.zeros((R,), np.uint64)
    out["point_list"] = np.zeros((R,), np.uint32)
    L.fnx_oracle_bin(C.c_int(P), C.c_int(W), C.c_int(H), _p(out["means2D"]), _p(out["depths"]),
                     _p(out["point_offsets"]), _p(out["radii"]), C.c_int64(R), _p(out["keys_sorted"]),
                     _p(out["point_list"]), _p(out["ranges"]))
    feats = colors_precomp if colors_precomp is not None else out["rgb"]
    out["features"] = feats
    L.fnx_oracle_render(C.c_int(Cn), C.c_int(W), C.c_int(H), _p(out["ranges"]), _p(out["point_list"]),
                        _p(out["means2D"]), _p(feats), _p(out["conic_opacity"]), _p(out["depths"]), _p(bg),
                        _p(out["final_T"]), _p(out["n_contrib"]), _p(out["color"]), _p(out["depth"]),
                        _p(out["examined"]))
    out["_inputs"] = dict(means3D=means3D, opacities=opacities, bg=bg, viewmatrix=viewmatrix, projmatrix=projmatrix,
                          campos=campos, tan_fovx=tan_fovx, tan_fovy=tan_fovy, colors_precomp=colors_precomp, shs=shs,
                          scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
                          scale_modifier=scale_modifier)
    return out


def backward(fwd: dict, dL_dcolor):
    """Rasterizer::backward (rasterizer_impl.cu:323-414) on the state returned by forward().
    Returns the eight gradients of rasterize_points.cu:150-158 (+ dL_dconic)."""
    L = lib()
    P, W, H, Cn, M, D = fwd["P"], fwd["W"], fwd["H"], fwd["C"], fwd["M"], fwd["D"]
    g = dict(
        dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dcolors=np.zeros((P, Cn), np.float32),
        dL_dopacity=np.zeros((P, 1), np.float32), dL_dmeans3D=np.zeros((P, 3), np.float32),
        dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
        dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32),
        dL_dconic=np.zeros((P, 2, 2), np.float32))
    if P == 0:
        return g
    i = fwd["_inputs"]
    dL = np.ascontiguousarray(np.asarray(dL_dcolor, np.float32).reshape(Cn, H, W))
    L.fnx_oracle_render_backward(
        C.c_int(Cn), C.c_int(P), C.c_int(W), C.c_int(H), _p(fwd["ranges"]), _p(fwd["point_list"]), _p(i["bg"]),
        _p(fwd["means2D"]), _p(fwd["conic_opacity"]), _p(fwd["features"]), _p(fwd["final_T"]), _p(fwd["n_contrib"]),
        _p(dL), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]))
    cov3D = i["cov3D_precomp"] if i["cov3D_precomp"] is not None else fwd["cov3D"]
    L.fnx_oracle_preprocess_backward(
        C.c_int(P), C.c_int(D), C.c_int(M), _p(i["means3D"]), _p(fwd["radii"]), _p(i["shs"]), _p(fwd["clamped"]),
        _p(i["scales"]), _p(i["rotations"]), C.c_float(i["scale_modifier"]), _p(cov3D), _p(i["viewmatrix"]),
        _p(i["projmatrix"]), C.c_int(W), C.c_int(H), C.c_float(i["tan_fovx"]), C.c_float(i["tan_fovy"]),
        _p(i["campos"]), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcolors"]),
        _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _f32(means3D)
    P = 0 if means3D is None else means3D.shape[0]
    out = np.zeros((P,), np.uint8)
    if P:
        lib().fnx_oracle_mark_visible(C.c_int(P), _p(means3D), _p(_f32(viewmatrix)), _p(_f32(projmatrix)), _p(out))
    return out.astype(bool)
