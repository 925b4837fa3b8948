"""ctypes binding of libfnx_raster.so (the C ABI in include/fnx_raster.h).

There is deliberately no CPU or PyTorch fallback: if the HIP library is missing or cannot be
loaded, every entry point raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_RASTER = None

c_void_p, c_int, c_float, c_int64, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t

FNX_OK = 0
FNX_MAX_VIEWS = 16
FNX_ERR_INVALID_ARG = 1
FNX_ERR_NON_RGB_NEEDS_COLORS = 2
FNX_ERR_HIP = 3
FNX_ERR_CAPACITY = 4
FNX_ERR_UNSUPPORTED = 5
FNX_ERR_SORT_SPAN = 6


class GeomLayout(C.Structure):
    _fields_ = [(n, c_size_t) for n in
                ("depths", "clamped", "radii", "means2D", "cov3D", "conic_opacity", "rgb", "tiles_touched", "sort_key0",
                 "sort_key1", "sort_val0", "sort_val1", "rect", "rect_sorted", "krec", "sort_hist", "blk_hist", "blk_rel", "blend_rec", "total")]


class ImageLayout(C.Structure):
    _fields_ = [(n, c_size_t) for n in
                ("header", "final_T", "n_contrib", "ranges", "tile_count", "dyn_start", "acc_final", "tile_order", "tile_deep", "total")]


class BinningLayout(C.Structure):
    _fields_ = [(n, c_size_t) for n in ("point_list", "pairs", "bstate", "bwd_items", "block_masks", "total")]


class StaticLayout(C.Structure):
    _fields_ = [(n, c_size_t) for n in ("header", "starts", "radii", "blend_rec", "pairs", "total")]


ALLOC_FN = C.CFUNCTYPE(c_void_p, c_size_t, c_void_p)

FNX_OPT_DEFAULT = -2147483648
FNX_SORT_FULL, FNX_SORT_NARROW, FNX_SORT_COHERENT = 0, 1, 2


class RasterDual(C.Structure):
    """fnx_raster_dual_t (include/fnx_raster.h): the second, single-channel image of a dual-mode view batch."""
    _fields_ = [("image_buffers1", c_void_p), ("background1", c_void_p), ("out_color1", c_void_p),
                ("out_depth1", c_void_p), ("dL_dpix1", c_void_p)]


class RasterOpts(C.Structure):
    """fnx_raster_opts_t (include/fnx_raster.h): the options of ONE call."""
    _fields_ = [("size", C.c_uint32), ("blend_math", C.c_int32), ("lean_geometry", C.c_int32), ("sort_mode", C.c_int32),
                ("deep_kernel", C.c_int32), ("grad_splat_limit", C.c_int32), ("deep_threshold", C.c_uint32),
                ("reserved0", C.c_uint32), ("zero3", c_void_p), ("sort_state", c_void_p),
                ("dual", C.POINTER(RasterDual)), ("segment_scratch", c_void_p)]


def make_opts(blend_math=0, lean_geometry=0, sort_mode=FNX_SORT_FULL, deep_kernel=0, grad_splat_limit=-1,
              deep_threshold=0, zero3=None, sort_state=None, dual=None, segment_scratch=None) -> RasterOpts:
    o = RasterOpts()
    o.size = C.sizeof(RasterOpts)
    o.blend_math, o.lean_geometry, o.sort_mode, o.deep_kernel = int(blend_math), int(lean_geometry), int(sort_mode), int(deep_kernel)
    o.grad_splat_limit, o.deep_threshold = int(grad_splat_limit), int(deep_threshold)
    o.zero3, o.sort_state = zero3, sort_state
    o.segment_scratch = segment_scratch
    if dual is not None:  # a RasterDual the caller keeps alive for the duration of the call
        o.dual = C.pointer(dual)
    return o


def make_dual(image_buffers1, background1, out_color1=None, out_depth1=None, dL_dpix1=None) -> RasterDual:
    d = RasterDual()
    d.image_buffers1, d.background1, d.out_color1, d.out_depth1, d.dL_dpix1 = (image_buffers1, background1, out_color1,
                                                                               out_depth1, dL_dpix1)
    return d

# every symbol include/fnx_raster.h declares (tests check the library exports all of them)
def raw_stream() -> int:
    """hipStream_t of torch's current stream as an integer.  torch.cuda.current_stream().cuda_stream builds a Stream object
    and walks through torch.cuda.is_available() on the way (4-5 us a call, sixteen calls per loop iteration: a fifth of the
    host time of an eager iteration, tools/iter_host_profile.py); the two private getters below take 0.3 us."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


SYMBOLS = (
    "fnx_abi_version", "fnx_last_error", "fnx_geom_bytes", "fnx_image_bytes", "fnx_binning_bytes",
    "fnx_rasterize_forward", "fnx_forward_stage1", "fnx_read_num_rendered", "fnx_forward_stage2", "fnx_read_status",
    "fnx_rasterize_backward", "fnx_rasterize_backward_ex", "fnx_mark_visible", "fnx_geom_layout", "fnx_image_layout", "fnx_binning_layout",
    "fnx_profile_enable", "fnx_profile_read",
    "fnx_forward_stage1_views", "fnx_forward_stage2_views", "fnx_forward_stage2_views_status",
    "fnx_rasterize_backward_views",
    "fnx_static_bytes", "fnx_binning_bytes_split", "fnx_static_finalize_views", "fnx_forward_stage1_views_split",
    "fnx_forward_stage2_views_split", "fnx_rasterize_backward_views_split", "fnx_binning_layout_split", "fnx_static_layout",
    "fnx_set_deep_threshold", "fnx_set_blend_math", "fnx_get_blend_math", "fnx_set_deep_kernel", "fnx_set_lean_geometry", "fnx_set_sort_narrow", "fnx_request_zero3", "fnx_request_gradient_limit",
    "fnx_forward_stage1_views_split_opts", "fnx_forward_stage2_views_split_opts", "fnx_rasterize_backward_views_split_opts",
    "fnx_sort_state_bytes", "fnx_sort_state_read", "fnx_sort_state_outliers", "fnx_binning_bytes_dual",
    "fnx_segment_scratch_bytes", "fnx_segment_scratch_read", "fnx_set_backward_form", "fnx_get_backward_form",
)

# Version of the C ABI this binding was written against (include/fnx_raster.h FNX_ABI_VERSION): the layouts of the
# scratch blobs and several argument lists changed since version 1, and a stale library would read garbage silently.
ABI_VERSION = 7


def raster_path() -> str:
    # FNX_RASTER_LIB: developer override used for kernel timing experiments (tools/)
    return os.environ.get("FNX_RASTER_LIB") or os.path.join(_HERE, "libfnx_raster.so")


def raster():
    """Load libfnx_raster.so; raises RuntimeError when it has not been built."""
    global _RASTER
    if _RASTER is not None:
        return _RASTER
    path = raster_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build the HIP extension first (python -m fluidnexus_amd.build). "
            "fluidnexus_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64: loaded AFTER this library it would be a second HIP runtime in the process
    # (this library bound to the system one), and device memory of one runtime is unknown to the other ("no
    # ROCm-capable device is detected").  Loading torch first makes the dependency resolve to the runtime torch uses.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    p, i, f = c_void_p, c_int, c_float
    lib.fnx_abi_version.restype = i
    if lib.fnx_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path} exports C ABI version {lib.fnx_abi_version()}, this package needs {ABI_VERSION}: "
                           "rebuild it (python -m fluidnexus_amd.build --force)")
    lib.fnx_last_error.restype = C.c_char_p
    lib.fnx_geom_bytes.restype = c_size_t
    lib.fnx_geom_bytes.argtypes = [i, i, i]
    lib.fnx_image_bytes.restype = c_size_t
    lib.fnx_image_bytes.argtypes = [i, i]
    lib.fnx_binning_bytes.restype = c_size_t
    lib.fnx_binning_bytes.argtypes = [c_int64]
    lib.fnx_forward_stage1.restype = i
    lib.fnx_forward_stage1.argtypes = [i, p, p, i, i, i, i, i, p, p, p, p, p, f, p, p, p, p, p, f, f, i, p, p]
    lib.fnx_read_num_rendered.restype = i
    lib.fnx_read_num_rendered.argtypes = [p, i, i, p, C.POINTER(i)]
    lib.fnx_read_status.restype = i
    lib.fnx_read_status.argtypes = [p, i, i, p]
    lib.fnx_forward_stage2.restype = i
    lib.fnx_forward_stage2.argtypes = [i, p, p, c_int64, p, i, i, i, p, p, p, p, p, p]
    lib.fnx_rasterize_forward.restype = i
    lib.fnx_rasterize_forward.argtypes = [i, ALLOC_FN, p, ALLOC_FN, p, ALLOC_FN, p, i, i, i, p, i, i, p, p, p, p, p, f,
                                          p, p, p, p, p, f, f, i, p, p, p, p, C.POINTER(i)]
    lib.fnx_rasterize_backward.restype = i
    lib.fnx_rasterize_backward.argtypes = [i, i, i, i, i, p, i, i, p, p, p, p, f, p, p, p, p, p, f, f, p, p, p, p, p,
                                           p, p, p, p, p, p, p, p, p, p]
    lib.fnx_rasterize_backward_ex.restype = i
    lib.fnx_rasterize_backward_ex.argtypes = lib.fnx_rasterize_backward.argtypes[:-1] + [i, i, p]
    fp = C.POINTER(f)  # host arrays of per-view tan(fov/2)
    lib.fnx_forward_stage1_views.restype = i
    lib.fnx_forward_stage1_views.argtypes = [i, i, p, p, i, i, i, i, i, p, p, p, p, p, f, p, p, p, p, p, fp, fp, i, p, p]
    lib.fnx_forward_stage2_views.restype = i
    lib.fnx_forward_stage2_views.argtypes = [i, i, p, p, c_int64, p, i, i, i, p, p, p, p, p]
    lib.fnx_forward_stage2_views_status.restype = i
    lib.fnx_forward_stage2_views_status.argtypes = [i, i, p, p, c_int64, p, i, i, i, p, p, p, p, p, p]
    lib.fnx_rasterize_backward_views.restype = i
    lib.fnx_rasterize_backward_views.argtypes = [i, i, i, i, i, p, i, i, p, p, p, p, f, p, p, p, p, p, fp, fp, p,
                                                 p, p, c_int64, p, p, p, p, p, p, p, p, p, p, p, p, p, i, i, p]
    # static-split extension
    lib.fnx_static_bytes.restype = c_size_t
    lib.fnx_static_bytes.argtypes = [i, i, i, c_int64]
    lib.fnx_binning_bytes_split.restype = c_size_t
    lib.fnx_binning_bytes_split.argtypes = [c_int64, c_int64]
    lib.fnx_binning_bytes_dual.restype = c_size_t
    lib.fnx_binning_bytes_dual.argtypes = [c_int64, c_int64]
    lib.fnx_static_finalize_views.restype = i
    lib.fnx_static_finalize_views.argtypes = [i, p, p, p, i, i, i, i, c_int64, p, p, p]
    lib.fnx_forward_stage1_views_split.restype = i
    lib.fnx_forward_stage1_views_split.argtypes = lib.fnx_forward_stage1_views.argtypes[:-1] + [p, i, c_int64, p, p]
    lib.fnx_forward_stage2_views_split.restype = i
    lib.fnx_forward_stage2_views_split.argtypes = [i, i, p, p, c_int64, p, i, i, i, p, p, p, p, p, i, c_int64, i, p, p]
    lib.fnx_set_deep_threshold.restype = i
    lib.fnx_set_deep_threshold.argtypes = [C.c_uint]
    lib.fnx_set_blend_math.restype = i
    lib.fnx_set_blend_math.argtypes = [i]
    lib.fnx_get_blend_math.restype = i
    lib.fnx_set_deep_kernel.restype = i
    lib.fnx_set_lean_geometry.restype = i
    lib.fnx_set_sort_narrow.restype = i
    lib.fnx_request_zero3.restype = i
    lib.fnx_request_zero3.argtypes = [p]
    lib.fnx_request_gradient_limit.restype = i
    lib.fnx_request_gradient_limit.argtypes = [i]
    lib.fnx_set_sort_narrow.argtypes = [i]
    lib.fnx_set_lean_geometry.argtypes = [i]
    lib.fnx_set_deep_kernel.argtypes = [i]
    lib.fnx_set_backward_form.restype = i
    lib.fnx_set_backward_form.argtypes = [i]
    lib.fnx_get_backward_form.restype = i
    lib.fnx_rasterize_backward_views_split.restype = i
    lib.fnx_rasterize_backward_views_split.argtypes = lib.fnx_rasterize_backward_views.argtypes[:-1] + [p, i, c_int64, p]
    op = C.POINTER(RasterOpts)
    lib.fnx_forward_stage1_views_split_opts.restype = i
    lib.fnx_forward_stage1_views_split_opts.argtypes = lib.fnx_forward_stage1_views_split.argtypes[:-1] + [op, p]
    lib.fnx_forward_stage2_views_split_opts.restype = i
    lib.fnx_forward_stage2_views_split_opts.argtypes = lib.fnx_forward_stage2_views_split.argtypes[:-1] + [op, p]
    lib.fnx_rasterize_backward_views_split_opts.restype = i
    lib.fnx_rasterize_backward_views_split_opts.argtypes = lib.fnx_rasterize_backward_views_split.argtypes[:-1] + [p, op, p]
    lib.fnx_segment_scratch_bytes.restype = c_size_t
    lib.fnx_segment_scratch_bytes.argtypes = [i, i]
    lib.fnx_segment_scratch_read.restype = i
    lib.fnx_segment_scratch_read.argtypes = [p, i, i, i, p, C.POINTER(C.c_uint32)]
    lib.fnx_sort_state_bytes.restype = c_size_t
    lib.fnx_sort_state_bytes.argtypes = [i]
    lib.fnx_sort_state_outliers.restype = i
    lib.fnx_sort_state_outliers.argtypes = [p, i, i, p, C.POINTER(C.c_uint32)]
    lib.fnx_sort_state_read.restype = i
    lib.fnx_sort_state_read.argtypes = [p, i, i, p, C.POINTER(C.c_uint32)]
    lib.fnx_binning_layout_split.argtypes = [c_int64, c_int64, C.POINTER(BinningLayout)]
    lib.fnx_static_layout.argtypes = [i, i, i, c_int64, C.POINTER(StaticLayout)]
    lib.fnx_mark_visible.restype = i
    lib.fnx_mark_visible.argtypes = [i, p, p, p, p, p]
    lib.fnx_profile_enable.restype = i
    lib.fnx_profile_enable.argtypes = [i]
    lib.fnx_profile_read.restype = i
    lib.fnx_profile_read.argtypes = [i, C.POINTER(C.c_double), C.POINTER(i)]
    lib.fnx_geom_layout.argtypes = [i, i, i, C.POINTER(GeomLayout)]
    lib.fnx_image_layout.argtypes = [i, i, C.POINTER(ImageLayout)]
    lib.fnx_binning_layout.argtypes = [c_int64, C.POINTER(BinningLayout)]
    _RASTER = lib
    return lib


class FnxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


def check(rc: int):
    if rc != FNX_OK:
        msg = raster().fnx_last_error().decode("utf-8", "replace")
        raise FnxError(rc, msg)


def profile_enable(on: bool):
    check(raster().fnx_profile_enable(1 if on else 0))


def profile_read(which: int):
    """(total_ms, launches) of kernel class `which` since profile_enable(True)."""
    ms, n = C.c_double(0.0), c_int(0)
    check(raster().fnx_profile_read(which, C.byref(ms), C.byref(n)))
    return ms.value, n.value


def geom_layout(P: int, W: int, H: int) -> GeomLayout:
    L = GeomLayout()
    raster().fnx_geom_layout(P, W, H, C.byref(L))
    return L


def image_layout(W: int, H: int) -> ImageLayout:
    L = ImageLayout()
    raster().fnx_image_layout(W, H, C.byref(L))
    return L


def binning_layout(R: int, R_static: int | None = None) -> BinningLayout:
    """Layout of a binning blob; `R_static` (not None) selects the static-split layout."""
    L = BinningLayout()
    if R_static is None:
        raster().fnx_binning_layout(R, C.byref(L))
    else:
        raster().fnx_binning_layout_split(R, R_static, C.byref(L))
    return L


def static_layout(P_static: int, W: int, H: int, R_static: int) -> StaticLayout:
    L = StaticLayout()
    raster().fnx_static_layout(P_static, W, H, R_static, C.byref(L))
    return L
