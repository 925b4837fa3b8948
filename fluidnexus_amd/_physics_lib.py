"""ctypes binding of libfnx_physics.so (include/fnx_physics.h).  No CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ABI_VERSION = 3  # include/fnx_physics.h FNX_PHYSICS_ABI_VERSION

SYMBOLS = ("fnx_physics_abi_version", "fnx_physics_last_error", "fnx_grid_bytes", "fnx_grid_build",
           "fnx_density_forward", "fnx_density_backward", "fnx_visual_interp_forward", "fnx_visual_interp_backward",
           "fnx_physical_stage", "fnx_adam_step", "fnx_pbf_predict", "fnx_pbf_neighbor_counts", "fnx_pbf_project",
           "fnx_pbf_confirm", "fnx_visual_advect", "fnx_knn_mean_dist2", "fnx_visual_interp_forward_cells", "fnx_grid_cell_items_bytes",
           "fnx_grid_cell_items", "fnx_visual_interp_backward_cells", "fnx_distance_loss", "fnx_distance_loss_partials",
           "fnx_visual_interp_forward_cells_div", "fnx_visual_interp_backward_cells_sum", "fnx_adam_step_grid",
           "fnx_visual_interp_forward_cells_vel", "fnx_distance_table_bytes", "fnx_distance_loss_lists", "fnx_stream_delay",
           "fnx_knn_cut", "fnx_density_forward_kcap", "fnx_density_backward_kcap", "fnx_visual_interp_forward_kcap",
           "fnx_visual_interp_backward_kcap", "fnx_distance_verlet_bytes", "fnx_distance_loss_verlet", "fnx_knn_watch")


def physics():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("FNX_PHYSICS_LIB") or os.path.join(_HERE, "libfnx_physics.so")  # env: developer variants
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build the HIP extension first (python -m fluidnexus_amd.build). "
                           "fluidnexus_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64: loaded AFTER this library it would be a second HIP runtime in the process
    # (this library bound to the system one), and device memory of one runtime is unknown to the other ("no
    # ROCm-capable device is detected").  Loading torch first makes the dependency resolve to the runtime torch uses.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    p, i, f = C.c_void_p, C.c_int, C.c_float
    lib.fnx_physics_abi_version.restype = i
    if lib.fnx_physics_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path} exports C ABI version {lib.fnx_physics_abi_version()}, this package needs "
                           f"{ABI_VERSION}: rebuild it (python -m fluidnexus_amd.build --force)")
    lib.fnx_physics_last_error.restype = C.c_char_p
    lib.fnx_grid_bytes.restype = C.c_size_t
    lib.fnx_grid_bytes.argtypes = [i]
    lib.fnx_grid_build.restype = i
    lib.fnx_grid_build.argtypes = [p, i, f, p, p]
    lib.fnx_distance_loss_partials.restype = i
    lib.fnx_distance_loss_partials.argtypes = [i]
    lib.fnx_distance_loss.restype = i
    lib.fnx_distance_loss.argtypes = [p, i, f, p, p, p, p]
    lib.fnx_stream_delay.restype = i
    lib.fnx_stream_delay.argtypes = [f, p]
    lib.fnx_distance_table_bytes.restype = C.c_size_t
    lib.fnx_distance_table_bytes.argtypes = [i]
    lib.fnx_distance_loss_lists.restype = i
    lib.fnx_distance_loss_lists.argtypes = [p, i, f, p, p, p, p]
    lib.fnx_knn_watch.restype = i
    lib.fnx_knn_watch.argtypes = [p, i]
    lib.fnx_distance_verlet_bytes.restype = C.c_size_t
    lib.fnx_distance_verlet_bytes.argtypes = [i, i]
    lib.fnx_distance_loss_verlet.restype = i
    lib.fnx_distance_loss_verlet.argtypes = [p, i, f, f, p, p, i, p, p, p]
    lib.fnx_density_forward.restype = i
    lib.fnx_density_forward.argtypes = [p, i, p, f, f, p, p, p]
    lib.fnx_density_backward.restype = i
    lib.fnx_density_backward.argtypes = [p, i, p, f, f, p, p, p, p]
    lib.fnx_visual_interp_forward.restype = i
    lib.fnx_visual_interp_forward.argtypes = [p, i, p, p, i, f, f, f, p, p, p, p, p]
    lib.fnx_visual_interp_forward_cells.restype = i
    lib.fnx_visual_interp_forward_cells.argtypes = [p, i, p, p, i, f, f, f, p, p, p, p, p, p, p]
    lib.fnx_visual_interp_forward_cells_div.restype = i
    lib.fnx_visual_interp_forward_cells_div.argtypes = [p, i, p, p, i, f, f, f, p, p, p, p, p, p, p, f, p]
    lib.fnx_visual_interp_forward_cells_vel.restype = i
    lib.fnx_visual_interp_forward_cells_vel.argtypes = [p, i, p, p, i, f, f, f, p, p, p, p, p, p, p, f, i, p]
    lib.fnx_adam_step_grid.restype = i
    lib.fnx_adam_step_grid.argtypes = [p, i, p, f, p, f, p, f, f, p, p, p, f, C.c_double, C.c_double, f, p, p, f, p,
                                       f, p, p, f, p]
    lib.fnx_grid_cell_items_bytes.restype = C.c_size_t
    lib.fnx_grid_cell_items_bytes.argtypes = [i]
    lib.fnx_grid_cell_items.restype = i
    lib.fnx_grid_cell_items.argtypes = [p, i, p, p]
    lib.fnx_knn_cut.restype = i
    lib.fnx_knn_cut.argtypes = [p, i, i, f, i, p, p, p]
    lib.fnx_density_forward_kcap.restype = i
    lib.fnx_density_forward_kcap.argtypes = [p, i, p, f, f, p, p, p, p]
    lib.fnx_density_backward_kcap.restype = i
    lib.fnx_density_backward_kcap.argtypes = [p, i, p, f, f, p, p, p, p, p]
    lib.fnx_visual_interp_forward_kcap.restype = i
    lib.fnx_visual_interp_forward_kcap.argtypes = [p, i, p, p, i, f, f, f, p, p, p, p, p, p]
    lib.fnx_visual_interp_backward_kcap.restype = i
    lib.fnx_visual_interp_backward_kcap.argtypes = [p, i, p, p, i, f, f, f, p, p, p, p, p, p, p]
    lib.fnx_visual_interp_backward.restype = i
    lib.fnx_visual_interp_backward.argtypes = [p, i, p, p, i, f, f, f, p, p, p, p, p, p]
    lib.fnx_visual_interp_backward_cells.restype = i
    lib.fnx_visual_interp_backward_cells.argtypes = [p, i, p, p, i, f, f, f, p, p, p, p, p, p, p, p]
    lib.fnx_visual_interp_backward_cells_sum.restype = i
    lib.fnx_visual_interp_backward_cells_sum.argtypes = [p, i, p, p, i, f, f, f, p, p, p, p, p, p, p, f, p, p]
    lib.fnx_physical_stage.restype = i
    lib.fnx_physical_stage.argtypes = [p, i, f, p, p, p, p, p, f, f, f, f, f, f, f, p, i, p, p, p, p, p, p]
    fp3 = C.POINTER(C.c_float)
    lib.fnx_pbf_predict.restype = i
    lib.fnx_pbf_predict.argtypes = [p, p, p, p, p, p, i, fp3, f, f, f, f, p]
    lib.fnx_pbf_neighbor_counts.restype = i
    lib.fnx_pbf_neighbor_counts.argtypes = [p, i, f, p, p, p]
    lib.fnx_pbf_project.restype = i
    lib.fnx_pbf_project.argtypes = [p, p, p, p, p, i, f, f, f, f, f, f, f, f, p, p, p]
    lib.fnx_pbf_confirm.restype = i
    lib.fnx_pbf_confirm.argtypes = [p, p, p, i, f, f, p]
    lib.fnx_visual_advect.restype = i
    lib.fnx_visual_advect.argtypes = [p, i, p, p, i, f, f, f, p, p, p]
    lib.fnx_knn_mean_dist2.restype = i
    lib.fnx_knn_mean_dist2.argtypes = [p, i, f, p, p, p]
    lib.fnx_adam_step.restype = i
    lib.fnx_adam_step.argtypes = [p, i, p, f, p, f, p, f, f, p, p, p, f, C.c_double, C.c_double, f, p, p, f, p, p]
    _LIB = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise RuntimeError(physics().fnx_physics_last_error().decode("utf-8", "replace"))
