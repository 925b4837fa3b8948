"""Developer experiment: build a variant of the raster library whose blend forward can be restricted to
one tile (env FNX_ONLY_TILE) to measure single-tile latency.  Output: build/exp/libexp.so"""
import os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "fluidnexus_amd", "csrc")
E = os.path.join(R, "build", "exp")
os.makedirs(E, exist_ok=True)
src = open(os.path.join(C, "raster_forward.hip")).read()
src = src.replace("    const int tile = xcd_tile(blockIdx.x, T);\n    const int tx = tile % gx, ty = tile / gx;\n    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;\n    const int px = tx * FNX_TILE_X + (w & 1) * 8",
                  "    const int tile = (only_tile >= 0) ? only_tile : xcd_tile(blockIdx.x, T);\n    const int tx = tile % gx, ty = tile / gx;\n    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;\n    const int px = tx * FNX_TILE_X + (w & 1) * 8", 1)
src = src.replace("float *__restrict__ out_color, float *__restrict__ out_depth, const uint32_t *__restrict__ header,\n                     uint32_t capacity) {\n    __shared__ float4 s_ra[256];",
                  "float *__restrict__ out_color, float *__restrict__ out_depth, const uint32_t *__restrict__ header,\n                     uint32_t capacity, int only_tile) {\n    __shared__ float4 s_ra[256];", 1)
src = src.replace("dim3(T), dim3(256), 0, s, T, gx, ranges, point_list, W, H,\n                           blend_rec, bg, final_T, n_contrib, out_color, out_depth, header, capacity);",
                  "dim3(only >= 0 ? 1 : T), dim3(256), 0, s, T, gx, ranges, point_list, W, H,\n                           blend_rec, bg, final_T, n_contrib, out_color, out_depth, header, capacity, only);")
src = src.replace("    const int gx = tiles_x(W), T = gx * tiles_y(H);\n    if (C == 3)\n        hipLaunchKernelGGL((blend_forward_kernel<3>)",
                  "    const int gx = tiles_x(W), T = gx * tiles_y(H);\n    const char *e = getenv(\"FNX_ONLY_TILE\");\n    const int only = e ? atoi(e) : -1;\n    if (C == 3)\n        hipLaunchKernelGGL((blend_forward_kernel<3>)", 1)
src = src.replace('#include "fnx_state.h"\n', '#include "fnx_state.h"\n#include <cstdlib>\n', 1)
assert "only_tile" in src and "FNX_ONLY_TILE" in src and "capacity, only);" in src
open(os.path.join(E, "raster_forward.hip"), "w").write(src)
for h in ("fnx_device.h", "fnx_state.h"):
    open(os.path.join(E, h), "w").write(open(os.path.join(C, h)).read().replace('"../../include/', '"../../include/'))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                       "-Wno-unused-value", "-I" + C, "-o", os.path.join(E, "libexp.so"), os.path.join(E, "raster_forward.hip"),
                       os.path.join(C, "raster_binning.hip"), os.path.join(C, "raster_backward.hip"), os.path.join(C, "raster_api.hip")])
print("built", os.path.join(E, "libexp.so"))
