// Forward pass of the gfx950 Gaussian rasteriser.
//
// Pipeline (one view):  preprocess -> tile_scan -> scatter -> tile_sort -> blend
//
// The reference (ch3/cuda_rasterizer/rasterizer_impl.cu:184-319) expands every splat into
// (tile|depth) 64-bit keys and runs a device-wide radix sort over them.  Here the binning is
// two-level instead, sized for a 256-CU part with 160 KiB of LDS per CU:
//   1. preprocess also histograms splats per tile (LDS-privatised counters),
//   2. one workgroup turns the histogram into the per-tile [start,end) ranges,
//   3. splats are scattered straight into their tile's segment as (depth bits, id) pairs,
//   4. every tile sorts its own segment inside LDS (global-memory fallback for huge tiles).
// (depth bits, id) is a total order, so the resulting list equals the reference's stable
// radix sort of (tile | depth) keys emitted in id order -- bit for bit -- with 12 B instead of
// 24+ B of traffic per instance and no host synchronisation.
#include "fnx_device.h"
#include "fnx_state.h"

namespace fnx {

// ---------------------------------------------------------------------------------------------
// SH -> RGB (ch3 forward.cu:20-67).  Only reachable with channels == 3.
__device__ inline void sh_to_rgb(int idx, int deg, int M, const float *means, const float *campos, const float *shs,
                                 uint8_t *clamped, float *out) {
    float dx = means[3 * idx] - campos[0], dy = means[3 * idx + 1] - campos[1], dz = means[3 * idx + 2] - campos[2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
    const float *sh = shs + (size_t)idx * M * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
#define SH(k) sh[(k) * 3 + c]
        float result = kSH0 * SH(0);
        if (deg > 0) {
            result = result - kSH1 * y * SH(1) + kSH1 * z * SH(2) - kSH1 * x * SH(3);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z;
                const float xy = x * y, yz = y * z, xz = x * z;
                result = result + kSH2[0] * xy * SH(4) + kSH2[1] * yz * SH(5) + kSH2[2] * (2.0f * zz - xx - yy) * SH(6) +
                         kSH2[3] * xz * SH(7) + kSH2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    result = result + kSH3[0] * y * (3.0f * xx - yy) * SH(9) + kSH3[1] * xy * z * SH(10) +
                             kSH3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                             kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                             kSH3[4] * x * (4.0f * zz - xx - yy) * SH(13) + kSH3[5] * z * (xx - yy) * SH(14) +
                             kSH3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        result += 0.5f;
        clamped[3 * idx + c] = (result < 0);
        out[c] = result > 0.0f ? result : 0.0f;
    }
}

// scale/rotation -> world covariance (ch3 forward.cu:113-145; quaternion used as given, :121).
__device__ inline void cov3d_from_scale_rot(const float *scale, float mod, const float *rot, float *cov3D) {
    M3 S = m3_cols(1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f);
    S.m[0][0] = mod * scale[0];
    S.m[1][1] = mod * scale[1];
    S.m[2][2] = mod * scale[2];
    const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    const M3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                         2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                         2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    const M3 Mm = m3_mul(S, R);
    const M3 Sg = m3_mul(m3_t(Mm), Mm);
    cov3D[0] = Sg.m[0][0];
    cov3D[1] = Sg.m[0][1];
    cov3D[2] = Sg.m[0][2];
    cov3D[3] = Sg.m[1][1];
    cov3D[4] = Sg.m[1][2];
    cov3D[5] = Sg.m[2][2];
}

// EWA projection of the covariance (ch3 forward.cu:70-108).
__device__ inline float3 cov2d_ewa(const float3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                                   const float *cov3D, const float *view) {
    float3 t = xform4x3(mean, view);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    const M3 J = m3_cols(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z), 0.0f, focal_y / t.z,
                         -(focal_y * t.y) / (t.z * t.z), 0.f, 0.f, 0.f);
    const M3 Wm = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    const M3 T = m3_mul(Wm, J);
    const M3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    M3 c2 = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
    c2.m[0][0] += 0.3f;
    c2.m[1][1] += 0.3f;
    return make_float3(c2.m[0][0], c2.m[0][1], c2.m[1][1]);
}

// ---------------------------------------------------------------------------------------------
// K1: per-Gaussian preprocess (ch3 forward.cu:148-244) + per-tile instance histogram.
// One thread per Gaussian, 256-thread workgroups.  `lds_tiles` > 0 => the workgroup keeps a
// private tile histogram in LDS and flushes it with one global atomic per touched tile.
template <int C>
__global__ void __launch_bounds__(256)
preprocess_kernel(int P, int D, int M, const float *__restrict__ means3D, const float *__restrict__ scales,
                  float scale_modifier, const float *__restrict__ rotations, const float *__restrict__ opacities,
                  const float *__restrict__ shs, uint8_t *__restrict__ clamped, const float *__restrict__ cov3D_precomp,
                  const float *__restrict__ colors_precomp, const float *__restrict__ view,
                  const float *__restrict__ proj, const float *__restrict__ campos, int W, int H, float tan_fovx,
                  float tan_fovy, float focal_x, float focal_y, int *__restrict__ radii, float2 *__restrict__ means2D,
                  float *__restrict__ depths, float *__restrict__ cov3Ds, float *__restrict__ rgb,
                  float4 *__restrict__ conic_opacity, int gx, int gy, uint32_t *__restrict__ tiles_touched,
                  uint32_t *__restrict__ tile_count, int lds_tiles, int prefiltered) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (lds_tiles > 0) {
        for (int i = threadIdx.x; i < lds_tiles; i += 256) s_hist[i] = 0;
        __syncthreads();
    }
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    bool live = false;
    if (idx < P) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        const float3 p_orig = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        const float3 p_view = xform4x3(p_orig, view);
        // near cull: only view-space z <= 0.2 (ch3 auxiliary.h:138)
        if (!(p_view.z <= 0.2f)) {
            const float4 p_hom = xform4x4(p_orig, proj);
            const float p_w = 1.0f / (p_hom.w + 0.0000001f);
            const float3 p_proj = make_float3(p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w);
            const float *cov3D;
            if (cov3D_precomp != nullptr) {
                cov3D = cov3D_precomp + (size_t)idx * 6;
            } else {
                cov3d_from_scale_rot(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx,
                                     cov3Ds + (size_t)idx * 6);
                cov3D = cov3Ds + (size_t)idx * 6;
            }
            const float3 cov = cov2d_ewa(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view);
            const float det = (cov.x * cov.z - cov.y * cov.y);
            if (det != 0.0f) {
                const float det_inv = 1.f / det;
                const float3 conic = make_float3(cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv);
                const float mid = 0.5f * (cov.x + cov.z);
                const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
                const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
                const float px = ndc2pix(p_proj.x, W), py = ndc2pix(p_proj.y, H);
                tile_rect(px, py, (int)my_radius, gx, gy, x0, y0, x1, y1);
                if ((uint32_t)(x1 - x0) * (uint32_t)(y1 - y0) != 0) {
                    if (colors_precomp == nullptr) {
                        float res[3];
                        sh_to_rgb(idx, D, M, means3D, campos, shs, clamped, res);
                        rgb[(size_t)idx * C + 0] = res[0];
                        if (C > 1) rgb[(size_t)idx * C + 1] = res[1];
                        if (C > 2) rgb[(size_t)idx * C + 2] = res[2];
                    }
                    depths[idx] = p_view.z;
                    radii[idx] = (int)my_radius;
                    means2D[idx] = make_float2(px, py);
                    conic_opacity[idx] = make_float4(conic.x, conic.y, conic.z, opacities[idx]);
                    tiles_touched[idx] = (uint32_t)(y1 - y0) * (uint32_t)(x1 - x0);
                    live = true;
                }
            }
        } else if (prefiltered) {
            __builtin_trap();  // ch3 auxiliary.h:140-143
        }
    }
    if (live) {
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                const int t = y * gx + x;
                if (lds_tiles > 0)
                    atomicAdd(&s_hist[t], 1u);
                else
                    atomicAdd(&tile_count[t], 1u);
            }
    }
    if (lds_tiles > 0) {
        __syncthreads();
        for (int i = threadIdx.x; i < lds_tiles; i += 256) {
            const uint32_t c = s_hist[i];
            if (c) atomicAdd(&tile_count[i], c);
        }
    }
}

// K2: per-tile counts -> [start,end) ranges (empty tiles keep (0,0) like the reference's memset,
// rasterizer_impl.cu:292), total instance count -> header, cursors zeroed.  One 1024-thread block.
__global__ void __launch_bounds__(1024)
tile_scan_kernel(int T, const uint32_t *__restrict__ tile_count, uint32_t *__restrict__ ranges,
                 uint32_t *__restrict__ tile_cursor, uint32_t *__restrict__ header) {
    __shared__ uint32_t s_part[1024];
    const int tid = threadIdx.x;
    const int per = (T + 1023) / 1024;
    const int b = tid * per, e = min(T, b + per);
    uint32_t sum = 0;
    for (int i = b; i < e; i++) sum += tile_count[i];
    s_part[tid] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = (tid >= off) ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;  // exclusive prefix of this thread's chunk
    for (int i = b; i < e; i++) {
        const uint32_t c = tile_count[i];
        ranges[2 * i] = c ? run : 0u;
        ranges[2 * i + 1] = c ? run + c : 0u;
        tile_cursor[i] = 0u;
        run += c;
    }
    if (tid == 1023) {
        header[HDR_NUM_RENDERED] = s_part[1023];
        header[HDR_STATUS] = 0u;
    }
}

// K3: scatter every (splat, tile) instance into its tile's segment (ch3 rasterizer_impl.cu:67-104
// emits the same instances; slot order inside a tile is fixed by the sort that follows).
__global__ void __launch_bounds__(256)
scatter_kernel(int P, const float2 *__restrict__ means2D, const float *__restrict__ depths,
               const int *__restrict__ radii, int gx, int gy, const uint32_t *__restrict__ ranges,
               uint32_t *__restrict__ tile_cursor, uint64_t *__restrict__ pairs, uint32_t *__restrict__ header,
               uint32_t capacity) {
    if (header[HDR_NUM_RENDERED] > capacity) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            header[HDR_STATUS] = FNX_ERR_CAPACITY;
            header[HDR_CAPACITY] = capacity;
        }
        return;
    }
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const int rad = radii[idx];
    if (rad > 0) {
        const float2 p = means2D[idx];
        int x0, y0, x1, y1;
        tile_rect(p.x, p.y, rad, gx, gy, x0, y0, x1, y1);
        const uint64_t hi = (uint64_t)__float_as_uint(depths[idx]) << 32;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                const int t = y * gx + x;
                const uint32_t slot = ranges[2 * t] + atomicAdd(&tile_cursor[t], 1u);
                pairs[slot] = hi | (uint32_t)idx;
            }
    }
}

// K4: one workgroup sorts one tile's (depth bits, id) pairs ascending.  All compare-exchanges of
// this bitonic network ("flip" then "disperse" steps) order ascending, so slots past n behave as
// +inf padding without being materialised.  Segments up to `lds_cap` pairs are sorted in LDS;
// larger ones in place in global memory by the same workgroup (rare, slow, correct).
template <typename KeyPtr>
__device__ inline void bitonic_all_ascending(KeyPtr key, uint32_t n, uint32_t npow2, int tid, int nthreads) {
    for (uint32_t k = 2; k <= npow2; k <<= 1) {
        const uint32_t half = k >> 1;
        for (uint32_t p = tid; p < (npow2 >> 1); p += nthreads) {  // flip
            const uint32_t blk = p / half, off = p - blk * half;
            const uint32_t i = blk * k + off, l = blk * k + (k - 1 - off);
            if (l < n) {
                const uint64_t a = key[i], b = key[l];
                if (a > b) {
                    key[i] = b;
                    key[l] = a;
                }
            }
        }
        __syncthreads();
        for (uint32_t j = half >> 1; j >= 1; j >>= 1) {  // disperse
            for (uint32_t p = tid; p < (npow2 >> 1); p += nthreads) {
                const uint32_t i = (p / j) * (j << 1) + (p % j), l = i + j;
                if (l < n) {
                    const uint64_t a = key[i], b = key[l];
                    if (a > b) {
                        key[i] = b;
                        key[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(256)
tile_sort_kernel(const uint32_t *__restrict__ ranges, uint64_t *__restrict__ pairs, uint32_t *__restrict__ point_list,
                 const uint32_t *__restrict__ header, uint32_t capacity, uint32_t lds_cap) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_key[];
    if (header[HDR_NUM_RENDERED] > capacity) return;
    const uint32_t start = ranges[2 * blockIdx.x], end = ranges[2 * blockIdx.x + 1];
    const uint32_t n = end - start;
    if (n == 0) return;
    const int tid = threadIdx.x;
    uint32_t npow2 = 1;
    while (npow2 < n) npow2 <<= 1;
    if (n <= lds_cap) {
        for (uint32_t i = tid; i < n; i += 256) s_key[i] = pairs[start + i];
        __syncthreads();
        if (n > 1) bitonic_all_ascending(s_key, n, npow2, tid, 256);
        for (uint32_t i = tid; i < n; i += 256) point_list[start + i] = (uint32_t)s_key[i];
    } else {
        uint64_t *g = pairs + start;
        __syncthreads();
        bitonic_all_ascending((volatile uint64_t *)g, n, npow2, tid, 256);
        for (uint32_t i = tid; i < n; i += 256) point_list[start + i] = (uint32_t)g[i];
    }
}

// XCD-aware tile order: workgroup b lands on XCD b % 8 (observed dispatch order, speed only), so
// hand each XCD a contiguous band of tiles -> neighbouring tiles share splat records in one L2.
__device__ __forceinline__ int xcd_tile(int bid, int T) {
    const int q = T >> 3, r = T & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// K5: front-to-back alpha blending, one 256-thread workgroup per 16x16 tile
// (ch3 forward.cu:249-373).  Splat records of a batch are staged once in LDS, colours included.
template <int C>
__global__ void __launch_bounds__(256)
blend_forward_kernel(int T, int gx, const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W,
                     int H, const float2 *__restrict__ means2D, const float *__restrict__ features,
                     const float4 *__restrict__ conic_opacity, const float *__restrict__ depths,
                     const float *__restrict__ bg, float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                     float *__restrict__ out_color, float *__restrict__ out_depth, const uint32_t *__restrict__ header,
                     uint32_t capacity) {
    __shared__ float2 s_xy[256];
    __shared__ float4 s_co[256];
    __shared__ float s_depth[256];
    __shared__ float s_col[C][256];
    if (header[HDR_NUM_RENDERED] > capacity) return;
    const int tile = xcd_tile(blockIdx.x, T);
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x;
    const int px = tx * FNX_TILE_X + (tid & 15), py = ty * FNX_TILE_Y + (tid >> 4);
    const bool inside = px < W && py < H;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    bool done = !inside;
    float Tr = 1.0f;
    uint32_t contributor = 0, last_contributor = 0;
    float acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
    float Dm = 15.0f;  // median depth default (ch3 forward.cu:295)
    for (uint32_t base = r0; base < r1; base += 256) {
        if (__syncthreads_count(done) == 256) break;
        const uint32_t cnt = min(256u, r1 - base);
        if ((uint32_t)tid < cnt) {
            const uint32_t id = point_list[base + tid];
            s_xy[tid] = means2D[id];
            s_co[tid] = conic_opacity[id];
            s_depth[tid] = depths[id];
#pragma unroll
            for (int ch = 0; ch < C; ch++) s_col[ch][tid] = features[(size_t)id * C + ch];
        }
        __syncthreads();
        for (uint32_t j = 0; !done && j < cnt; j++) {
            contributor++;
            const float2 xy = s_xy[j];
            const float dx = xy.x - pxf, dy = xy.y - pyf;
            const float4 co = s_co[j];
            const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
            if (power > 0.0f) continue;
            const float alpha = fminf(0.99f, co.w * exp_fixed(power));
            if (alpha < 1.0f / 255.0f) continue;
            const float test_T = Tr * (1 - alpha);
            if (test_T < 0.0001f) {
                done = true;
                continue;
            }
#pragma unroll
            for (int ch = 0; ch < C; ch++) acc[ch] += s_col[ch][j] * alpha * Tr;
            if (Tr > 0.5f && test_T < 0.5f) Dm = s_depth[j];
            Tr = test_T;
            last_contributor = contributor;
        }
    }
    if (inside) {
        final_T[pix_id] = Tr;
        n_contrib[pix_id] = last_contributor;
#pragma unroll
        for (int ch = 0; ch < C; ch++) out_color[(size_t)ch * H * W + pix_id] = acc[ch] + Tr * bg[ch];
        out_depth[pix_id] = Dm;
    }
}

// rasterizer_impl.cu:52-63
__global__ void __launch_bounds__(256)
mark_visible_kernel(int P, const float *__restrict__ means3D, const float *__restrict__ view, uint8_t *__restrict__ present) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    const float3 pv = xform4x3(p, view);
    present[idx] = !(pv.z <= 0.2f);
}

}  // namespace fnx

// ---------------------------------------------------------------------------------------------
// Host-side launchers (internal; the C ABI lives in raster_api.hip).
namespace fnx {

static constexpr int kLdsTilesMax = 8192;    // 32 KiB tile histogram per preprocess workgroup
static constexpr uint32_t kSortLdsCap = 4096;  // 32 KiB of (depth,id) pairs per sort workgroup

template <int C>
static void launch_preprocess_c(hipStream_t s, int P, int D, int M, const float *means3D, const float *scales,
                                float scale_modifier, const float *rotations, const float *opacities, const float *shs,
                                uint8_t *clamped, const float *cov3D_precomp, const float *colors_precomp,
                                const float *view, const float *proj, const float *campos, int W, int H, float tan_fovx,
                                float tan_fovy, int *radii, float2 *means2D, float *depths, float *cov3Ds, float *rgb,
                                float4 *conic_opacity, uint32_t *tiles_touched, uint32_t *tile_count, int prefiltered) {
    const int gx = tiles_x(W), gy = tiles_y(H), T = gx * gy;
    const float focal_y = H / (2.0f * tan_fovy);  // rasterizer_impl.cu:207-208
    const float focal_x = W / (2.0f * tan_fovx);
    const int lds_tiles = (T <= kLdsTilesMax) ? T : 0;
    hipLaunchKernelGGL((preprocess_kernel<C>), dim3((P + 255) / 256), dim3(256), (size_t)lds_tiles * 4, s, P, D, M,
                       means3D, scales, scale_modifier, rotations, opacities, shs, clamped, cov3D_precomp,
                       colors_precomp, view, proj, campos, W, H, tan_fovx, tan_fovy, focal_x, focal_y, radii, means2D,
                       depths, cov3Ds, rgb, conic_opacity, gx, gy, tiles_touched, tile_count, lds_tiles, prefiltered);
}

void launch_preprocess(int C, hipStream_t s, int P, int D, int M, const float *means3D, const float *scales,
                       float scale_modifier, const float *rotations, const float *opacities, const float *shs,
                       uint8_t *clamped, const float *cov3D_precomp, const float *colors_precomp, const float *view,
                       const float *proj, const float *campos, int W, int H, float tan_fovx, float tan_fovy, int *radii,
                       float2 *means2D, float *depths, float *cov3Ds, float *rgb, float4 *conic_opacity,
                       uint32_t *tiles_touched, uint32_t *tile_count, int prefiltered) {
    if (C == 3)
        launch_preprocess_c<3>(s, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, clamped,
                               cov3D_precomp, colors_precomp, view, proj, campos, W, H, tan_fovx, tan_fovy, radii,
                               means2D, depths, cov3Ds, rgb, conic_opacity, tiles_touched, tile_count, prefiltered);
    else
        launch_preprocess_c<1>(s, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, clamped,
                               cov3D_precomp, colors_precomp, view, proj, campos, W, H, tan_fovx, tan_fovy, radii,
                               means2D, depths, cov3Ds, rgb, conic_opacity, tiles_touched, tile_count, prefiltered);
}

void launch_tile_scan(hipStream_t s, int T, const uint32_t *tile_count, uint32_t *ranges, uint32_t *tile_cursor,
                      uint32_t *header) {
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, T, tile_count, ranges, tile_cursor, header);
}

void launch_scatter(hipStream_t s, int P, const float2 *means2D, const float *depths, const int *radii, int W, int H,
                    const uint32_t *ranges, uint32_t *tile_cursor, uint64_t *pairs, uint32_t *header,
                    uint32_t capacity) {
    hipLaunchKernelGGL(scatter_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means2D, depths, radii, tiles_x(W),
                       tiles_y(H), ranges, tile_cursor, pairs, header, capacity);
}

void launch_tile_sort(hipStream_t s, int T, const uint32_t *ranges, uint64_t *pairs, uint32_t *point_list,
                      const uint32_t *header, uint32_t capacity) {
    hipLaunchKernelGGL(tile_sort_kernel, dim3(T), dim3(256), (size_t)kSortLdsCap * 8, s, ranges, pairs, point_list,
                       header, capacity, kSortLdsCap);
}

void launch_blend_forward(int C, hipStream_t s, int W, int H, const uint32_t *ranges, const uint32_t *point_list,
                          const float2 *means2D, const float *features, const float4 *conic_opacity,
                          const float *depths, const float *bg, float *final_T, uint32_t *n_contrib, float *out_color,
                          float *out_depth, const uint32_t *header, uint32_t capacity) {
    const int gx = tiles_x(W), T = gx * tiles_y(H);
    if (C == 3)
        hipLaunchKernelGGL((blend_forward_kernel<3>), dim3(T), dim3(256), 0, s, T, gx, ranges, point_list, W, H,
                           means2D, features, conic_opacity, depths, bg, final_T, n_contrib, out_color, out_depth,
                           header, capacity);
    else
        hipLaunchKernelGGL((blend_forward_kernel<1>), dim3(T), dim3(256), 0, s, T, gx, ranges, point_list, W, H,
                           means2D, features, conic_opacity, depths, bg, final_T, n_contrib, out_color, out_depth,
                           header, capacity);
}

void launch_mark_visible(hipStream_t s, int P, const float *means3D, const float *view, uint8_t *present) {
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
}

}  // namespace fnx
