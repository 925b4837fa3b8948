// Forward pass of the gfx950 Gaussian rasteriser.
//
// This file: per-splat preprocess (+ per-(splat block, tile) instance counts and depth-sort keys)
// and the front-to-back blend.  The binning between them lives in raster_binning.hip.
#include <mutex>
#include <cstdlib>
#include "fnx_device.h"
#include "fnx_state.h"

#include "lab/fnx_lab.h"  // experiment switches (all off in the production build)
#ifdef FNX_EXP_STATS  // developer statistics of the blend forward's inner loop (tools/deep_probe.py)
__device__ unsigned long long g_fwd_stats[8];
extern "C" int fnx_debug_fwd_stats(unsigned long long *host, int reset) {
    if (reset) {
        unsigned long long z[8] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_stats), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fwd_stats), sizeof(g_fwd_stats));
}
#endif
#ifdef FNX_EXP_CLOCK  // developer timing: per-phase cycles of every wave's thread 0 of workgroup (0, 0) of the blend forward
__device__ unsigned long long g_fwd_clock[64];
extern "C" int fnx_debug_fwd_clock(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fwd_clock), sizeof(g_fwd_clock));
}
__device__ unsigned long long g_fwd_wg[4 * 16384];
__device__ unsigned long long g_fwd_wg2[16384];  // segmented tiles: when the last segment to arrive started putting them together
extern "C" int fnx_debug_fwd_wg_reset() {
    void *a = nullptr, *b = nullptr;
    (void)hipGetSymbolAddress(&a, HIP_SYMBOL(g_fwd_wg));
    (void)hipGetSymbolAddress(&b, HIP_SYMBOL(g_fwd_wg2));
    (void)hipMemset(a, 0, sizeof(g_fwd_wg));
    return (int)hipMemset(b, 0, sizeof(g_fwd_wg2));
}
extern "C" int fnx_debug_fwd_wg2(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fwd_wg2), (size_t)n * 8);
}  // per workgroup (view * T + rank): wall start, wall end, depth << 32 | list length, staged entries
extern "C" int fnx_debug_fwd_wg(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fwd_wg), (size_t)n * 8);
}
#define FNX_CLK(i) { const unsigned long long tn = clock64(); if (wg_rank == 0 && wg_view == 0 && lane == 0) g_fwd_clock[16 * w + (i)] += tn - t_last; t_last = tn; }
#else
#define FNX_CLK(i)
#endif
#ifndef FNX_LDS_BARRIER
#define FNX_LDS_BARRIER 1  // 0: plain __syncthreads() inside the batch loops (timing experiments)
#endif
#if FNX_LDS_BARRIER
#define FNX_LOOP_BARRIER() fnx::lds_barrier()
#else
#define FNX_LOOP_BARRIER() __syncthreads()
#endif
// lab probe (results WRONG): the staging / list barriers of the blend forward's batch loop become wave-local waits --
// an upper bound for what a barrier-free batch structure could buy (fnx_lab.h, FNX_EXP_NOBAR)
#if FNX_EXP_NOBAR & 1
#define FNX_LOOP_BARRIER_BC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define FNX_LOOP_BARRIER_BC() FNX_LOOP_BARRIER()
#endif
#ifndef FNX_DEEP_PRIO
#define FNX_DEEP_PRIO 3  // wave priority (0..3) of the tiles that went deep in the previous forward
#endif

namespace fnx {

// ---------------------------------------------------------------------------------------------
// SH -> RGB (ch3 forward.cu:20-67).  Only reachable with channels == 3.
__device__ __forceinline__ void sh_eval(int deg, float dx, float dy, float dz, const float *sh, uint8_t *clamped3, float *out) {
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
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
        clamped3[c] = (result < 0);
        out[c] = result > 0.0f ? result : 0.0f;
    }
}
__device__ inline void sh_to_rgb(int idx, int deg, int M, const float *means, const float *campos, const float *shs,
                                 uint8_t *clamped, float *out) {
    sh_eval(deg, means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2],
            shs + (size_t)idx * M * 3, clamped + 3 * (size_t)idx, out);
}

// View batches with SH colours: the colour of every (Gaussian, view) in one pass over the GAUSSIANS -- the 4 M 3 bytes of
// coefficients (192 for degree 3) are read once and evaluated for all V view directions, instead of once per view by
// the preprocess (V x 192 of the ~1 100 bytes per Gaussian a 5-view launch moved: profiles/r03_sh_*).  Same expression,
// same order: bit-identical colours and clamp flags; they are written for every Gaussian, visible in a view or not.
__global__ void __launch_bounds__(256)
sh_colors_views_kernel(int P, int D, int M, int V, const float *__restrict__ means3D, const float *__restrict__ campos,
                       const float *__restrict__ shs, uint8_t *__restrict__ clamped, float *__restrict__ rgb,
                       size_t geom_stride) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    float sh[48];
    const float *src = shs + (size_t)idx * M * 3;
#pragma unroll
    for (int k = 0; k < 48; k++) sh[k] = k < 3 * M ? src[k] : 0.f;
    const float mx = means3D[3 * idx], my = means3D[3 * idx + 1], mz = means3D[3 * idx + 2];
    for (int v = 0; v < V; v++) {
        float col[3];
        uint8_t cl[3];
        sh_eval(D, mx - campos[3 * v], my - campos[3 * v + 1], mz - campos[3 * v + 2], sh, cl, col);
        float *o = view_at(rgb, geom_stride, v) + 3 * (size_t)idx;
        uint8_t *c = view_at(clamped, geom_stride, v) + 3 * (size_t)idx;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            o[k] = col[k];
            c[k] = cl[k];
        }
    }
}

#include "sh_mfma.h"  // the same colours through v_mfma_f32_4x4x1_16B_f32: the fast arithmetic's kernel

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
// K1: per-Gaussian preprocess (ch3 forward.cu:148-244), one thread per splat.  Besides the
// reference's per-splat state it writes the packed 64-byte record the blend kernels read and
//   sort_key[idx]: depth bits of visible splats; bit 31 | depth bits for culled ones (they keep their place in depth so
//                  that a splat entering / leaving a view does not jump through the sorted order, raster_binning.hip),
//   key_min_blk[block]: the smallest key of the workgroup's splats (the sort works relative to the minimum).
template <int C>
__global__ void __launch_bounds__(256)
preprocess_kernel(int P, int D, int M, const float *__restrict__ means3D, const float *__restrict__ scales,
                  float scale_modifier, const float *__restrict__ rotations, const float *__restrict__ opacities,
                  const float *__restrict__ shs, uint8_t *__restrict__ clamped, const float *__restrict__ cov3D_precomp,
                  const float *__restrict__ colors_precomp, const float *__restrict__ view,
                  const float *__restrict__ proj, const float *__restrict__ campos, int W, int H,
                  int *__restrict__ radii, float2 *__restrict__ means2D,
                  float *__restrict__ depths, float *__restrict__ cov3Ds, float *__restrict__ rgb,
                  float4 *__restrict__ conic_opacity, int gx, int gy, uint32_t *__restrict__ tiles_touched,
                  uint32_t *__restrict__ sort_key, uint32_t *__restrict__ key_min_blk, uint2 *__restrict__ rect,
                  float4 *__restrict__ blend_rec, int prefiltered, const StaticRef st, const ViewBatch vb, int lean,
                  float *__restrict__ zero3, const CohRef coh, int sh_pre) {
    // lean (fnx_set_lean_geometry): the copies of the reference's GeometryState that nothing in this library reads back
    // (means2D, depths, conic_opacity, tiles_touched: the blend records carry the same numbers) are not written, and the
    // world covariance -- the same for every view -- is written by view 0 only (the backward reads it at stride 0)
    __shared__ uint32_t s_min[4];
    const int vw = blockIdx.y;  // view of the batch: camera, radii and the geometry blob are per view
    radii += (size_t)vw * vb.radii_stride;
    {
        // static-split mode: the workgroups behind the per-call splats copy the static splats' radii out of the
        // view's static blob (the caller's radii array covers all splats)
        const int nb_dyn = (P + 255) / 256;
        if ((int)blockIdx.x >= nb_dyn) {
            const int k = ((int)blockIdx.x - nb_dyn) * 256 + (int)threadIdx.x;
            if (k < st.P) {
                radii[(size_t)st.id0 + k] = reinterpret_cast<const int *>(st.base + st.stride * vw + st.radii)[k];
                if (zero3 && vw == 0) {
                    float *z = zero3 + 3 * ((size_t)st.id0 + k);
                    z[0] = z[1] = z[2] = 0.f;
                }
            }
            return;
        }
    }
    view += 16 * vw;
    proj += 16 * vw;
    if (campos) campos += 3 * vw;
    clamped = view_at(clamped, vb.geom, vw);
    means2D = view_at(means2D, vb.geom, vw);
    depths = view_at(depths, vb.geom, vw);
    cov3Ds = view_at(cov3Ds, vb.geom, vw);
    rgb = view_at(rgb, vb.geom, vw);
    conic_opacity = view_at(conic_opacity, vb.geom, vw);
    tiles_touched = view_at(tiles_touched, vb.geom, vw);
    sort_key = view_at(sort_key, vb.geom, vw);
    key_min_blk = view_at(key_min_blk, vb.geom, vw);
    rect = view_at(rect, vb.geom, vw);
    blend_rec = view_at(blend_rec, vb.geom, vw);
    const float tan_fovx = vb.tan_fovx[vw], tan_fovy = vb.tan_fovy[vw];
    const float focal_x = vb.focal_x[vw], focal_y = vb.focal_y[vw];
    uint32_t key = 0xFFFFFFFFu;  // bit 31 = culled (threads past the end included)
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < P) {
        int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        bool visible = false;
        if (zero3 && vw == 0) {  // the caller's [P_all, 3] accumulator of the positions-only backward, zeroed on the way
            zero3[3 * (size_t)idx] = 0.f;
            zero3[3 * (size_t)idx + 1] = 0.f;
            zero3[3 * (size_t)idx + 2] = 0.f;
        }
        radii[idx] = 0;
        if (!lean) tiles_touched[idx] = 0;
        const float3 p_orig = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        if (lean && vw == 0) {  // the one covariance array of the batch: for every splat, whichever views see it
            float c0[6];
            cov3d_from_scale_rot(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, c0);
#pragma unroll
            for (int k = 0; k < 6; k++) cov3Ds[(size_t)idx * 6 + k] = c0[k];
        }
        const float3 p_view = xform4x3(p_orig, view);
        key = 0x80000000u | (p_view.z > 0.0f ? (__float_as_uint(p_view.z) & 0x7FFFFFFFu) : 0u);  // culled until proven visible
        // near cull: only view-space z <= 0.2 (ch3 auxiliary.h:138)
        if (!(p_view.z <= 0.2f)) {
            const float4 p_hom = xform4x4(p_orig, proj);
            const float p_w = 1.0f / (p_hom.w + 0.0000001f);
            const float3 p_proj = make_float3(p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w);
            // (the six values in registers either way: a pointer that may refer to a local array puts the array into
            // scratch memory, and a kernel with a scratch segment costs the queue ~5 us in front of its launch)
            float c3[6];
            if (cov3D_precomp != nullptr) {
#pragma unroll
                for (int k = 0; k < 6; k++) c3[k] = cov3D_precomp[(size_t)idx * 6 + k];
            } else {
                cov3d_from_scale_rot(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, c3);
                if (!lean) {
#pragma unroll
                    for (int k = 0; k < 6; k++) cov3Ds[(size_t)idx * 6 + k] = c3[k];
                }
            }
            const float3 cov = cov2d_ewa(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, c3, view);
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
                    float col[3] = {0.f, 0.f, 0.f};
                    if (colors_precomp == nullptr && sh_pre) {  // sh_colors_views_kernel evaluated it (and the clamp flags)
#pragma unroll
                        for (int ch = 0; ch < C; ch++) col[ch] = rgb[(size_t)idx * C + ch];
                    } else if (colors_precomp == nullptr) {
                        sh_to_rgb(idx, D, M, means3D, campos, shs, clamped, col);
                        rgb[(size_t)idx * C + 0] = col[0];
                        if (C > 1) rgb[(size_t)idx * C + 1] = col[1];
                        if (C > 2) rgb[(size_t)idx * C + 2] = col[2];
                    } else {
#pragma unroll
                        for (int ch = 0; ch < C; ch++) col[ch] = colors_precomp[(size_t)idx * C + ch];
                    }
                    const float opac = opacities[idx];
                    radii[idx] = (int)my_radius;
                    if (!lean) {
                        depths[idx] = p_view.z;
                        means2D[idx] = make_float2(px, py);
                        conic_opacity[idx] = make_float4(conic.x, conic.y, conic.z, opac);
                    }
                    // packed record for the blend kernels (one 64-byte line per splat)
                    float thr, ex, ey;
                    splat_footprint(conic.x, conic.y, conic.z, opac, thr, ex, ey);
                    float4 *rec = blend_rec + 4 * (size_t)idx;
                    rec[0] = make_float4(px, py, conic.x, conic.y);
                    rec[1] = make_float4(conic.z, opac, thr, p_view.z);
                    rec[2] = make_float4(ex, ey, col[0], col[1]);
                    // (.yzw: the splat's world position -- the positions-only backward's flush reads it with the record
                    //  instead of gathering it from means3D, raster_backward_lanes.h)
                    rec[3] = make_float4(col[2], p_orig.x, p_orig.y, p_orig.z);
                    if (!lean) tiles_touched[idx] = (uint32_t)(y1 - y0) * (uint32_t)(x1 - x0);
                    // a NaN depth passes the near cull as in the reference; its key keeps bit 31 out of the way of the flag
                    key = __float_as_uint(p_view.z) & 0x7FFFFFFFu;
                    visible = true;
                }
            }
        } else if (prefiltered) {
            __builtin_trap();  // ch3 auxiliary.h:140-143
        }
        sort_key[idx] = key;
        if (coh.state) {
            // temporal-coherence sort (raster_binning.hip): the splat's record goes to the slot of its PREVIOUS depth rank,
            // stamped with this call's number -- a record the repair kernel finds without that stamp was not written now
            char *stv = coh.state + coh.stride * (size_t)vw;
            const uint32_t slot = reinterpret_cast<const uint32_t *>(stv + coh.inv)[idx];
            if (slot < (uint32_t)P) {
                uint32_t *hdr = reinterpret_cast<uint32_t *>(stv + coh.hdr);
                const uint32_t kbits = key & 0x7FFFFFFFu, epoch = coh_stamp(hdr);
                uint4 rec = make_uint4(kbits, (uint32_t)idx,
                                       visible ? ((uint32_t)x0 | ((uint32_t)x1 << 8) | ((uint32_t)y0 << 16) | ((uint32_t)y1 << 24)) : 0u,
                                       epoch);
                // has the splat left the neighbourhood of its previous rank (by the previous order's sampled keys)?  Then
                // its record travels in the outlier list and the slot gets a hole; the repair kernel merges the list in.
                if (hdr[COH_SAMPLES_OK] == 1u) {
                    const uint32_t *smp = reinterpret_cast<const uint32_t *>(stv + coh.samples);
                    const uint32_t jb = slot / (uint32_t)kCohSampleStep, ns = ((uint32_t)P + kCohSampleStep - 1) / kCohSampleStep;
                    const uint32_t lo = jb >= (uint32_t)kCohSampleReach ? smp[jb - kCohSampleReach] : 0u;
                    const uint32_t hi = jb + kCohSampleReach + 1 < ns ? smp[jb + kCohSampleReach + 1] : 0x7FFFFFFFu;
                    if (kbits < lo || kbits > hi) {
                        // (a full list is no failure: the splat stays in its slot like any other, and the repair window
                        // either reaches its new place or the call takes the full sort, as it would without the list)
                        const uint32_t n = atomicAdd(&hdr[COH_NOUT], 1u);
                        if (n < (uint32_t)kCohOutlierCap) {
                            reinterpret_cast<uint4 *>(stv + coh.olist)[n] = rec;
                            atomicAdd(&reinterpret_cast<uint32_t *>(stv + coh.holes)[slot >> 10], 1u);
                            rec = make_uint4(0u, kCohHoleId, 0u, epoch);
                        }
                    }
                }
                view_at(coh.krec, vb.geom, vw)[slot] = rec;
            }
        }
        // tile rectangle for the binning kernels ((0, 0) = no instances)
        rect[idx] = visible ? make_uint2((uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16))
                            : make_uint2(0u, 0u);
    }
    // smallest key of the workgroup (visible keys have bit 31 clear: they win against every culled one)
    uint32_t m = key;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) key_min_blk[blockIdx.x] = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
}

// ---------------------------------------------------------------------------------------------
// Inner loops of the blend forward's FAST arithmetic (fnx_set_blend_math(1)), shared by the per-tile kernel and the
// super-batch kernel for deep tiles.  `mylist`: the lane's block list (LDS byte offsets of staged entries, padded with
// the NULL record), n_w: steps of the wave (its longest list).  Staged records: s_ra = (x, y, A, B), s_rb = (C, log2 o,
// colour 0, colour 1 | depth), s_rc = (colour 2, depth) with (A, B, C) = -log2(e) (a / 2, b, c / 2) of the conic.
template <int C>
__device__ __forceinline__ float fast_alpha(const float4 ra, const float4 rb, float pxf, float pyf) {
    const float dx = ra.x - pxf, dy = ra.y - pyf;
    const float u = __builtin_fmaf(ra.z, dx, ra.w * dy);
    const float q = __builtin_fmaf(u, dx, (rb.x * dy) * dy);  // log2(e) * power
    const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(q + rb.y));
    return (!(q > 0.0f) && !(alpha < 1.0f / 255.0f)) ? alpha : 0.0f;  // forward.cu:326-335
}

// Full walk: `alive` is the WORKING transmittance (T while the pixel blends, 0 once it has stopped or if it lies
// outside the image), Tr the pixel's transmittance, acc its colour, hit_off the LDS offset of the last entry taken.
// DUAL (fnx_raster_dual_t): a second, single-channel image over the DYNAMIC entries only, blended in the same walk.  A list
// entry of a static splat carries bit 15 (kListStatic) on top of its LDS byte offset; for the second image such an entry
// has alpha 0, everything else -- the entry's alpha, the order, the stop rule -- is the first image's arithmetic on the
// second image's own transmittance, so its pixels equal a 1-channel render of the dynamic splats alone.
constexpr uint32_t kListStatic = 0x8000u, kListOffMask = 0x1FFFu;
struct DualPixel {
    float acc, Tr, alive, Dm;
    uint32_t hit_off;
};
// REC (segmented tiles): vstop keeps the test value of the entry that stopped the pixel (T (1 - alpha) < 1e-4; 0: none yet).
template <int C, bool DUAL, bool REC, bool HALF>
__device__ __forceinline__ void fast_walk_impl(const uint16_t *mylist, uint32_t n_w, const float4 *s_ra, const float4 *s_rb,
                                          const float4 *s_rc, float pxf, float pyf, float (&acc)[C], float &Tr,
                                          float &alive, float &Dm, uint32_t &hit_off, DualPixel *du, float *vstop) {
    constexpr int kGroup = 4;
    // entries of this walk after which the pixel's T is still >= 1/2: T never rises, so they are a prefix of the
    // walk, and if T crosses 1/2 here the entry that took it across is mylist[n_half] (forward.cu:351-354)
    uint32_t n_half = 0, n_half1 = 0;
    const float T_in = Tr, T1_in = DUAL ? du->Tr : 0.f;
#if FNX_WALK_LIST_AHEAD
    // the list words of the NEXT group are requested in front of this group's records: a deep tile's wave runs all but alone on
    // its SIMD and waits out every LDS round trip itself -- list word -> records are two dependent ones per group
    uint32_t jn[kGroup / 2];
#pragma unroll
    for (int k = 0; k < kGroup / 2; k++) jn[k] = reinterpret_cast<const uint32_t *>(mylist)[k];
#endif
    for (uint32_t i0 = 0; i0 < n_w; i0 += kGroup) {
        if (__all(alive == 0.0f && (!DUAL || du->alive == 0.0f))) break;
        uint32_t jw[kGroup / 2];
#if FNX_WALK_LIST_AHEAD
#pragma unroll
        for (int k = 0; k < kGroup / 2; k++) jw[k] = jn[k];
#pragma unroll
        for (int k = 0; k < kGroup / 2; k++) jn[k] = reinterpret_cast<const uint32_t *>(mylist + i0 + kGroup)[k];  // (the lists are padded)
#else
#pragma unroll
        for (int k = 0; k < kGroup / 2; k++) jw[k] = reinterpret_cast<const uint32_t *>(mylist + i0)[k];
#endif
        float a_h[kGroup], col[kGroup][3];
#pragma unroll
        for (int k = 0; k < kGroup; k++) {
            const uint32_t off = (jw[k >> 1] >> (16 * (k & 1))) & (DUAL ? kListOffMask : 0xFFFFu);
            const float4 ra = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_ra) + off);
            const float4 rb = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rb) + off);
            col[k][0] = rb.z;
            col[k][1] = rb.w;
            col[k][2] = C == 3 ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(s_rc) + off) : 0.f;
            a_h[k] = fast_alpha<C>(ra, rb, pxf, pyf);
        }
        if (DUAL) {
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                const uint32_t raw = (jw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, off = raw & kListOffMask;
                const float a1 = (raw & kListStatic) ? 0.0f : a_h[k];
                const float wq = a1 * du->alive;
                const float t = du->alive - wq;
                const bool stop = t < 0.0001f;
                const float wgt = stop ? 0.0f : wq;
                du->acc = __builtin_fmaf(col[k][0], wgt, du->acc);
                du->Tr = stop ? du->Tr : t;
                du->alive = stop ? 0.0f : t;
                if (HALF) n_half1 += (du->Tr >= 0.5f) ? 1u : 0u;
                du->hit_off = (wgt > 0.0f) ? off : du->hit_off;
            }
        }
#pragma unroll
        for (int k = 0; k < kGroup; k++) {
            const uint32_t off = (jw[k >> 1] >> (16 * (k & 1))) & (DUAL ? kListOffMask : 0xFFFFu);
            const float wq = a_h[k] * alive;      // alpha T (0 for a stopped pixel: its working T is 0)
            const float t = alive - wq;           // test_T
            const bool stop = t < 0.0001f;        // also true for every entry behind the one that stopped the pixel
            if (REC) *vstop = fmaxf(*vstop, stop ? t : 0.0f);  // (behind the stop the working T is 0: t = 0)
            const float wgt = stop ? 0.0f : wq;
            acc[0] = __builtin_fmaf(col[k][0], wgt, acc[0]);
            if (C > 1) acc[C > 1 ? 1 : 0] = __builtin_fmaf(col[k][1], wgt, acc[C > 1 ? 1 : 0]);
            if (C > 2) acc[C > 2 ? 2 : 0] = __builtin_fmaf(col[k][2], wgt, acc[C > 2 ? 2 : 0]);
            Tr = stop ? Tr : t;
            alive = stop ? 0.0f : t;
            if (HALF) n_half += (Tr >= 0.5f) ? 1u : 0u;
            hit_off = (wgt > 0.0f) ? off : hit_off;
        }
    }
    if (HALF && T_in >= 0.5f && Tr < 0.5f) {
        const uint32_t off = mylist[n_half] & (DUAL ? kListOffMask : 0xFFFFu);
        Dm = C == 3 ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(s_rc) + off)[1]
                    : reinterpret_cast<const float *>(reinterpret_cast<const char *>(s_rb) + off)[3];
    }
    if (HALF && DUAL && T1_in >= 0.5f && du->Tr < 0.5f) {
        const uint32_t off = mylist[n_half1] & kListOffMask;
        du->Dm = C == 3 ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(s_rc) + off)[1]
                        : reinterpret_cast<const float *>(reinterpret_cast<const char *>(s_rb) + off)[3];
    }
}

// The median-depth bookkeeping (n_half: two instructions per entry and image) only where some pixel of the wave can still
// cross T = 1/2 in this walk: T never rises, and behind the first few batches of a deep tile no pixel's does (round 6).
template <int C, bool DUAL = false, bool REC = false>
__device__ __forceinline__ void fast_walk(const uint16_t *mylist, uint32_t n_w, const float4 *s_ra, const float4 *s_rb,
                                          const float4 *s_rc, float pxf, float pyf, float (&acc)[C], float &Tr,
                                          float &alive, float &Dm, uint32_t &hit_off, DualPixel *du = nullptr,
                                          float *vstop = nullptr) {
    if (__any((Tr >= 0.5f && alive != 0.0f) || (DUAL && du->Tr >= 0.5f && du->alive != 0.0f)))
        fast_walk_impl<C, DUAL, REC, true>(mylist, n_w, s_ra, s_rb, s_rc, pxf, pyf, acc, Tr, alive, Dm, hit_off, du, vstop);
    else
        fast_walk_impl<C, DUAL, REC, false>(mylist, n_w, s_ra, s_rb, s_rc, pxf, pyf, acc, Tr, alive, Dm, hit_off, du, vstop);
}

// Transmittance only: the product of (1 - alpha) over the lane's list, without the stop rule (the caller applies it
// to the running product between sub-batches, blend_forward_deep_kernel).
template <int C>
__device__ __forceinline__ float fast_walk_transmittance(const uint16_t *mylist, uint32_t n_w, const float4 *s_ra,
                                                         const float4 *s_rb, float pxf, float pyf) {
    constexpr int kGroup = 4;
    float P = 1.0f;
    for (uint32_t i0 = 0; i0 < n_w; i0 += kGroup) {
        uint32_t jw[kGroup / 2];
#pragma unroll
        for (int k = 0; k < kGroup / 2; k++) jw[k] = reinterpret_cast<const uint32_t *>(mylist + i0)[k];
        float a_h[kGroup];
#pragma unroll
        for (int k = 0; k < kGroup; k++) {
            const uint32_t off = (jw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
            const float4 ra = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_ra) + off);
            const float4 rb = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rb) + off);
            a_h[k] = fast_alpha<C>(ra, rb, pxf, pyf);
        }
#pragma unroll
        for (int k = 0; k < kGroup; k++) P = __builtin_fmaf(-a_h[k], P, P);
    }
    return P;
}

// XCD-aware tile order: workgroup b lands on XCD b % 8 (observed dispatch order, speed only), so
// hand each XCD a contiguous band of tiles -> neighbouring tiles share splat records in one L2.
__device__ __forceinline__ int xcd_tile(int bid, int T) {
    const int q = T >> 3, r = T & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// Inclusive prefix sum over the 1024 threads of a workgroup: shuffles inside the waves, the 16 wave totals through
// LDS (two barriers; a Hillis-Steele scan through LDS takes twenty).  s_wave: 16 words, reusable after the call returns
// (the trailing barrier).
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t *s_wave) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) before += (k < w) ? s_wave[k] : 0u;
    __syncthreads();
    return inc + before;
}

// K2: per-tile counts -> [start,end) ranges (empty tiles keep (0,0) like the reference's memset,
// rasterizer_impl.cu:292), total instance count -> header.  One 1024-thread block.  It also lays out the
// emission work items of the view (raster_binning.hip): per rank block, 1..kEmitBands bands of tile rows
// depending on the block's instance count, in block order (nearest = heaviest first).
__global__ void __launch_bounds__(1024)
tile_scan_kernel(int T, const uint32_t *__restrict__ tile_count, uint32_t *__restrict__ ranges,
                 uint32_t *__restrict__ dyn_start, uint32_t *__restrict__ header, size_t img_stride, int NB, int gy,
                 const uint32_t *__restrict__ blk_total, uint32_t *__restrict__ emit_ctl,
                 uint32_t *__restrict__ emit_items, size_t geom_stride, uint32_t *__restrict__ depth_hint,
                 uint32_t deep_min, uint32_t prio_min, uint32_t *__restrict__ tile_order, uint8_t *__restrict__ tile_deep,
                 const StaticRef st, const uint32_t *__restrict__ sort_ctl, const SegRef sg) {
    constexpr int kDeepSorted = 1024;
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_deep_n;
    __shared__ uint2 s_deep[kDeepSorted];  // (depth hint, tile) of the deep tiles
    __shared__ uint2 s_sorted[kDeepSorted];    // ... deepest first (segmented launch)
    __shared__ uint32_t s_before[kDeepSorted];  // cut tiles in front of a sorted rank
    __shared__ uint32_t s_tot;
    if (threadIdx.x == 0) s_deep_n = 0;
    tile_order = view_at(tile_order, img_stride, blockIdx.y);
    tile_deep = view_at(tile_deep, img_stride, blockIdx.y);
    if (depth_hint) depth_hint += (size_t)blockIdx.y * T;
    blk_total = view_at(blk_total, geom_stride, blockIdx.y);
    emit_items = view_at(emit_items, geom_stride, blockIdx.y);
    tile_count = view_at(tile_count, img_stride, blockIdx.y);
    ranges = view_at(ranges, img_stride, blockIdx.y);
    dyn_start = view_at(dyn_start, img_stride, blockIdx.y);
    header = view_at(header, img_stride, blockIdx.y);
    // static-split mode: `ranges` address the merged list of a tile (static + per-call instances), dyn_start the
    // tile's slice of the per-call instances alone
    const uint32_t *st_starts =
        st.base ? reinterpret_cast<const uint32_t *>(st.base + st.stride * blockIdx.y + st.starts) : nullptr;
    const int tid = threadIdx.x;
    const int per = (T + 1023) / 1024;
    const int b = tid * per, e = min(T, b + per);
    uint32_t sum = 0;
    for (int i = b; i < e; i++) sum += tile_count[i];
    uint32_t run = block_scan_1024(sum, s_wave) - sum;  // exclusive prefix of this thread's chunk (the barrier inside orders s_deep_n = 0)
    for (int i = b; i < e; i++) {
        const uint32_t cd = tile_count[i];
        const uint32_t s0 = st_starts ? st_starts[i] : 0u, cs = st_starts ? st_starts[i + 1] - s0 : 0u;
        const uint32_t c = cd + cs, at = run + s0;
        ranges[2 * i] = c ? at : 0u;
        ranges[2 * i + 1] = c ? at + c : 0u;
        dyn_start[i] = run;
        run += cd;
        // Tiles that went deep in the previous forward of this view (a list that does not saturate: thousands of
        // contributing entries per pixel) are handed to the first workgroups of the blend launch, at raised wave
        // priority: their sequential walks are the critical path of the launch and must not start last.
        const uint32_t hint = depth_hint ? depth_hint[i] : 0u;
        const bool deep = depth_hint && c && hint >= deep_min;
        // 2: also at raised wave priority (prio_min >= deep_min: the ORDER may reach further down than the priority)
        tile_deep[i] = deep ? (hint >= prio_min ? 2 : 1) : 0;
        if (deep) {
            const uint32_t at = atomicAdd(&s_deep_n, 1u);
            tile_order[at] = (uint32_t)i;
            if (at < kDeepSorted) s_deep[at] = make_uint2(hint, (uint32_t)i);
        }
        if (depth_hint) depth_hint[i] = 0u;  // the blend of this forward records the new depth
    }
    __syncthreads();
    {
        // deepest first (the deep-tile kernel hands them out in this order: longest jobs first); ties by tile index.
        // Beyond kDeepSorted deep tiles the rest keep their arrival order.
        const uint32_t nd = min(s_deep_n, (uint32_t)kDeepSorted);
        if (nd <= 128u) {  // few: every element counts the ones in front of it
            for (uint32_t e = tid; e < nd; e += 1024) {
                const uint2 me = s_deep[e];
                uint32_t rank = 0;
                for (uint32_t o = 0; o < nd; o++) {
                    const uint2 ot = s_deep[o];
                    rank += (ot.x > me.x || (ot.x == me.x && ot.y < me.y)) ? 1u : 0u;
                }
                tile_order[rank] = me.y;
                s_sorted[rank] = me;
            }
        } else {
            // many (the order reaches down to shallow tiles, FNX_DEEP_ORDER_MIN): bitonic sort of kDeepSorted = 1024 keys
            // (hint, ~tile), one per thread, DESCENDING; exchanges at distances below 64 stay inside a wave (shuffles),
            // the ten at distances 64 .. 512 go through LDS.  Empty slots hold key 0 and sink to the end.
            static_assert(kDeepSorted == 1024, "one key per thread");
            unsigned long long key = 0ull;
            if ((uint32_t)tid < nd) {
                const uint2 me = s_deep[tid];
                key = ((unsigned long long)me.x << 32) | (unsigned long long)(0xFFFFFFFFu - me.y);
            }
            unsigned long long *s_key = reinterpret_cast<unsigned long long *>(s_sorted);  // (written for good below)
            for (uint32_t k = 2; k <= 1024u; k <<= 1) {
                for (uint32_t j = k >> 1; j >= 1u; j >>= 1) {
                    unsigned long long other;
                    if (j >= 64u) {
                        __syncthreads();
                        s_key[tid] = key;
                        __syncthreads();
                        other = s_key[(uint32_t)tid ^ j];
                    } else {
                        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)key, (int)j);
                        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), (int)j);
                        other = ((unsigned long long)hi << 32) | lo;
                    }
                    // descending overall: in a block whose bit k is clear the larger key goes to the lower index
                    const bool lower = ((uint32_t)tid & j) == 0u, desc = ((uint32_t)tid & k) == 0u || k == 1024u;
                    const bool take_max = lower == desc;
                    key = take_max ? (key > other ? key : other) : (key < other ? key : other);
                }
            }
            __syncthreads();
            if ((uint32_t)tid < nd) {
                const uint2 me = make_uint2((uint32_t)(key >> 32), 0xFFFFFFFFu - (uint32_t)key);
                tile_order[tid] = me.y;
                s_sorted[tid] = me;
            }
        }
    }
    __syncthreads();
    {
        // the other tiles follow in the XCD-aware order (xcd_tile): position = deep count + rank among the others
        const uint32_t nd = s_deep_n;
        uint32_t mine = 0;
        for (int k = b; k < e; k++) mine += tile_deep[xcd_tile(k, T)] ? 0u : 1u;
        uint32_t at = nd + block_scan_1024(mine, s_wave) - mine;
        for (int k = b; k < e; k++) {
            const int t = xcd_tile(k, T);
            if (!tile_deep[t]) tile_order[at++] = (uint32_t)t;
        }
        __syncthreads();
        // + the bit length of the view's depth-key span in the top byte (the host learns whether three sort passes suffice)
        if (tid == 0) header[HDR_DEEP_COUNT] = min(nd, 0xFFFFFFu) | (view_at(sort_ctl, geom_stride, blockIdx.y)[SORT_CTL_SPAN] << 24);
    }
    if (sg.base) {
        // Work list of the segmented blend launch (fnx_state.h): the deep tiles, deepest first, are cut into segments as far
        // as the previous forward consumed their lists (+ a batch; what lies behind is taken up by the workgroup that puts
        // the segments together, if a pixel still blends there), while the record slots last; the segments lead the list,
        // the other tiles follow in tile order.
        char *sc = sg.base + sg.stride * blockIdx.y;
        uint32_t *sctl = reinterpret_cast<uint32_t *>(sc + sg.L.ctl), *items = reinterpret_cast<uint32_t *>(sc + sg.L.items);
        const uint32_t gen = (sctl[SEG_CTL_PARITY] & 1u) ^ 1u;  // this call's generation of tile_slot / hints
        uint32_t *tile_slot = reinterpret_cast<uint32_t *>(sc + sg.L.tile_slot) + (size_t)gen * T;
        uint32_t *arrive = reinterpret_cast<uint32_t *>(sc + sg.L.arrive);
        __syncthreads();  // ranges / tile_order of this block are written (and every thread has read the generation)
        const uint32_t nds = min(s_deep_n, (uint32_t)kDeepSorted);
        uint32_t my_n = 0, my_tile = 0;
        if ((uint32_t)tid < nds) {
            const uint2 me = s_sorted[tid];  // (depth the tile's list was consumed to, tile)
            my_tile = me.y;
            const uint32_t len = ranges[2 * my_tile + 1] - ranges[2 * my_tile];
            my_n = seg_count_for(min(len, me.x + 256u));
        }
        // the budget of record slots goes to the deepest tiles: a tile is cut if it and everything in front of it fit
        const uint32_t want_inc = block_scan_1024(my_n, s_wave);
        if (want_inc > kSegMax) my_n = 0;
        const uint32_t packed = my_n | (my_n ? 1u << 16 : 0u);  // (segments | tiles cut): both sums stay below 2^16
        const uint32_t inc = block_scan_1024(packed, s_wave);
        const uint32_t at = (inc - packed) & 0xFFFFu;
        if (tid < kDeepSorted) s_before[tid] = (inc - packed) >> 16;
        if (tid == 1023) s_tot = inc;
        for (int i = b; i < e; i++) {
            tile_slot[i] = 0xFFFFFFFFu;
            arrive[i] = 0u;
        }
        __syncthreads();
        const uint32_t n_segs = s_tot & 0xFFFFu, n_cut = s_tot >> 16;
        if (my_n) {
            tile_slot[my_tile] = at | (my_n << 16);
            for (uint32_t j = 0; j < my_n; j++) items[at + j] = my_tile | (j << kItemTileBits) | (my_n << 24);
        }
        __syncthreads();
        for (int q = b; q < e; q++) {  // the tiles that stay whole, in tile order
            const uint32_t t = tile_order[q];
            if (tile_slot[t] != 0xFFFFFFFFu) continue;
            const uint32_t cut_before = (uint32_t)q < nds ? s_before[q] : n_cut;
            items[n_segs + (uint32_t)q - cut_before] = t;
        }
        if (tid == 0) {
            sctl[SEG_CTL_WORK] = n_segs + (uint32_t)T - n_cut;
            sctl[SEG_CTL_SEGMENTS] = n_segs;
            sctl[SEG_CTL_TILES] = n_cut;
            sctl[SEG_CTL_PARITY] = gen;
        }
    }
    if (tid == 1023) {
        header[HDR_NUM_RENDERED] = run;  // thread 1023 owns the last chunk of tiles: its running sum is the total
        header[HDR_STATUS] = view_at(sort_ctl, geom_stride, blockIdx.y)[SORT_CTL_OVERFLOW] ? (uint32_t)FNX_ERR_SORT_SPAN : 0u;
        header[HDR_CAPACITY] = 0u;
        header[HDR_NUM_STATIC] = st_starts ? st_starts[T] : 0u;
        header[HDR_BWD_ITEMS] = 0u;  // the blend forward appends the backward's work items
        header[HDR_BWD_TICKET] = 0u;  // the backward's dynamic ticket counter (view 0's words)
        header[HDR_BWD_DONE] = 0u;
        header[HDR_FWD_ENTRIES] = 0u;
        header[HDR_BWD_ENTRIES] = 0u;
    }
    // emission work items: exclusive prefix of the per-block band counts
    __syncthreads();
    const int bper = (NB + 1023) / 1024;
    const int bb = tid * bper, be = min(NB, bb + bper);
    uint32_t items = 0;
    for (int i = bb; i < be; i++) {
        const uint32_t inst = blk_total[i];
        if (inst) items += (uint32_t)max(1, min(min(kEmitBands, gy), (int)((inst + kEmitBandTarget - 1u) / kEmitBandTarget)));
    }
    const uint32_t items_inc = block_scan_1024(items, s_wave);
    uint32_t at = items_inc - items;
    for (int i = bb; i < be; i++) {
        const uint32_t inst = blk_total[i];
        if (!inst) continue;  // a block without instances emits nothing
        const uint32_t nbands = (uint32_t)max(1, min(min(kEmitBands, gy), (int)((inst + kEmitBandTarget - 1u) / kEmitBandTarget)));
        for (uint32_t j = 0; j < nbands; j++) emit_items[at++] = emit_item_pack((uint32_t)i, j, nbands);
    }
    if (tid == 1023) {
        view_at(emit_ctl, geom_stride, blockIdx.y)[EMIT_CTL_ITEMS] = items_inc;
        if (blockIdx.y == 0) emit_ctl[EMIT_CTL_TICKET] = 0u;  // one ticket counter for all views (view 0's word)
    }
}

// K5: front-to-back alpha blending, one 256-thread workgroup per 16x16 tile
// (ch3 forward.cu:249-373).  Wave w owns the 8x8 quadrant (w & 1, w >> 1), and each ROW of 16 lanes of the wave one
// 4x4-pixel block of it.  A batch of 256 list entries is staged once in LDS; while staging, each entry is tested
// against the sixteen blocks (block_mask_exact), and every block gets its own compacted, still depth-ordered list:
// the four rows of a wave walk FOUR DIFFERENT lists at the same time (the LDS reads take per-lane addresses), so
// a lane only evaluates entries that can reach its 4x4 block.  With plume-sized splats an entry reaches ~2.2 of the
// 4 blocks of a quadrant it touches, and ~22 of its 64 pixels: per-block lists cut the evaluated (pixel, entry)
// pairs by a third to a half against per-quadrant lists.  Per pixel the arithmetic and its order are exactly the
// reference's; culled pairs are pairs it would have skipped (alpha < 1/255).
//
// SPLIT (static-split mode, include/fnx_raster.h): a tile has TWO depth-ordered streams of (depth bits, id)
// pairs -- the static splats' (binned once per frame) and this call's -- and the kernel merges them lazily: while
// batch b is blended, the next 256 entries of the merged order are found by a merge-path search over the next 256
// candidates of each stream (held in LDS), their records are requested, and the candidates after them are
// prefetched.  Ties in depth go to the per-call stream (lower ids), as in the reference's stable sort of ids
// emitted in id order.  The merged ids of every batch that is blended are written to point_list (the backward
// pass walks exactly that prefix); `materialize_all` keeps merging and writing after the pixels are done.
#ifndef FNX_WALK_LIST_AHEAD
#define FNX_WALK_LIST_AHEAD 0  // measured: 1022 / 1014 / 1016 against 1010 / 1016 / 1020 it/s (config 3), 527 against 529 (config 5): nothing
#endif
#ifndef FNX_MERGE_FAST_PATH
#define FNX_MERGE_FAST_PATH 1
#endif
#ifndef FNX_FWD_WAVES
#define FNX_FWD_WAVES 4  // waves per SIMD the register allocation of the blend forward aims at (4: 469 us, 3: 486 us on config 3)
#endif
//
// FAST (fnx_set_blend_math(1), "stated-tolerance" arithmetic): the same lists, the same decisions, but per (pixel, entry)
// the power is a chain of fused multiply-adds on coefficients pre-scaled by log2(e) at staging time, the opacity enters as
// log2(o) inside the exponent, and the exponential is ONE v_exp_f32 (<= 1 ulp) instead of the 14-instruction fixed
// sequence; T is updated as T - alpha T, the median depth is found from a per-batch count instead of per entry.
// ~30 instead of ~54-65 VALU instructions per entry.  Pixels agree with the exact mode to ~1e-6 except where a rounding
// moves an alpha across 1/255 or a T across 1e-4 (tests/test_fast_math_gpu.py states and checks the tolerance).
//
// SEG (fnx_raster_opts_t.segment_scratch; FAST, not DUAL): the launch takes its work from the list tile_scan_kernel laid down
// in the segment scratch -- segments of the deep tiles first, then the other tiles whole.  A segment is blended like a
// tile that starts at its first batch with transmittance 1 and colour 0; the segment of a tile that arrives last puts
// the segments together (fnx_state.h), and where a pixel's walk did depend on the transmittance in front of its segment
// -- the stop rule fires inside it, the median-depth entry lies in it -- the rest of the tile's list is blended again,
// by this workgroup, from the first such segment and from the true state in front of it.
// Stores / loads at device scope (sc1: past the XCD's L2), for data another workgroup -- on another XCD -- reads within
// the same launch behind an arrival counter.  A release fence instead (__threadfence: buffer_wbl2) writes back EVERYTHING
// dirty in the XCD's L2 -- during a kernel that streams megabytes of stores it took tens of microseconds per workgroup.
__device__ __forceinline__ void store_dev(float4 *p, const float4 v) {
    float *f = reinterpret_cast<float *>(p);
    __hip_atomic_store(f + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(f + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(f + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(f + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_dev(uint4 *p, const uint4 v) {
    uint32_t *f = reinterpret_cast<uint32_t *>(p);
    __hip_atomic_store(f + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(f + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(f + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(f + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 load_dev(const float4 *p) {
    float *f = const_cast<float *>(reinterpret_cast<const float *>(p));
    return make_float4(__hip_atomic_load(f + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(f + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ uint4 load_dev(const uint4 *p) {
    uint32_t *f = const_cast<uint32_t *>(reinterpret_cast<const uint32_t *>(p));
    return make_uint4(__hip_atomic_load(f + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                      __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                      __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                      __hip_atomic_load(f + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
#ifdef FNX_EXP_SEG_WHY  // developer counters: pixels whose segment walk was not trusted, by reason (ctl words 6 .. 10)
#define FNX_SEG_WHY(i) atomicAdd(&sctl[i], 1u);
#else
#define FNX_SEG_WHY(i)
#endif
template <int C, bool SPLIT, bool FAST, bool DUAL = false, bool SEG = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FNX_FWD_WAVES, FNX_FWD_WAVES)))
blend_forward_kernel(int T, int gx, const uint32_t *__restrict__ ranges, uint32_t *__restrict__ point_list, int W,
                     int H, const float4 *__restrict__ blend_rec, const float *__restrict__ bg,
                     float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                     float *__restrict__ out_color, float *__restrict__ out_depth, uint32_t *__restrict__ header,
                     uint32_t capacity, uint32_t *__restrict__ status_out, const uint32_t *__restrict__ tile_count,
                     const uint32_t *__restrict__ dyn_start, float *__restrict__ acc_final,
                     const uint32_t *__restrict__ tile_order, const uint8_t *__restrict__ tile_deep,
                     uint32_t *__restrict__ depth_hint, const StaticRef st, int materialize_all, const ViewBatch vb,
                     int skip_deep, uint32_t dyn_limit, const InvUpdate iu, const DualRef du, const SegRef sg) {
    static_assert(!SEG || (FAST && !DUAL), "segmented tiles: fast arithmetic, one image");
    const char *static_blob = nullptr;
    // Workgroup -> (view, rank in the view's tile order).  The hardware dispatches workgroups in linear order
    // (x fastest), and each view's order starts with its deep tiles: with the view as the slow grid dimension the
    // last view's longest walks would only START once every other view has been handed out.  So the views are
    // interleaved in chunks of 8 consecutive workgroups (one per XCD: rank % 8 stays the XCD the tile order was
    // built for), and the deep tiles of ALL views lead the launch.
    const int wg_linear = blockIdx.y * gridDim.x + blockIdx.x, n_views = gridDim.y;
    const int wg_view = (wg_linear >> 3) % n_views, wg_rank = ((wg_linear >> 3) / n_views) * 8 + (wg_linear & 7);
    if (!SEG && wg_rank >= T) return;  // gridDim.x is T rounded up to a multiple of 8
    // temporal-coherence depth sort (raster_binning.hip): the rank every splat has in this call's order, for the next
    // call's preprocess.  The T workgroups of a view share the P ranks.  FNX_INV_LATE writes them at the END of the workgroup
    // (scattered 4-byte stores in flight sit in front of every later load in the in-order vmcnt queue): measured, no change.
#ifndef FNX_INV_LATE
#define FNX_INV_LATE 0  // measured (round 6, A/B in one gpurun call): 277.8 / 276.7 us late against 278.4 / 275.0 early -- nothing
#endif
    auto write_inv = [&]() {
        if (iu.pairs && wg_rank < T) {
            const uint2 *pr = view_at(iu.pairs, vb.geom, wg_view);
            uint32_t *inv = reinterpret_cast<uint32_t *>(iu.state + iu.stride * (size_t)wg_view + iu.inv);
            const int per = (iu.P + T - 1) / T, r1_ = min(iu.P, (wg_rank + 1) * per);
            for (int r = wg_rank * per + (int)threadIdx.x; r < r1_; r += 256) {
                const uint32_t id = pr[r].y;
                if (id < (uint32_t)iu.P) inv[id] = (uint32_t)r;
            }
        }
    };
    if (!FNX_INV_LATE) write_inv();
    {
        const int vw = wg_view;
        ranges = view_at(ranges, vb.img, vw);
        final_T = view_at(final_T, vb.img, vw);
        n_contrib = view_at(n_contrib, vb.img, vw);
        header = view_at(header, vb.img, vw);
        point_list = view_at(point_list, vb.bin, vw);
        blend_rec = view_at(blend_rec, vb.geom, vw);
        out_color += (size_t)vw * C * H * W;
        out_depth += (size_t)vw * H * W;
        acc_final = view_at(acc_final, vb.img, vw);
        tile_order = view_at(tile_order, vb.img, vw);
        tile_deep = view_at(tile_deep, vb.img, vw);
        if (depth_hint) depth_hint += (size_t)vw * T;
        if (SPLIT) {
            tile_count = view_at(tile_count, vb.img, vw);
            dyn_start = view_at(dyn_start, vb.img, vw);
            static_blob = st.base + st.stride * vw;
        }
    }
    // DUAL: the second image's per-pixel arrays (an image blob of its own per view, same layout and stride)
    char *img1 = DUAL ? du.img1 + vb.img * (size_t)wg_view : nullptr;
#ifndef FNX_FWD_GROUP
#define FNX_FWD_GROUP 4
#endif
    constexpr int kGroup = FNX_FWD_GROUP;  // list entries per step of the blend loop
    // staged batch: three 16-byte records per slot; slot 256 is a NULL record (opacity 0 at the origin: alpha = 0)
    // that the tail of every block list points to, so the blend loop needs no end-of-list test per entry
    __shared__ float4 s_ra[257];  // x, y, conic a, conic b
    __shared__ float4 s_rb[257];  // conic c, opacity, exp-skip threshold, -
    __shared__ float4 s_rc[257];  // colour (C channels), depth in .w
    // per-block lists of LDS byte offsets (slot * 16) of the entries that can reach the block, depth order; lists
    // 4 w .. 4 w + 3 are built, padded (NULL record) and read by wave w alone
    constexpr int kListStride = (256 + kGroup + 7) & ~7;
    __shared__ __attribute__((aligned(16))) uint16_t s_list[16][kListStride];
    __shared__ uint16_t s_mask[256];  // block mask of every staged entry
    __shared__ uint32_t s_wk[SPLIT ? 2 : 1][SPLIT ? 256 : 1];  // merge windows: depth bits [static | per-call]
    __shared__ uint32_t s_wi[SPLIT ? 2 : 1][SPLIT ? 256 : 1];  //                ids
    __shared__ uint32_t s_adv;                                 // static entries among the batch just merged
    __shared__ uint32_t s_qmax[4];
    __shared__ uint32_t s_done[4];
    __shared__ unsigned long long s_dynmask[4];  // per staging wave: which of its 64 slots hold dynamic entries
    __shared__ uint32_t s_qdyn[4];
    // the view's header words (instance count, status, capacity) for the caller's deferred status check: the last
    // kernel of the forward copies them out, which saves the caller a strided device-to-device copy per call
    if (wg_rank == 0 && threadIdx.x == 0) {
        header[HDR_BIN_CAPACITY] = capacity;  // the backward checks its own against it
        header[HDR_DYN_LIMIT] = dyn_limit;    // ... and its gradient limit against this one
    }
    if (status_out && wg_rank == 0 && threadIdx.x < 8)
        status_out[8 * wg_view + threadIdx.x] = threadIdx.x == HDR_BIN_CAPACITY ? capacity : header[threadIdx.x];
    if (header[HDR_NUM_RENDERED] > capacity || header[HDR_STATUS] == (uint32_t)FNX_ERR_SORT_SPAN) {
        if (FNX_INV_LATE) write_inv();
        return;
    }
    // tile order of the view (tile_scan_kernel): tiles that went deep last time first, then the XCD-aware order.  The
    // waves of a deep tile raise their priority: the launch ends when the longest sequential walk ends, and a walk
    // that shares its SIMDs with four short tiles on equal terms takes several times longer than it has to.
    uint32_t seg = 0, nseg = 0;  // SEG: this workgroup's segment of the tile and the tile's segment count (0: the whole tile)
    // ... generation of this call's hints / slots, the tile's first record slot, its slot | segments << 16 of the previous call
    uint32_t seg_gen = 0, seg_slot0 = 0, seg_old = 0xFFFFFFFFu;
    char *seg_scratch = nullptr;
    int tile_ = 0;
    if (SEG) {
        seg_scratch = sg.base + sg.stride * (size_t)wg_view;
        const uint32_t *sctl = reinterpret_cast<const uint32_t *>(seg_scratch + sg.L.ctl);
        if ((uint32_t)wg_rank >= sctl[SEG_CTL_WORK]) {
            if (FNX_INV_LATE) write_inv();
            return;
        }
        const uint32_t item = reinterpret_cast<const uint32_t *>(seg_scratch + sg.L.items)[wg_rank];
        tile_ = (int)(item & kItemTileMask);
        seg = (item >> kItemTileBits) & 0x3FFu;
        nseg = item >> 24;
        if (nseg) {
            seg_gen = sctl[SEG_CTL_PARITY] & 1u;
            const uint32_t *ts = reinterpret_cast<const uint32_t *>(seg_scratch + sg.L.tile_slot);
            seg_slot0 = ts[(size_t)seg_gen * T + tile_] & 0xFFFFu;
            seg_old = ts[(size_t)(seg_gen ^ 1u) * T + tile_];
        }
    } else {
        tile_ = (int)tile_order[wg_rank];
    }
    const int tile = tile_;
    if (tile_deep[tile]) {
        // blend_forward_deep_kernel / blend_forward_ws_kernel takes the tiles that went deep in the previous forward (the
        // first skip_deep of the view's order at most: they lead it)
        if (wg_rank < skip_deep) {
            if (FNX_INV_LATE) write_inv();
            return;
        }
        if (tile_deep[tile] >= 2) __builtin_amdgcn_s_setprio(FNX_DEEP_PRIO);
    }
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // The blend loop multiplies the colour of an entry a pixel does NOT take by alpha = 0 instead of selecting per
    // channel (colours are assumed finite, like everywhere else).
    if (tid == 0) {
        s_ra[256] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_rb[256] = FAST ? make_float4(0.f, -200.0f, 0.f, 0.f) : make_float4(0.f, 0.f, -87.0f, 0.f);  // FAST: log2(o) = -200
        s_rc[256] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int row = lane >> 4;  // the wave's 4x4 block this lane belongs to (blend_pixel, fnx_device.h)
    const int px = tx * FNX_TILE_X + blend_pixel_x(w, lane), py = ty * FNX_TILE_Y + blend_pixel_y(w, lane);
    const bool inside = px < W && py < H;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const float pxf = (float)px, pyf = (float)py;
    const float tile_x0 = (float)(tx * FNX_TILE_X), tile_y0 = (float)(ty * FNX_TILE_Y);
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    // the part of the list this workgroup blends: all of it, or (SEG) a segment's batches
    uint32_t b_lo = r0, b_hi = r1;
    if (SEG && nseg) {
        b_lo = r0 + 256u * seg_first_batch(seg);
        b_hi = min(r1, r0 + 256u * seg_first_batch(seg + 1u));
    }
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // 1 while the pixel is still blending, 0 once it has stopped (T would drop below 1e-4) or if it is outside the image:
    // an arithmetic mask instead of a predicate, so that the recurrence below needs no lane-mask logic.
    // FAST: `alive` is the WORKING transmittance (T while blending, 0 once stopped / outside), Tr the pixel's T.
    float alive = inside ? 1.0f : 0.0f;
    float Tr = 1.0f;
    const uint32_t nseg_t = nseg;  // segments of this workgroup's tile (nseg is cleared once they have been put together)
    // SEG: the working T the walk of segment s2 of this tile starts from (fnx_state.h)
    auto seg_hint = [&](uint32_t s2) -> float {
        if (!inside) return 0.0f;
        if (s2 == 0u || seg_old == 0xFFFFFFFFu || s2 >= (seg_old >> 16)) return 1.0f;
        const float h = reinterpret_cast<const float *>(seg_scratch + sg.L.hint)[((size_t)(seg_gen ^ 1u) * kSegMax +
                                                                                  (seg_old & 0xFFFFu) + s2) * 256 + threadIdx.x];
        return fminf(1.0f, h);
    };
    float *hint_new = SEG ? reinterpret_cast<float *>(seg_scratch + sg.L.hint) + ((size_t)seg_gen * kSegMax + seg_slot0) * 256 + threadIdx.x
                          : nullptr;
    // second round of a cut tile: the segment this pixel joins at (0xFFFF: it does not, 0xFFFE: it has)
    uint32_t my_act = 0xFFFFu;
    float seg_vstop = 0.0f;  // a segment's walk: the test value of the entry that stopped the pixel (fast_walk REC)
    if (SEG && nseg && seg) {
        alive = seg_hint(seg);
        Tr = alive;
    }
    uint32_t last_contributor = 0;
    // Splats with id >= dyn_limit take no gradient (the frozen background of the position stages).  A backward pass that
    // differentiates only ids below it needs nothing from the entries BEHIND a pixel's last such ("dynamic") entry: what
    // lies behind an entry enters its gradient only through (final colour - prefix) and final T, which the forward
    // stores.  last_dyn: list position (1-based) of the last dynamic entry at or in front of the pixel's last
    // contributor, 0 if none -- found per BATCH from a ballot of the staged entries' flags, not per entry.
    uint32_t last_dyn = 0, dyn_before = 0;  // dyn_before (workgroup-uniform): last dynamic position of the earlier batches
    float acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
    float Dm = 15.0f;  // median depth default (ch3 forward.cu:295)
    DualPixel d1;  // DUAL: the second image's pixel (same conventions as alive / Tr / acc / Dm above)
    d1.acc = 0.f;
    d1.Tr = 1.0f;
    d1.alive = inside ? 1.0f : 0.0f;
    d1.Dm = 15.0f;
    d1.hit_off = 0xFFFFFFFFu;
    uint32_t last_contributor1 = 0;
    // Software pipeline over batches: the records of batch b+1 (and the list ids of batch b+2) are
    // requested from memory before batch b is blended, so only the first batch pays the two
    // dependent global-memory latencies (id -> record) on the tile's critical path.
    float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa, pc = pa;
    float pd = 0.f;
    uint32_t id_ahead = 0;
    // SPLIT state: the two streams of the tile, how far each has been merged, the next window of each in registers
    const uint2 *sp = nullptr, *fp = nullptr;
    const float4 *rec_s = nullptr;
    uint32_t ns = 0, nf = 0, si = 0, fj = 0, my_id = 0;
    uint2 ws = make_uint2(0u, 0u), wf = ws;
    auto record_of = [&](uint32_t id) -> const float4 * {
        return (SPLIT && id >= st.id0) ? rec_s + 4 * (size_t)(id - st.id0) : blend_rec + 4 * (size_t)id;
    };
    auto load_windows = [&]() {  // the next (up to) 256 entries of each stream behind (si, fj)
        ws = (si + (uint32_t)tid < ns) ? sp[si + tid] : make_uint2(0xFFFFFFFFu, 0u);
        wf = (fj + (uint32_t)tid < nf) ? fp[fj + tid] : make_uint2(0xFFFFFFFFu, 0u);
    };
    auto store_windows = [&]() {
        s_wk[0][SPLIT ? tid : 0] = ws.x;
        s_wi[0][SPLIT ? tid : 0] = ws.y;
        s_wk[SPLIT ? 1 : 0][SPLIT ? tid : 0] = wf.x;
        s_wi[SPLIT ? 1 : 0][SPLIT ? tid : 0] = wf.y;
    };
    // Merge path: slot t of the next batch holds the (t+1)-th smallest of the two windows in LDS.  i = number of
    // static entries among the first t; a static entry precedes a per-call one only if its depth bits are SMALLER.
    auto merge_batch = [&](uint32_t cnt_next) -> uint32_t {
        uint32_t id = 0;
        if ((uint32_t)tid < cnt_next) {
            const uint32_t nsw = min(256u, ns - si), nfw = min(256u, nf - fj);
            const uint32_t *ks = s_wk[0], *kf = s_wk[SPLIT ? 1 : 0];
            const uint32_t t = (uint32_t)tid;
#if FNX_MERGE_FAST_PATH
            // One stream alone fills the batch (inside a plume: per-call entries only, the frozen background lies behind it):
            // two window reads instead of the search's ~9 dependent pairs -- a tenth of a deep tile's chain per batch.
            if (nfw >= cnt_next && (nsw == 0u || kf[cnt_next - 1u] <= ks[0])) {
                if (t == 0u) s_adv = 0u;
                return s_wi[SPLIT ? 1 : 0][t];
            }
            if (nsw >= cnt_next && (nfw == 0u || ks[cnt_next - 1u] < kf[0])) {
                if (t == 0u) s_adv = cnt_next;
                return s_wi[0][t];
            }
#endif
            uint32_t lo = t > nfw ? t - nfw : 0u, hi = min(t, nsw);
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ks[mid] < kf[t - mid - 1]) lo = mid + 1; else hi = mid;
            }
            const uint32_t i = lo, j = t - lo;
            const bool from_static = !(j < nfw && (i >= nsw || kf[j] <= ks[i]));
            id = from_static ? s_wi[0][i] : s_wi[SPLIT ? 1 : 0][j];
            if (t == cnt_next - 1) s_adv = i + (from_static ? 1u : 0u);
        }
        return id;
    };
    uint32_t staged = 0;   // list entries of the batches this workgroup blended
#ifdef FNX_EXP_CLOCK
    const unsigned long long wg_t0 = wall_clock64();
#endif
again:  // (SEG: the workgroup that put a tile's segments together comes back here to blend the rest of the list again)
    if (SPLIT) {
        const uint32_t *starts = reinterpret_cast<const uint32_t *>(static_blob + st.starts);
        const uint32_t s0 = starts[tile];
        ns = starts[tile + 1] - s0;
        sp = reinterpret_cast<const uint2 *>(static_blob + st.pairs) + s0;
        rec_s = reinterpret_cast<const float4 *>(static_blob + st.rec);
        nf = tile_count[tile];
        fp = reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(point_list) + vb.bin_pairs) + dyn_start[tile];
        si = fj = 0;
        if (SEG && b_lo != r0) {
            // where the two streams stand behind the first d merged entries: merge path, 64 probes per round (one wave)
            const uint32_t d = b_lo - r0;
            if (w == 0) {
                uint32_t lo = d > nf ? d - nf : 0u, hi = min(d, ns);
                while (lo < hi) {
                    const uint32_t step = (hi - lo + 63u) / 64u, m = lo + (uint32_t)lane * step;
                    const bool p = m < hi && sp[m].x < fp[d - m - 1u].x;  // static entry m lies in front of merged position d
                    const uint32_t c = (uint32_t)__popcll(__ballot(p));   // the probes that hold form a prefix
                    const uint32_t nlo = c ? lo + (c - 1u) * step + 1u : lo;
                    hi = (c < 64u && lo + c * step < hi) ? lo + c * step : hi;
                    lo = nlo;
                }
                if (lane == 0) s_adv = lo;
            }
            __syncthreads();
            si = s_adv;
            fj = d - si;
            __syncthreads();
        }
        // first batch: merge, request its records, prefetch the windows behind it
        load_windows();
        store_windows();
        __syncthreads();
        const uint32_t cnt0 = min(256u, b_hi - b_lo);
        my_id = merge_batch(cnt0);
        __syncthreads();
        if (cnt0) {
            const uint32_t a = s_adv;
            si += a;
            fj += cnt0 - a;
        }
        if ((uint32_t)tid < cnt0) {
            const float4 *rec = record_of(my_id);
            pa = rec[0];
            pb = rec[1];
            pc = rec[2];
            if (C > 2) pd = rec[3].x;
        }
        load_windows();
    } else {
        if (b_lo + (uint32_t)tid < b_hi) {
            my_id = point_list[b_lo + tid];
            const float4 *rec = blend_rec + 4 * (size_t)my_id;
            pa = rec[0];
            pb = rec[1];
            pc = rec[2];
            if (C > 2) pd = rec[3].x;
        }
        if (b_lo + 256u + (uint32_t)tid < b_hi) id_ahead = point_list[b_lo + 256u + tid];
    }
    // forward -> backward hand-over (fnx_state.h, kBlendBatch): per-pixel state in front of every batch after the first
    float4 *bstate = reinterpret_cast<float4 *>(reinterpret_cast<char *>(point_list) + vb.bin_bstate) +
                     (size_t)(r0 >> 8) * 256 + tid;
    float2 *bstate1 = DUAL ? reinterpret_cast<float2 *>(reinterpret_cast<char *>(point_list) + du.bin_bstate1) +
                                 (size_t)(r0 >> 8) * 256 + tid : nullptr;
#ifdef FNX_EXP_CLOCK
    unsigned long long t_last = clock64();
    if (wg_rank == 0 && wg_view == 0 && lane == 0) { for (int i = 0; i < 16; i++) g_fwd_clock[16 * w + i] = 0; g_fwd_clock[16 * w + 15] = r1 - r0; }
#endif
    bool blending = true;  // SPLIT + materialize_all: false once every pixel is done (merging and writing go on)
    for (uint32_t base = b_lo; base < b_hi; base += 256) {
        FNX_CLK(0)
        // barrier between the previous batch's walk and this batch's staging, and "has every pixel stopped?" in one:
        // each wave leaves its own answer in LDS before the barrier.  LDS-only barriers in this loop (lds_barrier): a
        // plain __syncthreads() also drains the wave's global stores (bstate, masks, point_list) and record prefetches,
        // an L2 round trip on the critical path of every batch, although no wave reads another's global data here.
        if (SEG && nseg_t && !nseg) {
            // second round of a cut tile (see behind the loop): a pixel joins at the first batch of the segment its walk was not
            // trusted from, with the state the segments in front of it add up to; the hints of the boundaries it passes
            const uint32_t j = (base - r0) >> 8;
            if (j >= kSegBatches0 && (j - kSegBatches0) % kSegBatches == 0u) {
                const uint32_t s2 = 1u + (j - kSegBatches0) / kSegBatches;
                if (my_act == s2) {  // its registers hold the state in front of this segment; it was only kept from blending
                    alive = Tr;
                    my_act = 0xFFFEu;
                }
                if (s2 < nseg_t && alive != 0.0f) hint_new[(size_t)s2 * 256] = alive;
            }
        }
        const uint32_t wave_done = __all(alive == 0.0f && (!DUAL || d1.alive == 0.0f) && (!SEG || my_act >= 0xFFFEu)) ? 1u : 0u;  // a vote of all 64 lanes: taken outside the branch
        if (lane == 0) s_done[w] = wave_done;
        FNX_LOOP_BARRIER();
        const bool all_done = (s_done[0] & s_done[1] & s_done[2] & s_done[3]) != 0u;
        FNX_CLK(1)
        if (all_done) {
            if (!SPLIT || !materialize_all) break;
            blending = false;
        }
        const uint32_t cnt = min(256u, b_hi - base);
        if (blending) staged += cnt;
        uint32_t qm = 0;
        {  // which staged entries are dynamic: one ballot per wave (slot = thread), read back behind the walk
            const unsigned long long dm = __ballot((uint32_t)tid < cnt && my_id < dyn_limit);
            if (lane == 0) s_dynmask[w] = dm;
        }
        if ((uint32_t)tid < cnt && blending) {
            qm = block_mask_exact(pa.x, pa.y, pa.z, pa.w, pb.x, pb.z, pc.x, pc.y, tile_x0, tile_y0);
            if (FAST) {
                // log2(e) power = A dx^2 + B dx dy + C dy^2 with (A, B, C) = -log2(e) (a / 2, b, c / 2); alpha = 2^(that + log2 o)
                constexpr float kL2e = 1.44269504088896341f;
                s_ra[tid] = make_float4(pa.x, pa.y, (-0.5f * kL2e) * pa.z, (-kL2e) * pa.w);
                const float lo = __builtin_amdgcn_logf(fmaxf(pb.y, 0.0f));  // v_log_f32; o = 0 -> -inf -> alpha = 0
                if (C == 3) {
                    s_rb[tid] = make_float4((-0.5f * kL2e) * pb.x, lo, pc.z, pc.w);
                    s_rc[tid] = make_float4(pd, pb.w, 0.f, 0.f);  // colour 2, depth
                } else {
                    s_rb[tid] = make_float4((-0.5f * kL2e) * pb.x, lo, pc.z, pb.w);  // colour 0, depth
                }
            } else {
                s_ra[tid] = pa;
                s_rb[tid] = pb;
                s_rc[tid] = make_float4(pc.z, C > 1 ? pc.w : 0.f, C > 2 ? pd : 0.f, pb.w);
            }
        }
        if (SPLIT) {
            store_windows();
        } else {
            if (base + 256u + (uint32_t)tid < b_hi) {  // next batch's records: in flight while this batch is blended
                my_id = id_ahead;
                const float4 *rec = blend_rec + 4 * (size_t)id_ahead;
                pa = rec[0];
                pb = rec[1];
                pc = rec[2];
                if (C > 2) pd = rec[3].x;
            }
            if (base + 512u + (uint32_t)tid < b_hi) id_ahead = point_list[base + 512u + tid];
        }
        s_mask[tid] = (uint16_t)qm;
        {  // this wave's four lists start out as NULL pointers (slot 256) from end to end
            const uint4 nul = make_uint4(0x10001000u, 0x10001000u, 0x10001000u, 0x10001000u);
            uint4 *mine = reinterpret_cast<uint4 *>(&s_list[4 * w][0]);
            for (int i = lane; i < 4 * kListStride / 8; i += 64) mine[i] = nul;
        }
        FNX_CLK(4)
        FNX_LOOP_BARRIER_BC();
        FNX_CLK(5)
        uint32_t len[4] = {0u, 0u, 0u, 0u};  // wave-uniform lengths of the wave's lists
        // a block whose 16 pixels have all stopped gets an empty list: the wave's step count is its longest list, and a
        // finished block must not be the one that keeps it walking
        const unsigned long long live = __ballot(alive != 0.0f || (DUAL && d1.alive != 0.0f));
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t mk = (uint32_t)s_mask[64 * k + lane] >> (4 * w);
            // DUAL: a static splat's entry is marked in its list word (staging wave k left the batch's dynamic flags)
            const uint32_t flag = (DUAL && !((s_dynmask[k] >> lane) & 1ull)) ? kListStatic : 0u;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const bool bit = ((mk >> b) & 1u) && ((live >> (16 * b)) & 0xFFFFull) != 0ull;
                const unsigned long long m = __ballot(bit);
                if (bit) s_list[4 * w + b][len[b] + (uint32_t)__popcll(m & lt_mask)] = (uint16_t)(((64 * k + lane) * 16) | flag);
                len[b] += (uint32_t)__popcll(m);
            }
        }
        FNX_CLK(6)
        uint32_t next_cnt = 0, next_id = 0;
        if (SPLIT) {
            next_cnt = base + 256u < b_hi ? min(256u, b_hi - base - 256u) : 0u;
            next_id = merge_batch(next_cnt);
        }
        FNX_CLK(7)
        // The batch's global STORES go out here, behind everything that waits for a load: s_waitcnt vmcnt counts loads and
        // stores in order, so a store issued in front of the mask computation (which waits for the batch's records) or of the
        // windows' LDS copies (which wait for the windows) was waited for as well -- a store's round trip on the chain of
        // every batch (round 5).  Nothing waits on vmcnt between here and the next batch's staging.
#ifndef FNX_EXP_FWD_NOSTORE  // timing experiment (backward unusable): 1 no hand-over records, 2 no ids / masks either
#define FNX_EXP_FWD_NOSTORE 0
#endif
        if (!(FNX_EXP_FWD_NOSTORE & 1))
        if (blending && base != r0 && !(SEG && nseg_t && !nseg && alive == 0.0f)) {  // (second round of a cut tile: the pixels it blends)
            // hand-over record: the pixel's state in front of this batch (the walk below changes it)
            if (SEG && nseg)  // a segment: the workgroup that puts the tile together reads it in this launch
                store_dev(&bstate[(size_t)(((base - r0) >> 8) - 1) * 256], make_float4(Tr, acc[0], acc[C > 1 ? 1 : 0], acc[C > 2 ? 2 : 0]));
            else
            bstate[(size_t)(((base - r0) >> 8) - 1) * 256] = make_float4(Tr, acc[0], acc[C > 1 ? 1 : 0], acc[C > 2 ? 2 : 0]);
            if (DUAL) bstate1[(size_t)(((base - r0) >> 8) - 1) * 256] = make_float2(d1.Tr, d1.acc);
        }
        if (!(FNX_EXP_FWD_NOSTORE & 2))
        if (SPLIT && (uint32_t)tid < cnt) point_list[base + tid] = my_id;  // the merged order, as far as it is consumed
        if (!(FNX_EXP_FWD_NOSTORE & 2))
        if ((uint32_t)tid < cnt && blending)  // the backward pass builds its lists from the same masks
            reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(point_list) + vb.bin_masks)[base + tid] = (uint16_t)qm;
        FNX_LOOP_BARRIER_BC();
        FNX_CLK(10)
        if (SPLIT) {
            if (next_cnt) {
                const uint32_t a = s_adv;
                si += a;
                fj += next_cnt - a;
            }
            my_id = next_id;
            {
                // next batch's records: in flight while this batch is blended.  UNCONDITIONAL loads (a thread beyond the batch
                // reads the record of splat 0, which it never stages): under `if (tid < next_cnt)` the compiler merged the loaded
                // values into the loop-carried registers with copies right behind the loads -- s_waitcnt vmcnt in front of the
                // walk, a memory round trip on the critical path of every batch (round 5).
#ifdef FNX_EXP_COND_PREFETCH  // timing experiment: the loads under the condition, as they were
                if ((uint32_t)tid < next_cnt)
#endif
                {
                const float4 *rec = record_of(my_id);
                pa = rec[0];
                pb = rec[1];
                pc = rec[2];
                if (C > 2) pd = rec[3].x;
                }
            }
            load_windows();
            if (!blending) continue;
        }
        const uint32_t n_w = max(max(len[0], len[1]), max(len[2], len[3]));  // steps of the wave = its longest list
        const uint16_t *mylist = s_list[4 * w + row];
        const uint32_t pos0 = base - r0 + 1;  // list position (1-based) of slot 0
        // The only state carried from entry to entry is (T, colour, depth, alive); power / exp / alpha of an entry do
        // not depend on it.  A lone wave issues one instruction every ~4.5 cycles, and the launch ends when the deepest
        // tile's walk ends, so the loop is written for few instructions per entry: kGroup entries per step, all LDS
        // reads (three 16-byte records per entry, addressed by the byte offsets in the list) issued together, the
        // alphas evaluated as straight-line code (the scheduler interleaves the independent chains), then the
        // recurrence in list order with the arithmetic mask `alive` instead of lane predicates:
        //   a = alpha if the entry hits the pixel (power <= 0, alpha >= 1/255) else 0;  ae = a * alive
        //   test_T = T (1 - ae);  stop = test_T < 1e-4  (T >= 1e-4 always, so stop implies ae > 0: forward.cu:336-340)
        //   an entry that is not applied (ae = 0 or stop) adds colour * 0 and leaves T as it is (T * 1 = T).
        // Per pixel the arithmetic on applied entries and its order are exactly the reference's.
        FNX_CLK(2)
#ifdef FNX_EXP_CLOCK
        if (wg_rank == 0 && wg_view == 0 && lane == 0) { g_fwd_clock[16 * w + 8] += n_w; g_fwd_clock[16 * w + 9] += 1; }
#endif
        uint32_t hit_off = 0xFFFFFFFFu;  // LDS offset of the last entry of this batch the pixel took
        d1.hit_off = 0xFFFFFFFFu;
        if (FAST) {
            if (SEG && nseg)
                fast_walk<C, false, true>(mylist, n_w, s_ra, s_rb, s_rc, pxf, pyf, acc, Tr, alive, Dm, hit_off, nullptr, &seg_vstop);
            else
            fast_walk<C, DUAL>(mylist, n_w, s_ra, s_rb, s_rc, pxf, pyf, acc, Tr, alive, Dm, hit_off, &d1);
        } else
        for (uint32_t i0 = 0; i0 < n_w; i0 += kGroup) {
            if (__all(alive == 0.0f && (!DUAL || d1.alive == 0.0f))) break;
            uint32_t jw[kGroup / 2];  // the next kGroup entries of this lane's list
#pragma unroll
            for (int k = 0; k < kGroup / 2; k++) jw[k] = reinterpret_cast<const uint32_t *>(mylist + i0)[k];
            float a_h[kGroup];
            float4 rc[kGroup];
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                const uint32_t off = (jw[k >> 1] >> (16 * (k & 1))) & (DUAL ? kListOffMask : 0xFFFFu);
                const float4 ra = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_ra) + off);
                const float4 rb = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rb) + off);
                rc[k] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rc) + off);
                const float dx = ra.x - pxf, dy = ra.y - pyf;
                const float power = -0.5f * (ra.z * dx * dx + rb.x * dy * dy) - ra.w * dx * dy;
                // below -87 the fixed exp is exactly 0 (alpha = 0 < 1/255): clamping keeps the guard-free exp in range
                const float alpha = fminf(0.99f, rb.y * exp_fixed_in_range(fmaxf(power, -87.0f)));
                a_h[k] = (!(power > 0.0f) && !(alpha < 1.0f / 255.0f)) ? alpha : 0.0f;
            }
#ifdef FNX_EXP_STATS
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                const unsigned long long hit = __ballot(a_h[k] > 0.0f && alive != 0.0f);
                const unsigned long long live = __ballot(alive != 0.0f);
                if (lane == 0 && i0 + k < n_w) {
                    atomicAdd(&g_fwd_stats[0], 1ull);
                    atomicAdd(&g_fwd_stats[1], (unsigned long long)__popcll(hit));
                    atomicAdd(&g_fwd_stats[2], hit ? 1ull : 0ull);
                    atomicAdd(&g_fwd_stats[4], (unsigned long long)__popcll(live));
                    atomicAdd(&g_fwd_stats[3], (unsigned long long)((i0 + k < len[0]) + (i0 + k < len[1]) + (i0 + k < len[2]) + (i0 + k < len[3])));
                    int halves = 0, rows_hit = 0;
                    for (int h = 0; h < 8; h++) halves += ((hit >> (8 * h)) & 0xFFull) != 0;
                    for (int h = 0; h < 4; h++) rows_hit += ((hit >> (16 * h)) & 0xFFFFull) != 0;
                    atomicAdd(&g_fwd_stats[5], (unsigned long long)halves);
                    atomicAdd(&g_fwd_stats[6], (unsigned long long)rows_hit);
                }
            }
#endif
            if (DUAL) {  // the second image: the same entries, static ones with alpha 0, its own T and stop rule
#pragma unroll
                for (int k = 0; k < kGroup; k++) {
                    const uint32_t raw = (jw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, off = raw & kListOffMask;
                    const float ae = ((raw & kListStatic) ? 0.0f : a_h[k]) * d1.alive;
                    const float test_T = d1.Tr * (1 - ae);
                    const bool stop = test_T < 0.0001f;
                    const float a_eff = stop ? 0.0f : ae;
                    d1.acc = d1.acc + rc[k].x * a_eff * d1.Tr;
                    d1.Dm = (d1.Tr > 0.5f && test_T < 0.5f) ? rc[k].w : d1.Dm;
                    d1.Tr = stop ? d1.Tr : test_T;
                    d1.hit_off = (a_eff > 0.0f) ? off : d1.hit_off;
                    d1.alive = stop ? 0.0f : d1.alive;
                }
            }
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                const uint32_t off = (jw[k >> 1] >> (16 * (k & 1))) & (DUAL ? kListOffMask : 0xFFFFu);
                const float ae = a_h[k] * alive;
                const float test_T = Tr * (1 - ae);
                const bool stop = test_T < 0.0001f;
                const float a_eff = stop ? 0.0f : ae;
                acc[0] = acc[0] + rc[k].x * a_eff * Tr;
                if (C > 1) acc[C > 1 ? 1 : 0] = acc[C > 1 ? 1 : 0] + rc[k].y * a_eff * Tr;
                if (C > 2) acc[C > 2 ? 2 : 0] = acc[C > 2 ? 2 : 0] + rc[k].z * a_eff * Tr;
                Dm = (Tr > 0.5f && test_T < 0.5f) ? rc[k].w : Dm;  // cannot hold for an entry that is not applied
                Tr = stop ? Tr : test_T;
                hit_off = (a_eff > 0.0f) ? off : hit_off;
                alive = stop ? 0.0f : alive;
            }
        }
        {
            // highest dynamic slot of the batch at or below slot `upto` (0xFFFFFFFF: none); wave ws staged slots 64 ws ..
            auto dyn_at_or_below = [&](uint32_t upto) -> uint32_t {
                int ws = (int)(upto >> 6);
                unsigned long long m = s_dynmask[ws] & ((2ull << (upto & 63u)) - 1ull);  // (2 << 63) wraps to 0: all ones
                while (m == 0ull && ws > 0) m = s_dynmask[--ws];
                return m ? (uint32_t)(64 * ws + 63 - __clzll((long long)m)) : 0xFFFFFFFFu;
            };
            if (hit_off != 0xFFFFFFFFu) {
                last_contributor = pos0 + (hit_off >> 4);  // list position (1-based) of that entry
                const uint32_t d = dyn_at_or_below(hit_off >> 4);
                last_dyn = d != 0xFFFFFFFFu ? pos0 + d : dyn_before;
            }
            const uint32_t d_all = dyn_at_or_below(255u);  // workgroup-uniform
            if (d_all != 0xFFFFFFFFu) dyn_before = pos0 + d_all;
            if (DUAL && d1.hit_off != 0xFFFFFFFFu) last_contributor1 = pos0 + (d1.hit_off >> 4);
        }
        FNX_CLK(3)
    }
    if (SEG && nseg) {
        // ---- a segment of a cut tile: leave the pixels' record, arrive; the last one to arrive goes on ----
        uint32_t *sctl = reinterpret_cast<uint32_t *>(seg_scratch + sg.L.ctl);
        float4 *rec_a = reinterpret_cast<float4 *>(seg_scratch + sg.L.rec_a) + (size_t)seg_slot0 * 256 + tid;
        uint4 *rec_b = reinterpret_cast<uint4 *>(seg_scratch + sg.L.rec_b) + (size_t)seg_slot0 * 256 + tid;
        uint4 *meta = reinterpret_cast<uint4 *>(seg_scratch + sg.L.meta) + seg_slot0;
        // (T behind the segment -- in front of the entry that stopped the pixel, if one did --, colour taken in it, both on the
        //  hint's scale | last contributor, last dynamic entry at or in front of it (0: none in this segment), median depth,
        //  working T behind the segment (0: stopped))
        store_dev(&rec_a[(size_t)seg * 256], make_float4(Tr, acc[0], acc[C > 1 ? 1 : 0], acc[C > 2 ? 2 : 0]));
        store_dev(&rec_b[(size_t)seg * 256], make_uint4(last_contributor, last_dyn, __float_as_uint(Dm), __float_as_uint(seg_vstop)));
        if (tid == 0) {
            store_dev(&meta[seg], make_uint4(dyn_before, 0u, 0u, 0u));
            if (staged) atomicAdd(&header[HDR_FWD_ENTRIES], staged);
        }
        staged = 0;
        // the record and this segment's hand-over records (device-scope stores, all of them) have been written when their
        // stores have returned: then the arrival
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) s_adv = atomicAdd(reinterpret_cast<uint32_t *>(seg_scratch + sg.L.arrive) + tile, 1u);
        __syncthreads();
        const bool last_to_arrive = s_adv == nseg - 1u;
        __syncthreads();
        if (!last_to_arrive) {
#ifdef FNX_EXP_CLOCK
            if (tid == 0) {
                const int wg = wg_view * (T + (int)kSegMax) + wg_rank;
                if (wg < 16384) {
                    g_fwd_wg[4 * wg] = wg_t0;
                    g_fwd_wg[4 * wg + 1] = wall_clock64();
                    g_fwd_wg[4 * wg + 2] = r1 - r0;
                    g_fwd_wg[4 * wg + 3] = ((unsigned long long)(seg | (nseg_t << 8)) << 48);
                }
            }
#endif
            if (FNX_INV_LATE) write_inv();
            return;
        }
#ifdef FNX_EXP_CLOCK
        if (tid == 0 && wg_view * (T + (int)kSegMax) + wg_rank < 16384) g_fwd_wg2[wg_view * (T + (int)kSegMax) + wg_rank] = wall_clock64();
#endif
        const uint32_t nb_t = (r1 - r0 + 255u) >> 8;  // batches of the tile
        // One pass per pixel over the tile's segments, in order, with the true state (T, colour, last contributor ...) in
        // front of each: with rho = true T_in / the hint the walk started from, every T of the walk is rho times what the walk
        // saw and the colour it took rho times its colour.  A segment's walk stands for the pixel if none of its decisions
        // depended on T_in: it did not stop the pixel, the smallest T its stop rule accepted -- the last: T only falls --
        // times rho is still >= 1e-4, and walk and truth agree on whether T crosses 1/2 in it (the median-depth entry is then
        // taken from the walk; it may be a neighbour of the true one: stated tolerance of this mode).  The first segment whose
        // walk does not stand is where the pixel is blended AGAIN, from the true state, in a second round by this workgroup
        // (the stop rule fires in that segment or soon behind it: a pixel's second round is about a segment long).  The first
        // segment starts from the truth and always stands.
        uint32_t first_bad = 0xFFFFu;
        {
            float T = 1.0f;
            bool live = inside;
            float ca[3] = {0.f, 0.f, 0.f};
            uint32_t lc = 0, ld = 0, db = 0;
            float dm = 15.0f;
            for (uint32_t s2 = 0; s2 < nseg; s2++) {
                const float4 a = load_dev(&rec_a[(size_t)s2 * 256]);
                const uint4 b = load_dev(&rec_b[(size_t)s2 * 256]);
                const uint32_t mdyn = load_dev(&meta[s2]).x;
                const float h = seg_hint(s2);
                const uint32_t j0 = seg_first_batch(s2), j1 = min(seg_first_batch(s2 + 1u), nb_t);
                if (first_bad == 0xFFFFu) {
                    hint_new[(size_t)s2 * 256] = live ? T : 0.0f;
                    float rho = 0.0f;
                    bool cross = false;
                    if (live) {
                        const float vstop = __uint_as_float(b.w);  // > 0: the walk stopped the pixel, at an entry with this test value
                        rho = h > 0.0f ? T / h : 0.0f;
                        const float t_end = rho * a.x;
                        cross = T >= 0.5f && t_end < 0.5f;
                        const bool cross_h = h >= 0.5f && a.x < 0.5f;
                        // the entry that stopped the pixel must stop it under the true T as well; the median-depth entry is
                        // only taken from a walk that saw T within 2 % of the truth
                        const bool stop_moves = vstop > 0.0f && !(rho * vstop < 0.0001f);
                        const bool depth_off = cross != cross_h || (cross && fabsf(rho - 1.0f) > 0.02f);
                        if (s2 > 0 && (h == 0.0f || stop_moves || t_end < 0.0001f || depth_off)) {
                            first_bad = s2;
                            FNX_SEG_WHY(h == 0.0f ? 6 : stop_moves ? 8 : t_end < 0.0001f ? 7 : 9)
                        }
                    }
                    if (first_bad == 0xFFFFu) {
                        if (s2 > 0) {
                            // the hand-over records of the segment's batches: "hint's scale, colour from 0" -> the tile's own
                            bstate[(size_t)(j0 - 1u) * 256] = make_float4(T, ca[0], ca[1], ca[2]);
                            for (uint32_t j = j0 + 1u; j < j1; j++) {
                                const float4 q = load_dev(&bstate[(size_t)(j - 1u) * 256]);
                                bstate[(size_t)(j - 1u) * 256] =
                                    live ? make_float4(rho * q.x, __builtin_fmaf(rho, q.y, ca[0]), __builtin_fmaf(rho, q.z, ca[1]),
                                                       __builtin_fmaf(rho, q.w, ca[2]))
                                         : make_float4(T, ca[0], ca[1], ca[2]);
                            }
                        }
                        if (live) {
                            if (cross) dm = __uint_as_float(b.z);
                            ca[0] = __builtin_fmaf(rho, a.y, ca[0]);
                            ca[1] = __builtin_fmaf(rho, a.z, ca[1]);
                            ca[2] = __builtin_fmaf(rho, a.w, ca[2]);
                            if (b.x) {
                                lc = b.x;
                                ld = b.y ? b.y : db;
                            }
                            live = __uint_as_float(b.w) == 0.0f;  // (no entry stopped it)
                            T = rho * a.x;
                        }
                        db = max(db, mdyn);
                    }
                } else {
                    hint_new[(size_t)s2 * 256] = 0.0f;  // (what the second round passes overwrites it)
                }
            }
            // the list goes on behind the segments (they cover what the tile consumed last time): pixels still blending go on
            if (first_bad == 0xFFFFu && live && seg_first_batch(nseg) < nb_t) {
                first_bad = nseg;
                FNX_SEG_WHY(10)
            }
            Tr = T;
            alive = live ? T : 0.0f;
#pragma unroll
            for (int ch = 0; ch < C; ch++) acc[ch] = ca[ch];
            last_contributor = lc;
            last_dyn = ld;
            Dm = dm;
            dyn_before = db;  // (per pixel here; made the tile's below)
        }
        if (tid == 0) s_adv = 0xFFFFu;
        __syncthreads();
        if (first_bad != 0xFFFFu) atomicMin(&s_adv, first_bad);
        __syncthreads();
        const uint32_t s_first = s_adv;  // the first segment any pixel joins the second round at
        __syncthreads();
        nseg = 0;
        if (s_first != 0xFFFFu) {
            // Second round: the batches from segment s_first on, once more, as ONE list; a pixel takes part from the segment
            // its walk was not trusted from (its state waits in the scratch until the loop reaches that segment) until the stop
            // rule -- applied to the true T now -- ends it.  A pixel's registers hold its state in front of that segment (the
            // final one, for a pixel that does not take part): a working T of 0 keeps it from blending until then.
            my_act = first_bad;
            alive = 0.0f;
            // the last dynamic list position in front of segment s_first (the tile's, not the pixel's)
            uint32_t db = 0;
            for (uint32_t s2 = 0; s2 < min(s_first, nseg_t); s2++) db = max(db, load_dev(&meta[s2]).x);
            dyn_before = db;
            b_lo = r0 + 256u * seg_first_batch(s_first);
            b_hi = r1;
            if (tid == 0) {
                atomicAdd(&sctl[SEG_CTL_REPAIRED], 1u);
                atomicAdd(&sctl[SEG_CTL_REPAIR_BATCHES], (r1 - min(b_lo, r1) + 255u) >> 8);
            }
            if (b_lo < r1) goto again;
        }
    }
    if (inside) {
        final_T[pix_id] = Tr;
        n_contrib[pix_id] = last_contributor;
        n_contrib[(size_t)W * H + pix_id] = last_dyn;  // second half of the array: what a limited backward walks to
#pragma unroll
        for (int ch = 0; ch < C; ch++) {
            out_color[(size_t)ch * H * W + pix_id] = acc[ch] + Tr * bg[ch];
            acc_final[(size_t)ch * H * W + pix_id] = acc[ch];
        }
        out_depth[pix_id] = Dm;
        if (DUAL) {  // the second image: every contributor of its is a dynamic entry, so both halves of n_contrib agree
            reinterpret_cast<float *>(img1 + du.final_T)[pix_id] = d1.Tr;
            uint32_t *nc1 = reinterpret_cast<uint32_t *>(img1 + du.n_contrib);
            nc1[pix_id] = last_contributor1;
            nc1[(size_t)W * H + pix_id] = last_contributor1;
            reinterpret_cast<float *>(img1 + du.acc_final)[pix_id] = d1.acc;
            du.out_color1[(size_t)wg_view * H * W + pix_id] = d1.acc + d1.Tr * du.bg1[0];
            du.out_depth1[(size_t)wg_view * H * W + pix_id] = d1.Dm;
        }
    }
    // one backward work item per batch that holds a DYNAMIC entry in front of some pixel's last contributor (the batches
    // behind hold nothing a backward pass within the gradient limit needs); qmax: how deep the tile went
    uint32_t m = DUAL ? max(last_contributor, last_contributor1) : last_contributor;
    uint32_t md = DUAL ? max(last_dyn, last_contributor1) : last_dyn;  // DUAL: the backward walks as far as either image needs
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        m = max(m, (uint32_t)__shfl_xor((int)m, off));
        md = max(md, (uint32_t)__shfl_xor((int)md, off));
    }
    if (lane == 0) {
        s_qmax[w] = m;
        s_qdyn[w] = md;
    }
    __syncthreads();
    const uint32_t qmax = max(max(s_qmax[0], s_qmax[1]), max(s_qmax[2], s_qmax[3]));
    const uint32_t nb = (max(max(s_qdyn[0], s_qdyn[1]), max(s_qdyn[2], s_qdyn[3])) + 255u) >> 8;
    if (nb) {
        if (tid == 0) s_adv = atomicAdd(&header[HDR_BWD_ITEMS], nb);
        __syncthreads();
        uint32_t *items = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(point_list) + vb.bin_items) + s_adv;
        for (uint32_t k = tid; k < nb; k += 256) items[k] = (uint32_t)tile | (k << kItemTileBits);
    }
    if (depth_hint && tid == 0) depth_hint[tile] = qmax;  // how deep the tile went: the next forward's tile order
    if (tid == 0 && staged) atomicAdd(&header[HDR_FWD_ENTRIES], staged);
    if (FNX_INV_LATE) write_inv();
#ifdef FNX_EXP_CLOCK
    if (tid == 0) {
        const int wg = wg_view * (T + (SEG ? (int)kSegMax : 0)) + wg_rank;
        if (wg < 16384) {
            g_fwd_wg[4 * wg] = wg_t0;
            g_fwd_wg[4 * wg + 1] = wall_clock64();
            g_fwd_wg[4 * wg + 2] = ((unsigned long long)qmax << 32) | (r1 - r0);
            g_fwd_wg[4 * wg + 3] = ((unsigned long long)(seg | (nseg_t << 8)) << 48) | ((unsigned long long)(nb & 0xFFFFu) << 32) | staged;
        }
    }
#endif
}

#include "lab/blend_forward_deep.h"  // K5b: opt-in super-batch forward for deep tiles (lab)
#include "raster_forward_ws.h"       // staging waves (deep_kernel = 3 / 4)

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

template <int C>
static void launch_preprocess_c(hipStream_t s, int P, int D, int M, const float *means3D, const float *scales,
                                float scale_modifier, const float *rotations, const float *opacities, const float *shs,
                                uint8_t *clamped, const float *cov3D_precomp, const float *colors_precomp,
                                const float *view, const float *proj, const float *campos, int W, int H,
                                int *radii, float2 *means2D, float *depths, float *cov3Ds, float *rgb,
                                float4 *conic_opacity, uint32_t *tiles_touched, uint32_t *sort_key,
                                uint32_t *key_min_blk, uint2 *rect,
                                float4 *blend_rec, int prefiltered, int V, const ViewBatch &vb, const StaticRef &st, int lean,
                                float *zero3, const CohRef &coh, int fast_sh) {
    const int gx = tiles_x(W), gy = tiles_y(H);
    // view batches with SH colours: every Gaussian's coefficients read once for all views (the SH pipe is 3-channel, and
    // its means3D / campos are the same arrays the preprocess reads)
    const int sh_pre = (C == 3 && V > 1 && colors_precomp == nullptr && shs != nullptr && campos != nullptr && M <= 16) ? 1 : 0;
    if (sh_pre) {
        // The fast arithmetic (stated tolerance) evaluates the colours on the matrix cores (sh_mfma.h: 16 Gaussians per
        // wave instruction, 102 -> 79 us for the per-splat stage of config 3's Gaussians x 5 views, colours equal to 2.4e-7);
        // the exact arithmetic keeps the scalar kernel, whose expression order is the reference's.  FNX_LAB_SH_MFMA=0 / 1
        // in the environment pins either (bench.py times one against the other).
        const char *lab = getenv("FNX_LAB_SH_MFMA");
        const bool mfma = lab ? lab[0] == '1' : fast_sh != 0;
        if (mfma)
            hipLaunchKernelGGL(sh_colors_views_mfma_kernel, dim3((P + 63) / 64), dim3(256), 0, s, P, D, M, V, means3D, campos,
                               shs, clamped, rgb, vb.geom);
        else
            hipLaunchKernelGGL(sh_colors_views_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, D, M, V, means3D, campos, shs,
                               clamped, rgb, vb.geom);
    }
    const int blocks = (P + 255) / 256 + (st.base ? (st.P + 255) / 256 : 0);  // + copy of the static splats' radii
    hipLaunchKernelGGL((preprocess_kernel<C>), dim3(blocks, V), dim3(256), 0, s, P, D, M, means3D, scales,
                       scale_modifier, rotations, opacities, shs, clamped, cov3D_precomp, colors_precomp, view, proj,
                       campos, W, H, radii, means2D, depths, cov3Ds, rgb, conic_opacity, gx, gy, tiles_touched,
                       sort_key, key_min_blk, rect, blend_rec, prefiltered, st, vb, lean, zero3, coh, sh_pre);
}

void launch_preprocess(int C, hipStream_t s, int P, int D, int M, const float *means3D, const float *scales,
                       float scale_modifier, const float *rotations, const float *opacities, const float *shs,
                       uint8_t *clamped, const float *cov3D_precomp, const float *colors_precomp, const float *view,
                       const float *proj, const float *campos, int W, int H, int *radii,
                       float2 *means2D, float *depths, float *cov3Ds, float *rgb, float4 *conic_opacity,
                       uint32_t *tiles_touched, uint32_t *sort_key, uint32_t *key_min_blk, uint2 *rect,
                       float4 *blend_rec, int prefiltered, int V, const ViewBatch &vb, const StaticRef &st, int lean,
                       float *zero3, const CohRef &coh, int fast_sh) {
    if (C == 3)
        launch_preprocess_c<3>(s, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, clamped,
                               cov3D_precomp, colors_precomp, view, proj, campos, W, H, radii,
                               means2D, depths, cov3Ds, rgb, conic_opacity, tiles_touched, sort_key, key_min_blk, rect,
                               blend_rec, prefiltered, V, vb, st, lean, zero3, coh, fast_sh);
    else
        launch_preprocess_c<1>(s, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, clamped,
                               cov3D_precomp, colors_precomp, view, proj, campos, W, H, radii,
                               means2D, depths, cov3Ds, rgb, conic_opacity, tiles_touched, sort_key, key_min_blk, rect,
                               blend_rec, prefiltered, V, vb, st, lean, zero3, coh, fast_sh);
}

void launch_tile_scan(hipStream_t s, int T, const uint32_t *tile_count, uint32_t *ranges, uint32_t *dyn_start,
                      uint32_t *header, int P, int H, uint32_t *sort_scratch_words, uint32_t *depth_hint,
                      uint32_t deep_min, uint32_t *tile_order, uint8_t *tile_deep, int V, const ViewBatch &vb,
                      const StaticRef &st, const SegRef &sg) {
    const SortScratch L = sort_scratch(P);
    // FNX_DEEP_ORDER_MIN (developer switch): tiles at least this deep are ORDERED deepest first; the raised wave priority stays
    // with those at least deep_min deep
    static const uint32_t order_min = [] { const char *e = getenv("FNX_DEEP_ORDER_MIN"); return e ? (uint32_t)atoi(e) : 0u; }();
    const uint32_t prio_min = deep_min;
    if (order_min && order_min < deep_min) deep_min = order_min;
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1, V), dim3(1024), 0, s, T, tile_count, ranges, dyn_start, header, vb.img,
                       splat_blocks(P), tiles_y(H), sort_scratch_words + L.blk_total, sort_scratch_words + L.emit_ctl,
                       sort_scratch_words + L.emit_items, vb.geom, depth_hint, deep_min, prio_min, tile_order, tile_deep, st,
                       sort_scratch_words + L.ctl, sg);
}

void launch_blend_forward(int C, hipStream_t s, int W, int H, const uint32_t *ranges, uint32_t *point_list,
                          const float4 *blend_rec, const float *bg, float *final_T, uint32_t *n_contrib,
                          float *out_color, float *out_depth, uint32_t *header, uint32_t capacity,
                          uint32_t *status_out, const uint32_t *tile_count, const uint32_t *dyn_start,
                          float *acc_final, const uint32_t *tile_order, const uint8_t *tile_deep, uint32_t *depth_hint,
                          const StaticRef &st, int materialize_all, int V, const ViewBatch &vb, int fast, int deep,
                          uint32_t dyn_limit, const InvUpdate &iu, const DualRef &du, const SegRef &sg) {
    const int gx = tiles_x(W), T = gx * tiles_y(H);
    // static splats never take gradients: in static-split mode the limit is at most the first static id
    if (st.base && dyn_limit > st.id0) dyn_limit = st.id0;
    // Deep tiles (depth hints of the previous forward) go to the super-batch kernel; it exists for the fast arithmetic.
    // A deep workgroup holds a whole compute unit at modest utilisation to cut the tile's LATENCY, which pays when the
    // launch is bound by its longest walks -- few views per launch (a rank's share of a sharded batch) -- and costs
    // throughput when thousands of other tiles wait for those compute units: deep = 1 (auto) uses it up to two views.
    const int use_deep = (fast && depth_hint && !du.img1 && (deep == 2 || (deep == 1 && V <= 2))) ? 1 : 0;
    // deep = 3: the deep tiles go to the staging-wave kernel (raster_forward_ws.h; both arithmetics, bit-identical to the
    // per-tile kernel) on the helper stream; deep = 4: every tile does, instead of the per-tile kernel
    const bool ws_ok = !(sg.base && fast && !du.img1 && !materialize_all && depth_hint);
    // deep = 5 (the default): by the number of views in the launch -- one or two views are bound by their deepest tiles'
    // chains (a one-view forward of config 3: 233 us, of which the five-view launch adds only 70) and take the staging waves
    // for every tile (233 -> 166 us), three views for the deep tiles only, more views are bound by the compute units'
    // instruction throughput and keep the per-tile kernel (measured)
    if (deep == 5) deep = !ws_ok ? 0 : V <= 2 ? 4 : (V == 3 && depth_hint) ? 3 : 0;
    const int use_ws = (ws_ok && deep == 3 && depth_hint) ? 1 : (ws_ok && deep == 4) ? 2 : 0;
    // deep tiles per view the staging-wave launch takes (the deepest ones: they lead the order; the rest stay with the
    // per-tile kernel).  FNX_WS_MAX (developer switch, multiple of 8) overrides it.
    static const int kWsDeepMax = [] {
        const char *e = getenv("FNX_WS_MAX");
        const int v = e ? atoi(e) : 64;  // (3 views: 1 611 it/s at 512, 1 641 at 64; profiles/r06_lab_staging_waves.md)
        return v < 8 ? 8 : (v + 7) & ~7;
    }();
    // the two kernels touch disjoint tiles: the deep one runs on a helper stream beside the per-tile kernel
    // (one helper per caller stream and device: two renders on two streams -- the 3-channel and the 1-channel one of a
    // dual-channel iteration -- may be in flight, or being captured into two branches of a graph, at the same time)
    struct Helper {
        hipStream_t of;
        int device;
        hipStream_t stream;
        hipEvent_t ev_fork, ev_join;
    };
    static Helper helpers[32];
    static int n_helpers = 0;
    static std::mutex helpers_mu;
    hipStream_t helper = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t sd = s, s_main = s;
    if (use_deep || use_ws == 1) {
        const int n_cu = device_cu_count();
        {
            int dev = 0;
            (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> lock(helpers_mu);
            int at = -1;
            for (int i = 0; i < n_helpers; i++)
                if (helpers[i].of == s && helpers[i].device == dev) at = i;
            if (at < 0 && n_helpers < 32) {
                Helper h{s, dev, nullptr, nullptr, nullptr};
                if (hipStreamCreateWithFlags(&h.stream, hipStreamNonBlocking) == hipSuccess &&
                    hipEventCreateWithFlags(&h.ev_fork, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&h.ev_join, hipEventDisableTiming) == hipSuccess) {
                    at = n_helpers;
                    helpers[n_helpers++] = h;
                }
            }
            if (at >= 0) {
                helper = helpers[at].stream;
                ev_fork = helpers[at].ev_fork;
                ev_join = helpers[at].ev_join;
            }
        }
        if (helper && hipEventRecord(ev_fork, s) == hipSuccess && hipStreamWaitEvent(helper, ev_fork, 0) == hipSuccess)
            sd = helper;
        (void)hipGetLastError();
#define FNX_LAUNCH_BD(CC, SS)                                                                                          \
    hipLaunchKernelGGL((blend_forward_deep_kernel<CC, SS>), dim3(n_cu), dim3(256 * FNX_DEEP_GROUPS), 0, sd, T, gx,     \
                       ranges, point_list, W, H, blend_rec, bg, final_T, n_contrib, out_color, out_depth, header,      \
                       capacity, tile_count, dyn_start, acc_final, tile_order, depth_hint, st, materialize_all, vb, V)
#define FNX_LAUNCH_WS(CC, SS, FF, GX, DEEP_ONLY, ST) FNX_LAUNCH_WS_(CC, SS, FF, false, GX, DEEP_ONLY, ST)
#define FNX_LAUNCH_WS_(CC, SS, FF, DD, GX, DEEP_ONLY, ST)                                                              \
    hipLaunchKernelGGL((blend_forward_ws_kernel<CC, SS, FF, DD>), dim3(GX, V), dim3(512), 0, ST, T, gx, ranges,        \
                       point_list, W, H, blend_rec, bg, final_T, n_contrib, out_color, out_depth, header, capacity,    \
                       status_out, tile_count, dyn_start, acc_final, tile_order, tile_deep, depth_hint, st,            \
                       materialize_all, vb, DEEP_ONLY, dyn_limit, iu, du)
#define FNX_LAUNCH_WS_ALL(GX, DEEP_ONLY, ST)                                                                            \
    do {                                                                                                               \
        if (du.img1) { if (fast) FNX_LAUNCH_WS_(3, true, true, true, GX, DEEP_ONLY, ST); else FNX_LAUNCH_WS_(3, true, false, true, GX, DEEP_ONLY, ST); } \
        else if (C == 3 && st.base) { if (fast) FNX_LAUNCH_WS(3, true, true, GX, DEEP_ONLY, ST); else FNX_LAUNCH_WS(3, true, false, GX, DEEP_ONLY, ST); } \
        else if (C == 3) { if (fast) FNX_LAUNCH_WS(3, false, true, GX, DEEP_ONLY, ST); else FNX_LAUNCH_WS(3, false, false, GX, DEEP_ONLY, ST); }     \
        else if (st.base) { if (fast) FNX_LAUNCH_WS(1, true, true, GX, DEEP_ONLY, ST); else FNX_LAUNCH_WS(1, true, false, GX, DEEP_ONLY, ST); }      \
        else { if (fast) FNX_LAUNCH_WS(1, false, true, GX, DEEP_ONLY, ST); else FNX_LAUNCH_WS(1, false, false, GX, DEEP_ONLY, ST); }                 \
    } while (0)
        // The staging-wave kernel goes on the CALLER's stream and the per-tile kernel on the helper: a captured graph keeps
        // the first-enqueued successor of a node on the node's hardware queue, and the deep tiles' chains must START first --
        // 512-thread workgroups enqueued beside a per-tile launch that already fills the chip wait for two of its
        // workgroups to retire on the same compute unit (measured: the deep tiles then start ~100 us late).
        // FNX_WS_SWAP=0 (developer switch): the other way round.
        static const bool ws_swap = [] { const char *e = getenv("FNX_WS_SWAP"); return !e || atoi(e) != 0; }();
        if (use_ws == 1) {
            FNX_LAUNCH_WS_ALL(std::min((T + 7) & ~7, kWsDeepMax), 1, (ws_swap ? s : sd));
            if (ws_swap) s_main = sd;
        } else
        if (C == 3 && st.base) FNX_LAUNCH_BD(3, true);
        else if (C == 3) FNX_LAUNCH_BD(3, false);
        else if (st.base) FNX_LAUNCH_BD(1, true);
        else FNX_LAUNCH_BD(1, false);
#undef FNX_LAUNCH_BD
    }
    if (use_ws == 2) {
        FNX_LAUNCH_WS_ALL((T + 7) & ~7, 0, s);
        return;
    }
    const int skip_deep = use_ws == 1 ? kWsDeepMax : use_deep ? 0x7FFFFFFF : 0;
#define FNX_LAUNCH_BF(CC, SS)                                                                                          \
    if (fast)                                                                                                          \
        FNX_LAUNCH_BF_(CC, SS, true);                                                                                  \
    else                                                                                                               \
        FNX_LAUNCH_BF_(CC, SS, false)
#define FNX_LAUNCH_BF_(CC, SS, FF) FNX_LAUNCH_BF__(CC, SS, FF, false)
#define FNX_LAUNCH_BF__(CC, SS, FF, DD)                                                                                \
    hipLaunchKernelGGL((blend_forward_kernel<CC, SS, FF, DD>), dim3((T + 7) & ~7, V), dim3(256), 0, s_main, T, gx, ranges,  \
                       point_list, W, H,                                                                               \
                       blend_rec, bg, final_T, n_contrib, out_color, out_depth, header, capacity, status_out,          \
                       tile_count, dyn_start, acc_final, tile_order, tile_deep, depth_hint, st, materialize_all, vb,   \
                       skip_deep, dyn_limit, iu, du, sg)
    // segmented deep tiles (the work list tile_scan_kernel left in the segment scratch): fast arithmetic, one image
    const bool use_seg = sg.base && fast && !du.img1 && !materialize_all && !use_deep && !use_ws && depth_hint;
#define FNX_LAUNCH_SEG(CC, SS)                                                                                         \
    hipLaunchKernelGGL((blend_forward_kernel<CC, SS, true, false, true>), dim3((T + (int)kSegMax + 7) & ~7, V), dim3(256), \
                       0, s, T, gx, ranges, point_list, W, H, blend_rec, bg, final_T, n_contrib, out_color, out_depth,  \
                       header, capacity, status_out, tile_count, dyn_start, acc_final, tile_order, tile_deep,           \
                       depth_hint, st, materialize_all, vb, skip_deep, dyn_limit, iu, du, sg)
    if (use_seg) {
        if (C == 3 && st.base) { FNX_LAUNCH_SEG(3, true); }
        else if (C == 3) { FNX_LAUNCH_SEG(3, false); }
        else if (st.base) { FNX_LAUNCH_SEG(1, true); }
        else { FNX_LAUNCH_SEG(1, false); }
    } else
    if (du.img1) {  // dual mode: 3 channels + the single-channel image of the per-call splats, static-split lists
        if (fast) { FNX_LAUNCH_BF__(3, true, true, true); }
        else { FNX_LAUNCH_BF__(3, true, false, true); }
    } else
    if (C == 3 && st.base) { FNX_LAUNCH_BF(3, true); }
    else if (C == 3) { FNX_LAUNCH_BF(3, false); }
    else if (st.base) { FNX_LAUNCH_BF(1, true); }
    else { FNX_LAUNCH_BF(1, false); }
#undef FNX_LAUNCH_BF
#undef FNX_LAUNCH_BF_
#undef FNX_LAUNCH_BF__
#undef FNX_LAUNCH_SEG
    if ((use_deep || use_ws == 1) && sd != s) {  // join
        (void)hipEventRecord(ev_join, sd);
        (void)hipStreamWaitEvent(s, ev_join, 0);
    }
}

void launch_mark_visible(hipStream_t s, int P, const float *means3D, const float *view, uint8_t *present) {
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
}

}  // namespace fnx
