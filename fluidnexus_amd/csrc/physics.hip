// Fused particle-physics kernels for gfx950 (C ABI: include/fnx_physics.h).
//
// Neighbour search: uniform hash grid with cell edge H.  Points are bucketed by a hash of their
// integer cell (count -> scan -> fill, LDS-free: the clouds are 10^4..10^5 points and these
// kernels are latency-, not bandwidth-bound); a query walks the buckets of the 27 surrounding cells and accepts
// every candidate that passes the distance test.  The low 9 bits of a bucket id are the cell coordinates mod 8
// (cell_hash), so those 27 cells land in 27 different buckets: a point within H of the query is met exactly
// once, and points of far cells that share a bucket fail the distance test -- hash collisions are harmless
// without a cell comparison.  Bucket records are float4 (x, y, z, id) so a query streams 16 B per candidate,
// coalesced within a bucket.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#include "../../include/fnx_physics.h"
#include "../../include/fnx_raster.h"

namespace {

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

struct GridView {
    uint32_t *header;  // [0] M (buckets), [1] N
    uint32_t *count;   // [M]
    uint32_t *start;   // [M + 1]
    uint32_t *cursor;  // [M]
    float4 *rec;       // [N] (x, y, z, id bits)
    float4 *aux0;      // [N] per-slot payload (kernel-specific, rewritten by each call)
    float4 *aux1;      // [N]
    uint32_t M;
};

inline uint32_t buckets_for(int N) {
    uint32_t m = 4096;
    while (m < (uint32_t)(N > 0 ? N : 1)) m <<= 1;
    return m;
}

inline size_t grid_bytes(int N) {
    const size_t M = buckets_for(N);
    size_t off = align_up(64);
    off = align_up(off + M * 4);
    off = align_up(off + (M + 1) * 4);
    off = align_up(off + M * 4);
    off = align_up(off + (size_t)(N > 0 ? N : 0) * 16);
    off = align_up(off + (size_t)(N > 0 ? N : 0) * 16);
    off = align_up(off + (size_t)(N > 0 ? N : 0) * 16);
    return off + kAlign;
}

// M_override: a smaller table than buckets_for(N) inside a blob of the standard size (fnx_distance_loss: more points
// per bucket, a shorter scan)
inline GridView carve(char *blob, int N, uint32_t M_override = 0) {
    char *b = (char *)(((uintptr_t)blob + kAlign - 1) / kAlign * kAlign);
    GridView g;
    g.M = M_override ? M_override : buckets_for(N);
    size_t off = 0;
    g.header = (uint32_t *)(b + off); off = align_up(off + 64);
    g.count = (uint32_t *)(b + off);  off = align_up(off + (size_t)g.M * 4);
    g.start = (uint32_t *)(b + off);  off = align_up(off + ((size_t)g.M + 1) * 4);
    g.cursor = (uint32_t *)(b + off); off = align_up(off + (size_t)g.M * 4);
    g.rec = (float4 *)(b + off);  off = align_up(off + (size_t)(N > 0 ? N : 0) * 16);
    g.aux0 = (float4 *)(b + off); off = align_up(off + (size_t)(N > 0 ? N : 0) * 16);
    g.aux1 = (float4 *)(b + off);
    return g;
}

__device__ __forceinline__ int3 cell_of(float x, float y, float z, float inv_cell) {
    return make_int3((int)floorf(x * inv_cell), (int)floorf(y * inv_cell), (int)floorf(z * inv_cell));
}
// Bucket of an integer cell.  The low 9 bits are (cx, cy, cz) mod 8, so the 27 cells around any
// query land in 27 DIFFERENT buckets: a point within H of the query lies in exactly one of those
// cells, hence is met exactly once; points of far cells that share a bucket fail the distance test.
__device__ __forceinline__ uint32_t cell_hash(int3 c, uint32_t mask) {
    const uint32_t lo = ((uint32_t)c.x & 7u) | (((uint32_t)c.y & 7u) << 3) | (((uint32_t)c.z & 7u) << 6);
    const uint32_t hi = ((uint32_t)(c.x >> 3) * 73856093u) ^ ((uint32_t)(c.y >> 3) * 19349663u) ^
                        ((uint32_t)(c.z >> 3) * 83492791u);
    return (lo | (hi << 9)) & mask;
}

// zero-fill as a kernel: hipMemsetAsync nodes inside a captured hipGraph were found to make replays fault after an
// unrelated device-to-host copy on this ROCm, and a kernel costs the same launch
__global__ void __launch_bounds__(256)
zero_u32_kernel(uint32_t *__restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}

__global__ void __launch_bounds__(256)
grid_count_kernel(const float *__restrict__ xyz, int N, float inv_cell, uint32_t mask, uint32_t *__restrict__ count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int3 c = cell_of(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], inv_cell);
    atomicAdd(&count[cell_hash(c, mask)], 1u);
}

// Exclusive scan of the bucket counts by one 1024-thread workgroup.  The kernel is pure latency (it sits on the
// critical path of every iteration, before the first neighbour search): up to eight 4096-bucket chunks (coalesced
// uint4 per thread) are loaded at once and scanned side by side, so the workgroup synchronises twice per 32768
// buckets instead of twice per 4096, and the 16 wave totals are scanned with shuffles.
constexpr int kScanChunks = 8;
__device__ __forceinline__ void grid_scan_block(uint32_t M, const uint32_t *__restrict__ count,
                                                uint32_t *__restrict__ start, uint32_t *__restrict__ cursor,
                                                uint32_t (*s_wave)[16]) {
    constexpr int K = kScanChunks;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < M; base += K * 4096u) {  // M is a power of two >= 4096
        uint4 c[K];
        uint32_t sum[K], inc[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t at = base + (uint32_t)k * 4096u + tid * 4u;
            c[k] = at < M ? *reinterpret_cast<const uint4 *>(count + at) : make_uint4(0, 0, 0, 0);
            sum[k] = c[k].x + c[k].y + c[k].z + c[k].w;
            inc[k] = sum[k];
        }
        for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint32_t v = (uint32_t)__shfl_up((int)inc[k], off);
                if (lane >= (uint32_t)off) inc[k] += v;
            }
        }
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < K; k++) s_wave[k][w] = inc[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) {
            // every wave scans the 16 wave totals of chunk k itself (lanes 0..15)
            uint32_t winc = lane < 16 ? s_wave[k][lane] : 0u;
            for (int off = 1; off < 16; off <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)winc, off);
                if (lane >= (uint32_t)off) winc += v;
            }
            const uint32_t before = w ? (uint32_t)__shfl((int)winc, (int)w - 1) : 0u;
            const uint32_t total = (uint32_t)__shfl((int)winc, 15);
            const uint32_t at = base + (uint32_t)k * 4096u + tid * 4u;
            if (at < M) {
                uint4 o;
                o.x = carry + before + inc[k] - sum[k];
                o.y = o.x + c[k].x;
                o.z = o.y + c[k].y;
                o.w = o.z + c[k].z;
                *reinterpret_cast<uint4 *>(start + at) = o;
                *reinterpret_cast<uint4 *>(cursor + at) = make_uint4(0, 0, 0, 0);
            }
            carry += total;
        }
        __syncthreads();  // s_wave is rewritten by the next round
    }
    if (tid == 0) start[M] = carry;
}

__global__ void __launch_bounds__(1024)
grid_scan_kernel(uint32_t M, const uint32_t *__restrict__ count, uint32_t *__restrict__ start,
                 uint32_t *__restrict__ cursor) {
    __shared__ uint32_t s_wave[kScanChunks][16];
    // One workgroup, pure latency (two barriers per 32 768 buckets): alone it takes 5-7 us, but as a side-branch kernel under
    // the rasteriser's blend forward its 16 waves waited behind four throughput-bound waves per SIMD and took 65 us (config 3)
    // / 212 us (config 5) -- raised wave priority instead of more workgroups: a multi-workgroup scan of 8 chunks would add a
    // look-back chain to a kernel that is already nothing but latency, and the starvation is what costs (round 5).
    __builtin_amdgcn_s_setprio(3);
    grid_scan_block(M, count, start, cursor, s_wave);
}

__global__ void __launch_bounds__(256)
grid_fill_kernel(const float *__restrict__ xyz, int N, float inv_cell, uint32_t mask,
                 const uint32_t *__restrict__ start, uint32_t *__restrict__ cursor, float4 *__restrict__ rec) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const uint32_t h = cell_hash(cell_of(x, y, z, inv_cell), mask);
    const uint32_t slot = start[h] + atomicAdd(&cursor[h], 1u);
    rec[slot] = make_float4(x, y, z, __uint_as_float((uint32_t)i));
}

// Sum over the 64 lanes of a wave (DPP row operations); the total lands in lane 63.
__device__ __forceinline__ float wave_sum63(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
    return v;
}
// Sum over 8 consecutive lanes; totals land in lanes 7 and 15 of each 16-lane row.
__device__ __forceinline__ float oct_sum7(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xa, false));
    return v;
}
// Sum over a 16-lane row; the total lands in lane 15 of the row.
__device__ __forceinline__ float row_sum15(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false));
    return v;
}

// L lanes (16 = one DPP row, or 64 = one wave) share one query point p and stride over the
// candidates of the 27 surrounding buckets; f(slot, j, dx, dy, dz, r2) with d = p - x_j for every
// grid point with r2 < H2 (see cell_hash for why no candidate is met twice).
template <int L, typename F>
__device__ __forceinline__ void for_neighbours(int sub, float px, float py, float pz, float inv_cell, float H2,
                                               uint32_t mask, const uint32_t *__restrict__ start,
                                               const float4 *__restrict__ rec, F &&f) {
    const int3 c = cell_of(px, py, pz, inv_cell);
    // fractional position inside the own cell: a neighbouring cell whose closest point is farther than H
    // (0.1 % margin against the rounding of f) holds no neighbour
    const float fx = px * inv_cell - (float)c.x, fy = py * inv_cell - (float)c.y, fz = pz * inv_cell - (float)c.z;
    const float cell2 = (1.0f / inv_cell) * (1.0f / inv_cell);
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const float gx = dx < 0 ? fx : (dx > 0 ? 1.0f - fx : 0.0f), gy = dy < 0 ? fy : (dy > 0 ? 1.0f - fy : 0.0f),
                            gz = dz < 0 ? fz : (dz > 0 ? 1.0f - fz : 0.0f);
                if ((gx * gx + gy * gy + gz * gz) * cell2 > H2 * 1.001f) continue;
                const int3 cc = make_int3(c.x + dx, c.y + dy, c.z + dz);
                const uint32_t h = cell_hash(cc, mask);
                const uint32_t s0 = start[h], s1 = start[h + 1];
                for (uint32_t s = s0 + sub; s < s1; s += L) {
                    const float4 q = rec[s];
                    const float ex = px - q.x, ey = py - q.y, ez = pz - q.z;
                    const float r2 = ex * ex + ey * ey + ez * ez;
                    if (r2 < H2) f(s, __float_as_uint(q.w), ex, ey, ez, r2);
                }
            }
}

// gm_dynamics.py:1269-1294: p_i = sum_j poly6(r2_ij) / imass_i; p_ratio = p_i / p0.  16 lanes per particle.
__global__ void __launch_bounds__(256)
density_forward_kernel(const float *__restrict__ xyz, int N, const float *__restrict__ imass, float inv_cell, float H2,
                       float term1, float p0, uint32_t mask, const uint32_t *__restrict__ start,
                       const float4 *__restrict__ rec, float *__restrict__ p_ratio) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const int ii = min(i, N - 1);
    float acc = 0.f;
    for_neighbours<16>(sub, xyz[3 * ii], xyz[3 * ii + 1], xyz[3 * ii + 2], inv_cell, H2, mask, start, rec,
                       [&](uint32_t, uint32_t, float, float, float, float r2) {
                           const float t = H2 - r2;
                           acc += term1 * (t * t * t);
                       });
    acc = row_sum15(acc);
    if (sub == 15 && i < N) p_ratio[i] = acc / imass[i] / p0;
}

// Lanes per particle in the two kernels of the fused stage (a hidden bucket holds ~7 particles: with 16 lanes per
// particle more than half of the lanes idle at every bucket and the per-bucket overhead is paid twice as often)
#ifndef FNX_DENSITY_LANES
#define FNX_DENSITY_LANES 8
#endif
constexpr int kDL = FNX_DENSITY_LANES, kDPerWg = 256 / kDL;
__device__ __forceinline__ float density_lane_sum(float v) { return kDL == 16 ? row_sum15(v) : oct_sum7(v); }

__global__ void __launch_bounds__(256)
density_backward_kernel(const float *__restrict__ xyz, int N, const float *__restrict__ imass, float inv_cell, float H2,
                        float term1, float p0, uint32_t mask, const uint32_t *__restrict__ start,
                        const float4 *__restrict__ rec, const float *__restrict__ g, float gscale,
                        float *__restrict__ dL_dxyz) {
    const int i = blockIdx.x * kDPerWg + (threadIdx.x / kDL), sub = threadIdx.x & (kDL - 1);
    const int ii = min(i, N - 1);
    const float Gi = (g[ii] * gscale) / imass[ii] / p0;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for_neighbours<kDL>(sub, xyz[3 * ii], xyz[3 * ii + 1], xyz[3 * ii + 2], inv_cell, H2, mask, start, rec,
                       [&](uint32_t, uint32_t j, float ex, float ey, float ez, float r2) {
                           const float Gj = (g[j] * gscale) / imass[j] / p0;
                           const float t = H2 - r2;
                           const float dW = -3.0f * term1 * (t * t);  // d poly6 / d r2
                           const float k = (Gi + Gj) * dW * 2.0f;
                           ax += k * ex;
                           ay += k * ey;
                           az += k * ez;
                       });
    ax = density_lane_sum(ax);
    ay = density_lane_sum(ay);
    az = density_lane_sum(az);
    if (sub == kDL - 1 && i < N) {
        dL_dxyz[3 * i + 0] = ax;
        dL_dxyz[3 * i + 1] = ay;
        dL_dxyz[3 * i + 2] = az;
    }
}

// ---- max_num_neighbors (K-cap) mode ---------------------------------------------------------------
// torch_cluster's CUDA radius kernel (the one the reference's searches run on, gm_dynamics.py:1276,1302,1463 with
// max_num_neighbors = KNN_K) walks the points of x in INDEX order for every query and stops after K hits: a query
// keeps the K SMALLEST INDICES among its neighbours within r.  cut[q] = that K-th smallest index (0xFFFFFFFF when
// the query has at most K neighbours), so "q keeps neighbour j" is the one comparison j <= cut[q].  L lanes per query:
// a counting pass, and only for a query over the cap a bisection on the index (31 counting passes at most).
template <int L>
__device__ __forceinline__ uint32_t group_sum_all(uint32_t v) {  // sum over the L consecutive lanes of a group, in all of them
    for (int off = 1; off < L; off <<= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
    return v;
}

constexpr int kCutLanes = 8;
__global__ void __launch_bounds__(256)
knn_cut_kernel(const float *__restrict__ queries, int Nq, float inv_cell, float H2, uint32_t mask,
               const uint32_t *__restrict__ start, const float4 *__restrict__ rec, uint32_t K,
               uint32_t *__restrict__ cut) {
    const int q = blockIdx.x * (256 / kCutLanes) + (threadIdx.x / kCutLanes), sub = threadIdx.x & (kCutLanes - 1);
    const int qq = min(q, Nq - 1);
    const float px = queries[3 * qq], py = queries[3 * qq + 1], pz = queries[3 * qq + 2];
    auto count_upto = [&](uint32_t bound) {
        uint32_t c = 0;
        for_neighbours<kCutLanes>(sub, px, py, pz, inv_cell, H2, mask, start, rec,
                                  [&](uint32_t, uint32_t j, float, float, float, float) { c += j <= bound ? 1u : 0u; });
        return group_sum_all<kCutLanes>(c);
    };
    uint32_t result = 0xFFFFFFFFu;
    if (count_upto(0xFFFFFFFFu) > K) {  // uniform over the group's lanes
        uint32_t lo = 0, hi = 0x7FFFFFFFu;  // the smallest index with count(<= index) >= K
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (count_upto(mid) >= K) hi = mid; else lo = mid + 1;
        }
        result = lo;
    }
    if (sub == 0 && q < Nq) cut[q] = result;
}

// gm_dynamics.py:1276-1290 with the cap: the edge (query q -> kept neighbour i) adds poly6 at the NEIGHBOUR's index
// (radius_graph, flow = source_to_target, swaps the two rows), so p_i = sum over q with i <= cut[q] of poly6(r2_iq).
__global__ void __launch_bounds__(256)
density_forward_kcap_kernel(const float *__restrict__ xyz, int N, const float *__restrict__ imass, float inv_cell,
                            float H2, float term1, float p0, uint32_t mask, const uint32_t *__restrict__ start,
                            const float4 *__restrict__ rec, const uint32_t *__restrict__ cut,
                            float *__restrict__ p_ratio) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const int ii = min(i, N - 1);
    float acc = 0.f;
    for_neighbours<16>(sub, xyz[3 * ii], xyz[3 * ii + 1], xyz[3 * ii + 2], inv_cell, H2, mask, start, rec,
                       [&](uint32_t, uint32_t q, float, float, float, float r2) {
                           if ((uint32_t)ii > cut[q]) return;
                           const float t = H2 - r2;
                           acc += term1 * (t * t * t);
                       });
    acc = row_sum15(acc);
    if (sub == 15 && i < N) p_ratio[i] = acc / imass[i] / p0;
}

// d/dx_i of sum_i g_i p_ratio_i with the edge set frozen: pairs (i, j) enter once for "j keeps i" (weight G_i) and
// once for "i keeps j" (weight G_j).
__global__ void __launch_bounds__(256)
density_backward_kcap_kernel(const float *__restrict__ xyz, int N, const float *__restrict__ imass, float inv_cell,
                             float H2, float term1, float p0, uint32_t mask, const uint32_t *__restrict__ start,
                             const float4 *__restrict__ rec, const uint32_t *__restrict__ cut,
                             const float *__restrict__ g, float *__restrict__ dL_dxyz) {
    const int i = blockIdx.x * kDPerWg + (threadIdx.x / kDL), sub = threadIdx.x & (kDL - 1);
    const int ii = min(i, N - 1);
    const float Gi = g[ii] / imass[ii] / p0;
    const uint32_t cut_i = cut[ii];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for_neighbours<kDL>(sub, xyz[3 * ii], xyz[3 * ii + 1], xyz[3 * ii + 2], inv_cell, H2, mask, start, rec,
                        [&](uint32_t, uint32_t j, float ex, float ey, float ez, float r2) {
                            const float Gj = g[j] / imass[j] / p0;
                            const float t = H2 - r2;
                            const float dW = -3.0f * term1 * (t * t);
                            const float k = (((uint32_t)ii <= cut[j] ? Gi : 0.f) + (j <= cut_i ? Gj : 0.f)) * dW * 2.0f;
                            ax += k * ex;
                            ay += k * ey;
                            az += k * ez;
                        });
    ax = density_lane_sum(ax);
    ay = density_lane_sum(ay);
    az = density_lane_sum(az);
    if (sub == kDL - 1 && i < N) {
        dL_dxyz[3 * i + 0] = ax;
        dL_dxyz[3 * i + 1] = ay;
        dL_dxyz[3 * i + 2] = az;
    }
}

// ---- physical-particle stage, fused (fnx_physical_stage) -------------------------------------------
// x = x_nn * sf and the one-tick advected guess x' (gm_dynamics.py:1014-1030, same operation order as
// the reference's tensor expression); also sum (x - x_est)^2 -> terms[0].
__global__ void __launch_bounds__(256)
stage_points_kernel(const float *__restrict__ x_nn, int N, float sf, const float *__restrict__ x_est,
                    const float *__restrict__ x_prev, const float *__restrict__ buoyancy,
                    const float *__restrict__ force, float bmax, float secs, float *__restrict__ x,
                    float *__restrict__ xg, float *__restrict__ terms) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float e2 = 0.f;
    if (i < N) {
        const float yn = x_nn[3 * i + 1];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float xs = x_nn[3 * i + c] * sf;
            float b = buoyancy[3 * i + c];
            if (bmax > 0.0f) b = b * (1.0f - (yn / bmax));
            const float tmp_v = (xs - x_prev[3 * i + c]) / secs;
            const float ev = tmp_v + b * secs + secs * force[3 * i + c];
            x[3 * i + c] = xs;
            xg[3 * i + c] = xs + secs * ev;
            const float d = xs - x_est[3 * i + c];
            e2 += d * d;
        }
    }
    // per-workgroup partial sums (deterministic; stage_combine_kernel adds them up)
    __shared__ float s_w[4];
    e2 = wave_sum63(e2);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = e2;
    __syncthreads();
    if (threadIdx.x == 0) terms[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// d_i = p_ratio_i - 1 and per-workgroup sums of d_i^2 -> term[blockIdx.x].  kDL lanes per particle.
// WATCH (fnx_knn_watch): the reference's searches keep at most KNN_K neighbours per query (torch_cluster radius_graph /
// radius with max_num_neighbors, gm_dynamics.py:1276,1302,1463); these kernels take every pair within H, which is the same
// thing while no list is longer than K.  With a watch armed every query also counts its neighbours and raises `bit` in
// the caller's flag word when the count exceeds K: the run no longer equals the reference's and must say so.
template <bool WATCH>
__global__ void __launch_bounds__(256)
density_term_kernel(const float *__restrict__ xyz, int N, const float *__restrict__ imass, float inv_cell, float H2,
                    float term1, float p0, uint32_t mask, const uint32_t *__restrict__ start,
                    const float4 *__restrict__ rec, float *__restrict__ d_out, float *__restrict__ term,
                    uint32_t *__restrict__ knn_flags, float knn_k, uint32_t bit) {
    const int i = blockIdx.x * kDPerWg + (threadIdx.x / kDL), sub = threadIdx.x & (kDL - 1);
    const int ii = min(i, N - 1);
    float acc = 0.f, cnt = 0.f;
    for_neighbours<kDL>(sub, xyz[3 * ii], xyz[3 * ii + 1], xyz[3 * ii + 2], inv_cell, H2, mask, start, rec,
                       [&](uint32_t, uint32_t, float, float, float, float r2) {
                           const float t = H2 - r2;
                           acc += term1 * (t * t * t);
                           if (WATCH) cnt += 1.0f;  // the list of radius_graph(loop = True): the particle itself included
                       });
    acc = density_lane_sum(acc);
    if (WATCH) {
        cnt = density_lane_sum(cnt);
        if (sub == kDL - 1 && i < N && cnt > knn_k) atomicOr(knn_flags, bit);
    }
    float d2 = 0.f;
    if (sub == kDL - 1 && i < N) {
        const float d = acc / imass[i] / p0 - 1.0f;
        d_out[i] = d;
        d2 = d * d;
    }
    __shared__ float s_w[4];
    d2 = wave_sum63(d2);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = d2;
    __syncthreads();
    if (threadIdx.x == 0) term[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// grad = d loss / d x_nn (physics.py _PhysicalStageLoss.backward, same association) and the weighted loss.
__global__ void __launch_bounds__(256)
stage_combine_kernel(int N, float sf, const float *__restrict__ x, const float *__restrict__ x_est,
                     const float *__restrict__ dx_est, const float *__restrict__ dg, const float *__restrict__ buoyancy,
                     float bmax, float secs, float lam_e, float lam_g, float lam_n, const float *__restrict__ partials,
                     float *__restrict__ terms, float *__restrict__ loss, float *__restrict__ grad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0) {  // add up the per-workgroup partial sums (fixed order), weighted loss
        __shared__ float s_r[256];
        const int nb_e = (N + 255) / 256, nb_d = (N + kDPerWg - 1) / kDPerWg;
        const float *part[3] = {partials, partials + nb_e, partials + nb_e + nb_d};
        const int cnt[3] = {nb_e, (lam_g > 0.f) ? nb_d : 0, (lam_n > 0.f) ? nb_d : 0};
        float tot[3];
        for (int k = 0; k < 3; k++) {
            float a = 0.f;
            for (int j = threadIdx.x; j < cnt[k]; j += 256) a += part[k][j];
            s_r[threadIdx.x] = a;
            __syncthreads();
            for (int off = 128; off >= 1; off >>= 1) {
                if ((int)threadIdx.x < off) s_r[threadIdx.x] += s_r[threadIdx.x + off];
                __syncthreads();
            }
            tot[k] = s_r[0];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            terms[0] = tot[0];
            terms[1] = tot[1];
            terms[2] = tot[2];
            float L = 0.f;
            if (lam_e > 0.f) L = L + lam_e * (tot[0] / (float)(3 * (size_t)N));
            if (lam_g > 0.f) L = L + lam_g * (tot[1] / (float)N);
            if (lam_n > 0.f) L = L + lam_n * (tot[2] / (float)N);
            loss[0] = L;
        }
    }
    if (i >= N) return;
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float gx = 0.f;
        if (lam_e > 0.f) gx += (x[3 * i + c] - x_est[3 * i + c]) * (2.0f * lam_e / (float)(3 * (size_t)N));
        if (lam_g > 0.f) gx += dx_est[3 * i + c];
        o[c] = gx * sf;
        if (lam_n > 0.f) o[c] = o[c] + dg[3 * i + c] * (2.0f * sf);
    }
    if (lam_n > 0.f && bmax > 0.0f) {
        const float dot = (dg[3 * i] * buoyancy[3 * i] + dg[3 * i + 1] * buoyancy[3 * i + 1]) + dg[3 * i + 2] * buoyancy[3 * i + 2];
        o[1] += dot * (-(secs * secs) / bmax);
    }
    grad[3 * i] = o[0];
    grad[3 * i + 1] = o[1];
    grad[3 * i + 2] = o[2];
}

// ---- PBF predictor / solver of the per-frame step ---------------------------------------
// guess_hidden_particles (gm_dynamics.py:978-1012, no wind): new buoyancy = gravity * alpha for every particle,
// optionally attenuated with height for the velocity update and decayed for storage.
__global__ void __launch_bounds__(256)
pbf_predict_kernel(const float *__restrict__ xyz, float *__restrict__ velocity, float *__restrict__ buoyancy,
                   float *__restrict__ force, float *__restrict__ estimate, float *__restrict__ counts, int N, float gx,
                   float gy, float gz, float alpha, float secs, float scale_max_y, float decay) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float b[3] = {1.0f * (gx * alpha), 1.0f * (gy * alpha), 1.0f * (gz * alpha)};
    float coeff = 1.0f;
    if (scale_max_y > 0.0f) coeff = 1.0f - (xyz[3 * i + 1] / scale_max_y);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float cur = (scale_max_y > 0.0f) ? b[c] * coeff : b[c];
        const float v = velocity[3 * i + c] + (cur * secs + secs * force[3 * i + c]);
        velocity[3 * i + c] = v;
        buoyancy[3 * i + c] = (decay > 0.0f) ? b[c] * decay : b[c];
        force[3 * i + c] = 0.0f;
        estimate[3 * i + c] = xyz[3 * i + c] + secs * v;
    }
    counts[i] = 0.0f;
}

// number of other particles within H (remove_invalid_particles :1040-1045)
__global__ void __launch_bounds__(256)
pbf_count_kernel(const float *__restrict__ xyz, int N, float inv_cell, float H2, uint32_t mask,
                 const uint32_t *__restrict__ start, const float4 *__restrict__ rec, int *__restrict__ out) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const int ii = min(i, N - 1);
    float n = 0.f;
    for_neighbours<16>(sub, xyz[3 * ii], xyz[3 * ii + 1], xyz[3 * ii + 2], inv_cell, H2, mask, start, rec,
                       [&](uint32_t, uint32_t j, float, float, float, float) { n += (j != (uint32_t)ii) ? 1.f : 0.f; });
    n = row_sum15(n);
    if (sub == 15 && i < N) out[i] = (int)n;
}

// spiky kernel gradient (:193-199) of the offset e = x_i - x_j with r2 = |e|^2
__device__ __forceinline__ bool spiky_grad(float ex, float ey, float ez, float r2, float H, float eps, float term1,
                                           float &gx, float &gy, float &gz) {
    const float rlen = sqrtf(r2 + eps);
    if (!(rlen < H) || !(rlen > 0.f)) return false;
    const float inv = 1.0f / (rlen + eps);
    const float t = H - rlen;
    const float s = term1 * (t * t);
    gx = -(ex * inv) * s;
    gy = -(ey * inv) * s;
    gz = -(ez * inv) * s;
    return true;
}

// project_gas_constraints, node pass (:1086-1133): density ratio, lambda, neighbour count, force correction
__global__ void __launch_bounds__(256)
pbf_lambda_kernel(const float *__restrict__ x, const float *__restrict__ velocity, float *__restrict__ force,
                  const float *__restrict__ imass, int N, float inv_cell, float H, float H2, float poly6_t1,
                  float spiky_t1, float p0, float kk, float relax, float eps, uint32_t mask,
                  const uint32_t *__restrict__ start, const float4 *__restrict__ rec, float *__restrict__ lambdas,
                  float *__restrict__ nlen) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const int ii = min(i, N - 1);
    float pi = 0.f, cnt = 0.f, grx = 0.f, gry = 0.f, grz = 0.f, gd = 0.f;
    for_neighbours<16>(sub, x[3 * ii], x[3 * ii + 1], x[3 * ii + 2], inv_cell, H2, mask, start, rec,
                       [&](uint32_t, uint32_t j, float ex, float ey, float ez, float r2) {
                           const float t = H2 - r2;
                           pi += poly6_t1 * (t * t * t);
                           cnt += 1.f;
                           float gx, gy, gz;
                           if (j != (uint32_t)ii && spiky_grad(ex, ey, ez, r2, H, eps, spiky_t1, gx, gy, gz)) {
                               grx += gx;
                               gry += gy;
                               grz += gz;
                               const float a = gx / p0, b = gy / p0, c = gz / p0;
                               gd += a * a + b * b + c * c;
                           }
                       });
    pi = row_sum15(pi);
    cnt = row_sum15(cnt);
    grx = row_sum15(grx);
    gry = row_sum15(gry);
    grz = row_sum15(grz);
    gd = row_sum15(gd);
    if (sub == 15 && i < N) {
        const float p_ratio = pi / imass[i] / p0;
        const float a = grx / p0, b = gry / p0, c = grz / p0;
        const float denom = gd + (a * a + b * b + c * c);
        lambdas[i] = -(p_ratio - 1.0f) / (denom + relax);
        nlen[i] = cnt;
        const float f = (1.0f - p_ratio) * -kk;
#pragma unroll
        for (int d = 0; d < 3; d++) force[3 * i + d] += velocity[3 * i + d] * f;
    }
}

// project_gas_constraints, position pass (:1135-1160): Jacobi update from the OLD positions into x_new
__global__ void __launch_bounds__(256)
pbf_delta_kernel(const float *__restrict__ x, int N, float inv_cell, float H, float H2, float poly6_t1, float spiky_t1,
                 float p0, float K_P, float E_P, float corr_denom, float eps, uint32_t mask,
                 const uint32_t *__restrict__ start, const float4 *__restrict__ rec, const float *__restrict__ lambdas,
                 const float *__restrict__ nlen, const float *__restrict__ counts, float *__restrict__ x_new) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const int ii = min(i, N - 1);
    const float li = lambdas[ii];
    float dx = 0.f, dy = 0.f, dz = 0.f;
    for_neighbours<16>(sub, x[3 * ii], x[3 * ii + 1], x[3 * ii + 2], inv_cell, H2, mask, start, rec,
                       [&](uint32_t, uint32_t j, float ex, float ey, float ez, float r2) {
                           float gx, gy, gz;
                           if (j != (uint32_t)ii && spiky_grad(ex, ey, ez, r2, H, eps, spiky_t1, gx, gy, gz)) {
                               const float t = H2 - r2;
                               const float p6 = poly6_t1 * (t * t * t);
                               const float q = p6 / corr_denom;
                               // the reference's exponent is 4 (gm_dynamics.py:110): two squarings instead of powf
                               const float corr = -K_P * ((E_P == 4.0f) ? (q * q) * (q * q) : powf(q, E_P));
                               const float w = (li + lambdas[j]) + corr;
                               dx += w * gx;
                               dy += w * gy;
                               dz += w * gz;
                           }
                       });
    dx = row_sum15(dx);
    dy = row_sum15(dy);
    dz = row_sum15(dz);
    if (sub == 15 && i < N) {
        const float den = nlen[i] + counts[i];
        x_new[3 * i + 0] = x[3 * i + 0] + (dx / p0) / den;
        x_new[3 * i + 1] = x[3 * i + 1] + (dy / p0) / den;
        x_new[3 * i + 2] = x[3 * i + 2] + (dz / p0) / den;
    }
}

// confirm_guess_hidden_particles (:1323-1337)
__global__ void __launch_bounds__(256)
pbf_confirm_kernel(float *__restrict__ xyz, const float *__restrict__ estimate, float *__restrict__ velocity, int N,
                   float secs, float eps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float dx = estimate[3 * i] - xyz[3 * i], dy = estimate[3 * i + 1] - xyz[3 * i + 1],
                dz = estimate[3 * i + 2] - xyz[3 * i + 2];
    const bool still = sqrtf(dx * dx + dy * dy + dz * dz) < eps;
    velocity[3 * i + 0] = still ? 0.f : dx / secs;
    velocity[3 * i + 1] = still ? 0.f : dy / secs;
    velocity[3 * i + 2] = still ? 0.f : dz / secs;
    if (!still) {
        xyz[3 * i + 0] = estimate[3 * i + 0];
        xyz[3 * i + 1] = estimate[3 * i + 1];
        xyz[3 * i + 2] = estimate[3 * i + 2];
    }
}

// per grid slot: the given velocity of the hidden particle stored there (update_visual_particles)
__global__ void __launch_bounds__(256)
slot_given_velocity_kernel(const float4 *__restrict__ rec, int N, const float *__restrict__ velocity,
                           float4 *__restrict__ u) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= N) return;
    const uint32_t j = __float_as_uint(rec[s].w);
    u[s] = make_float4(velocity[3 * j], velocity[3 * j + 1], velocity[3 * j + 2], 0.f);
}

// ---- mean squared distance to the 3 nearest neighbours (simple-knn distCUDA2) -----------
// The reference (submodules/simple-knn/simple_knn.cu:134-166) finds the exact 3 nearest other points with a
// Morton sort + box pruning and returns (d1 + d2 + d3) / 3 of the squared distances.  Here: the uniform hash
// grid of this file, searched ring by ring (cube shells of cells around the query's cell); after ring r every
// unexamined point is at least r * cell away, so the search stops once the third best squared distance is
// <= (r cell)^2.  Candidates are accepted only if their own integer cell is the visited cell (hash collisions
// must not be counted twice).  Queries that run past kKnnMaxRing rings (isolated points) scan all points.
constexpr int kKnnMaxRing = 6;
__device__ __forceinline__ void knn_update3(float d, float *best) {  // updateKBest<3>, simple_knn.cu:120-131
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (best[j] > d) {
            const float t = best[j];
            best[j] = d;
            d = t;
        }
    }
}
__global__ void __launch_bounds__(256)
knn_mean_dist2_kernel(const float *__restrict__ xyz, int N, float inv_cell, float cell, uint32_t mask,
                      const uint32_t *__restrict__ start, const float4 *__restrict__ rec, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
    const int3 c = cell_of(px, py, pz, inv_cell);
    float best[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f};
    bool done = false;
    for (int r = 0; r <= kKnnMaxRing && !done; r++) {
        for (int dz = -r; dz <= r; dz++)
            for (int dy = -r; dy <= r; dy++)
                for (int dx = -r; dx <= r; dx++) {
                    if (max(max(abs(dx), abs(dy)), abs(dz)) != r) continue;  // shell of ring r only
                    const int3 cc = make_int3(c.x + dx, c.y + dy, c.z + dz);
                    const uint32_t h = cell_hash(cc, mask);
                    for (uint32_t s = start[h]; s < start[h + 1]; s++) {
                        const float4 q = rec[s];
                        if (__float_as_uint(q.w) == (uint32_t)i) continue;
                        const int3 qc = cell_of(q.x, q.y, q.z, inv_cell);
                        if (qc.x != cc.x || qc.y != cc.y || qc.z != cc.z) continue;
                        const float ex = q.x - px, ey = q.y - py, ez = q.z - pz;
                        knn_update3(ex * ex + ey * ey + ez * ez, best);
                    }
                }
        const float reach = (float)r * cell;
        done = best[2] <= reach * reach;
    }
    if (!done) {  // isolated query: exact scan
        best[0] = best[1] = best[2] = 3.402823466e38f;
        for (int j = 0; j < N; j++) {
            if (j == i) continue;
            const float ex = xyz[3 * j] - px, ey = xyz[3 * j + 1] - py, ez = xyz[3 * j + 2] - pz;
            knn_update3(ex * ex + ey * ey + ez * ez, best);
        }
    }
    out[i] = (best[0] + best[1] + best[2]) / 3.0f;
}

// ---- optimiser step of the particle positions (fnx_adam_step) ---------------------------------------
// g = ((g0 s0 + g1 s1) + g2 s2) * inv_batch, then torch.optim.Adam's update (amsgrad off, no weight decay),
// fp32 throughout like torch's fused kernel.  `step` holds the number of steps taken so far; every workgroup
// reads the old value, and the workgroup that finishes LAST (a self-resetting arrival counter) advances it, so
// the step costs one launch.  The counter is caller-provided per-optimiser state (`arrived`, one zero-initialised
// word next to `step`), so steps of different optimisers may be in flight on different streams.
// scaled_out (optional) receives the updated x * scale -- the positions in simulation units the next
// iteration's neighbour search starts from.
__global__ void __launch_bounds__(256)
adam_step_kernel(float *__restrict__ x, int n, const float *__restrict__ g0, float s0, const float *__restrict__ g1,
                 float s1, const float *__restrict__ g2, float s2, float inv_batch, float *__restrict__ m,
                 float *__restrict__ v, float *__restrict__ step, float lr, float b1, float b2, float omb1,
                 float omb2, float eps, float *__restrict__ grad_out, float *__restrict__ scaled_out, float scale,
                 unsigned int *__restrict__ arrived) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float t = step[0] + 1.0f;
    if (i < n) {
        float g = 0.f;
        if (g0) g = g + g0[i] * s0;
        if (g1) g = g + g1[i] * s1;
        if (g2) g = g + g2[i] * s2;
        g = g * inv_batch;
        if (grad_out) grad_out[i] = g;
        const float mi = m[i] + omb1 * (g - m[i]);  // exp_avg.lerp_(grad, 1 - beta1); 1 - beta rounded from double like torch
        const float vi = b2 * v[i] + omb2 * g * g;  // exp_avg_sq
        m[i] = mi;
        v[i] = vi;
        const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
        const float step_size = lr / bc1;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        const float xn = x[i] - step_size * (mi / denom);
        x[i] = xn;
        if (scaled_out) scaled_out[i] = xn * scale;
    }
    __syncthreads();  // every thread of the workgroup has read step[0]
    if (threadIdx.x == 0 && !(t < 0.5f)) {  // (always true: t >= 1) the comparison makes the read complete first
        if (atomicAdd(arrived, 1u) == gridDim.x - 1) {  // all workgroups have read the old value
            step[0] = t;
            *arrived = 0;
        }
    }
}

// fnx_adam_step_grid, first launch: the same step, one thread per PARTICLE (three consecutive floats), which then counts
// the particle into its bucket of the hash grid over the updated x * scale; the workgroup that arrives last also scans
// the bucket counts (grid_scan_block) -- Adam, zero-fill, count and scan of the per-iteration grid build in one launch.
// `count` is all zero on entry (the second launch leaves it so).
__global__ void __launch_bounds__(1024)
adam_count_scan_kernel(float *__restrict__ x, int N, const float *__restrict__ g0, float s0,
                       const float *__restrict__ g1, float s1, const float *__restrict__ g2, float s2, float inv_batch,
                       float *__restrict__ m, float *__restrict__ v, float *__restrict__ step, float lr, float b1,
                       float b2, float omb1, float omb2, float eps, float *__restrict__ grad_out,
                       float *__restrict__ scaled_out, float scale, unsigned int *__restrict__ arrived, float inv_cell,
                       uint32_t M, uint32_t *__restrict__ count, uint32_t *__restrict__ start,
                       uint32_t *__restrict__ cursor) {
    __shared__ uint32_t s_wave[kScanChunks][16];
    __shared__ uint32_t s_last;
    const int p = blockIdx.x * 1024 + threadIdx.x;
    const float t = step[0] + 1.0f;
    if (p < N) {
        const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
        const float step_size = lr / bc1;
        float sx[3];
        // every input of the particle is requested up front, unconditionally (an absent gradient term reads another term's
        // array and is not used): with the loads under `if (g1)` ... each sat in front of its own s_waitcnt vmcnt(0) -- a dozen
        // memory round trips in a row in the first kernel of every iteration (round 5)
        const float *any = g0 ? g0 : (g1 ? g1 : (g2 ? g2 : m));
        const float *q0 = g0 ? g0 : any, *q1 = g1 ? g1 : any, *q2 = g2 ? g2 : any;
        float a0[3], a1[3], a2[3], mo[3], vo[3], xo[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = 3 * p + k;
            a0[k] = q0[i];
            a1[k] = q1[i];
            a2[k] = q2[i];
            mo[k] = m[i];
            vo[k] = v[i];
            xo[k] = x[i];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = 3 * p + k;
            float g = 0.f;
            if (g0) g = g + a0[k] * s0;
            if (g1) g = g + a1[k] * s1;
            if (g2) g = g + a2[k] * s2;
            g = g * inv_batch;
            if (grad_out) grad_out[i] = g;
            const float mi = mo[k] + omb1 * (g - mo[k]);
            const float vi = b2 * vo[k] + omb2 * g * g;
            m[i] = mi;
            v[i] = vi;
            const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
            const float xn = xo[k] - step_size * (mi / denom);
            x[i] = xn;
            sx[k] = xn * scale;
            scaled_out[i] = sx[k];
        }
        atomicAdd(&count[cell_hash(cell_of(sx[0], sx[1], sx[2], inv_cell), M - 1)], 1u);
    }
    __syncthreads();  // every thread has read step[0] and issued its count
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = (!(t < 0.5f) && atomicAdd(arrived, 1u) == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        step[0] = t;
        *arrived = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other workgroups' counts (L2 atomics) before the loads below
    grid_scan_block(M, count, start, cursor, s_wave);
}

// second launch: every particle takes its slot (position + id, and its velocity (x - prev) / secs for the
// hidden -> visual interpolation), and the bucket counts are cleared for the next step
__global__ void __launch_bounds__(256)
grid_fill_velocity_kernel(const float *__restrict__ xyz, int N, float inv_cell, uint32_t M,
                          const uint32_t *__restrict__ start, uint32_t *__restrict__ cursor, float4 *__restrict__ rec,
                          const float *__restrict__ prev, float secs, float4 *__restrict__ u,
                          uint32_t *__restrict__ count) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    for (uint32_t k = gid; k < M; k += gridDim.x * 256u) count[k] = 0u;
    if (gid >= (uint32_t)N) return;
    const float x = xyz[3 * gid], y = xyz[3 * gid + 1], z = xyz[3 * gid + 2];
    const uint32_t h = cell_hash(cell_of(x, y, z, inv_cell), M - 1);
    const uint32_t slot = start[h] + atomicAdd(&cursor[h], 1u);
    rec[slot] = make_float4(x, y, z, __uint_as_float(gid));
    if (prev) u[slot] = make_float4((x - prev[3 * gid]) / secs, (y - prev[3 * gid + 1]) / secs, (z - prev[3 * gid + 2]) / secs, 0.f);
}

// Sum over 32 consecutive lanes; totals land in lanes 31 and 63.
__device__ __forceinline__ float half_sum31(float v) {
    v = row_sum15(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
    return v;
}

// ---- pairwise distance loss (fnx_distance_loss) --------------------------------------------------------
// utils/loss_utils.py:98-121: loss = sum over ORDERED pairs i != j with d_ij < thr of (thr - d_ij)^2 (the dense
// torch.cdist matrix holds every unordered pair twice); d loss / d x_i = -4 sum_j (thr - d_ij) (x_i - x_j) / d_ij,
// zero for coincident points (cdist's backward).  Radius-limited: hash grid with cell = 2 thr, so everything within
// thr of a point lies in the 2 x 2 x 2 cells on the point's side of its own cell (8 buckets instead of 27; they are
// part of the 3 x 3 x 3 neighbourhood, hence 8 different buckets, see cell_hash); 8 lanes per point, one bucket each.
__global__ void __launch_bounds__(256)
distance_loss_kernel(const float *__restrict__ xyz, int N, float inv_cell, float thr, uint32_t mask,
                     const uint32_t *__restrict__ start, const float4 *__restrict__ rec, float *__restrict__ partial,
                     float *__restrict__ grad) {
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
    float acc = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    if (i < N) {
        const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        const int3 c = cell_of(px, py, pz, inv_cell);
        const float fx = px * inv_cell - (float)c.x, fy = py * inv_cell - (float)c.y, fz = pz * inv_cell - (float)c.z;
        const int3 cc = make_int3(c.x + ((sub & 1) ? (fx < 0.5f ? -1 : 1) : 0), c.y + ((sub & 2) ? (fy < 0.5f ? -1 : 1) : 0),
                                  c.z + ((sub & 4) ? (fz < 0.5f ? -1 : 1) : 0));
        const uint32_t h = cell_hash(cc, mask);
        const float thr2 = thr * thr;
        // four records of the bucket per step, their loads in flight together (the kernel is pure latency, and it runs
        // next to the rasteriser's blend forward: the shorter its waves live, the less they displace); the terms are
        // added in bucket order, exactly as one by one
        for (uint32_t s = start[h], s1 = start[h + 1]; s < s1; s += 4) {
            float4 q[4];
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = rec[min(s + (uint32_t)k, s1 - 1u)];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (s + (uint32_t)k >= s1 || (int)__float_as_uint(q[k].w) == i) continue;
                const float ex = px - q[k].x, ey = py - q[k].y, ez = pz - q[k].z;
                const float r2 = ex * ex + ey * ey + ez * ez;
                if (!(r2 < thr2)) continue;
                const float d = sqrtf(r2);
                const float t = thr - d;
                if (!(t > 0.f)) continue;
                acc += t * t;
                if (d > 0.f) {
                    const float kk = -4.0f * t / d;
                    ax += kk * ex;
                    ay += kk * ey;
                    az += kk * ez;
                }
            }
        }
    }
    if (grad) {
        ax = oct_sum7(ax);
        ay = oct_sum7(ay);
        az = oct_sum7(az);
        if (sub == 7 && i < N) {
            grad[3 * i + 0] = ax;
            grad[3 * i + 1] = ay;
            grad[3 * i + 2] = az;
        }
    }
    __shared__ float s_w[4];
    acc = wave_sum63(acc);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// per grid slot: velocity of the hidden particle stored there, u = (x - x_prev) / secs
__global__ void __launch_bounds__(256)
slot_velocity_kernel(const float4 *__restrict__ rec, int N, const float *__restrict__ prev, float secs,
                     float4 *__restrict__ u) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= N) return;
    const float4 q = rec[s];
    const uint32_t j = __float_as_uint(q.w);
    u[s] = make_float4((q.x - prev[3 * j]) / secs, (q.y - prev[3 * j + 1]) / secs, (q.z - prev[3 * j + 2]) / secs, 0.f);
}

// gm_dynamics.py:1453-1498.  8 lanes per visual particle (a hidden bucket holds ~8 particles).
__global__ void __launch_bounds__(256)
visual_forward_kernel(const float *__restrict__ visual, int V, float inv_cell, float H2, float term1, float secs,
                      float eps, uint32_t mask, const uint32_t *__restrict__ start, const float4 *__restrict__ rec,
                      const float4 *__restrict__ u, float *__restrict__ out, float *__restrict__ sum_w,
                      float *__restrict__ wvel) {
    const int v = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
    const int vv = min(v, V - 1);
    const float px = visual[3 * vv], py = visual[3 * vv + 1], pz = visual[3 * vv + 2];
    float S = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    for_neighbours<8>(sub, px, py, pz, inv_cell, H2, mask, start, rec,
                      [&](uint32_t s, uint32_t, float, float, float, float r2) {
                          const float t = H2 - r2;
                          const float w = term1 * (t * t * t);
                          const float4 uj = u[s];
                          S += w;
                          ax += uj.x * w;
                          ay += uj.y * w;
                          az += uj.z * w;
                      });
    S = oct_sum7(S);
    ax = oct_sum7(ax);
    ay = oct_sum7(ay);
    az = oct_sum7(az);
    if (sub == 7 && v < V) {
        sum_w[v] = S;
        wvel[3 * v + 0] = ax;
        wvel[3 * v + 1] = ay;
        wvel[3 * v + 2] = az;
        const float Sc = fmaxf(S, eps);
        out[3 * v + 0] = px + ax * secs / Sc;
        out[3 * v + 1] = py + ay * secs / Sc;
        out[3 * v + 2] = pz + az * secs / Sc;
    }
}

// The same with the cap (gm_dynamics.py:1463-1468, radius(x = hidden, y = visual, max_num_neighbors = KNN_K)): the visual
// particle keeps the hidden particles with index <= cutv[v] (knn_cut_kernel).
__global__ void __launch_bounds__(256)
visual_forward_kcap_kernel(const float *__restrict__ visual, int V, float inv_cell, float H2, float term1, float secs,
                           float eps, uint32_t mask, const uint32_t *__restrict__ start, const float4 *__restrict__ rec,
                           const float4 *__restrict__ u, const uint32_t *__restrict__ cutv, float *__restrict__ out,
                           float *__restrict__ sum_w, float *__restrict__ wvel) {
    const int v = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
    const int vv = min(v, V - 1);
    const float px = visual[3 * vv], py = visual[3 * vv + 1], pz = visual[3 * vv + 2];
    const uint32_t cv = cutv[vv];
    float S = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    for_neighbours<8>(sub, px, py, pz, inv_cell, H2, mask, start, rec,
                      [&](uint32_t s, uint32_t j, float, float, float, float r2) {
                          if (j > cv) return;
                          const float t = H2 - r2;
                          const float w = term1 * (t * t * t);
                          const float4 uj = u[s];
                          S += w;
                          ax += uj.x * w;
                          ay += uj.y * w;
                          az += uj.z * w;
                      });
    S = oct_sum7(S);
    ax = oct_sum7(ax);
    ay = oct_sum7(ay);
    az = oct_sum7(az);
    if (sub == 7 && v < V) {
        sum_w[v] = S;
        wvel[3 * v + 0] = ax;
        wvel[3 * v + 1] = ay;
        wvel[3 * v + 2] = az;
        const float Sc = fmaxf(S, eps);
        out[3 * v + 0] = px + ax * secs / Sc;
        out[3 * v + 1] = py + ay * secs / Sc;
        out[3 * v + 2] = pz + az * secs / Sc;
    }
}

// Cell-centric form of the same interpolation.  The visual particles are static within a frame, so their own
// hash grid (built once) lists them cell by cell; one wave takes 64 consecutive slots of that list -- particles
// of one or two cells -- stages the hidden particles of the cell's 27-neighbourhood (position + velocity, ~200
// candidates) in LDS ONCE and every lane walks them with broadcast LDS reads.  The particle-centric kernel above
// re-reads start / records from L2 for every particle in arbitrary order; this one reads them once per cell.
#ifndef FNX_VFC_FMA
#define FNX_VFC_FMA 1  // fused multiply-adds in the interpolation forward walk (0: plain multiplies and adds; A/B in DESIGN 4.9)
#endif
constexpr int kCellCand = 256;  // staged candidates per round (2 x 4 KiB of LDS per one-wave workgroup)
// Work items of a grid for cell-by-cell kernels: every non-empty bucket becomes ceil(count / 64) items
// (first slot, slots), so that one wave handles particles of ONE cell (up to hash collisions) -- a wave that
// takes 64 consecutive slots instead spans up to dozens of sparse cells at the rim of the plume and becomes the
// tail of the launch.  A bucket is cut into full items of 64 and ONE remainder: the interpolation kernel spreads a
// short item's candidates over the idle lanes (64 / pow2ceil(slots) lanes per particle), so a remainder of 16 costs a
// quarter of a full item, where two balanced halves of 40 would cost two.  Item order is arbitrary (atomic append);
// nothing depends on it.
__global__ void __launch_bounds__(256)
grid_cell_items_kernel(uint32_t M, const uint32_t *__restrict__ start, uint2 *__restrict__ items,
                       uint32_t *__restrict__ n_items) {
    const uint32_t h = blockIdx.x * 256 + threadIdx.x;
    if (h >= M) return;
    const uint32_t s0 = start[h], cnt = start[h + 1] - s0;
    if (cnt == 0) return;
    const uint32_t k = (cnt + 63) / 64;
    const uint32_t at = atomicAdd(n_items, k);
    for (uint32_t j = 0; j < k; j++) {
        const uint32_t b = j * 64u;
        items[at + j] = make_uint2(s0 + b, min(64u, cnt - b));
    }
}

template <bool WATCH>
__global__ void __launch_bounds__(64)
visual_forward_cells_kernel(int V, float inv_cell, float H2, float term1, float secs, float eps,
                            const float4 *__restrict__ vrec, const uint2 *__restrict__ items,
                            const uint32_t *__restrict__ n_items, uint32_t hmask,
                            const uint32_t *__restrict__ hstart, const float4 *__restrict__ hrec,
                            const float4 *__restrict__ u, float *__restrict__ out, float *__restrict__ sum_w,
                            float *__restrict__ wvel, float *__restrict__ out_div, float divisor,
                            uint32_t *__restrict__ knn_flags, float knn_k) {
    // a candidate = 24 bytes of LDS, read as 16 + 8 (position + u.x | u.y, u.z): the walk is bound by LDS cycles
    __shared__ float4 s_pos[kCellCand];
    __shared__ float2 s_vel[kCellCand];
    const int lane = threadIdx.x;
    const uint32_t n_work = *n_items;
    for (uint32_t item = blockIdx.x; item < n_work; item += gridDim.x) {
        const uint2 it = items[item];
        // a short item spreads over the wave: `spread` lanes per particle, lane = slice * group + particle, and a
        // particle's lanes take every spread-th candidate each (their partial sums are added at the end)
        uint32_t group = 64;  // pow2ceil(slots), at least 1
        while ((group >> 1) >= it.y && group > 1) group >>= 1;
        const uint32_t spread = 64u / group;
        const uint32_t pi = (uint32_t)lane & (group - 1u), slice = (uint32_t)lane / group;
        const bool valid = pi < it.y;
        const float4 me = vrec[valid ? it.x + pi : it.x];
        const int3 c = cell_of(me.x, me.y, me.z, inv_cell);
        float S = 0.f, ax = 0.f, ay = 0.f, az = 0.f, cnt_n = 0.f;
        unsigned long long todo = __ballot(valid);
        while (todo) {  // one round per distinct cell among the lanes: one, unless cells collide in the bucket
            const int lead = __ffsll((long long)todo) - 1;
            const int3 c0 = make_int3(__shfl(c.x, lead), __shfl(c.y, lead), __shfl(c.z, lead));
            const bool mine = valid && c.x == c0.x && c.y == c0.y && c.z == c0.z;
            // lanes 0..26 and 32..58 own one neighbour bucket each (two lanes per bucket: even / odd records)
            const int bl = lane & 31, par = lane >> 5;
            uint32_t s0 = 0, cnt = 0;
            if (bl < 27) {
                const int dx = bl % 3 - 1, dy = (bl / 3) % 3 - 1, dz = bl / 9 - 1;
                const uint32_t h = cell_hash(make_int3(c0.x + dx, c0.y + dy, c0.z + dz), hmask);
                s0 = hstart[h];
                cnt = hstart[h + 1] - s0;
            }
            uint32_t inc = cnt;  // prefix over the 27 sizes, within each half of the wave
            for (int off = 1; off < 32; off <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)inc, off, 32);
                if (bl >= off) inc += t;
            }
            const uint32_t first = inc - cnt;  // position of this bucket's first record
            const uint32_t total = (uint32_t)__shfl((int)inc, 26);
            for (uint32_t base = 0; base < total; base += kCellCand) {
                const uint32_t n = min((uint32_t)kCellCand, total - base);
                for (uint32_t j = par; j < cnt; j += 4) {  // stage the part of this bucket that falls into the round
                    const uint32_t i0 = first + j, i1 = i0 + 2;
                    const bool in0 = i0 >= base && i0 < base + n, in1 = j + 2 < cnt && i1 >= base && i1 < base + n;
                    // (unconditional loads, all four in flight together: under `if (in0)` / `if (in1)` the second pair was
                    //  requested only when the first had returned)
                    const uint32_t j1 = j + 2 < cnt ? j + 2 : j;
                    const float4 p0 = hrec[s0 + j], v0 = u[s0 + j], p1 = hrec[s0 + j1], v1 = u[s0 + j1];
                    if (in0) { s_pos[i0 - base] = make_float4(p0.x, p0.y, p0.z, v0.x); s_vel[i0 - base] = make_float2(v0.y, v0.z); }
                    if (in1) { s_pos[i1 - base] = make_float4(p1.x, p1.y, p1.z, v1.x); s_vel[i1 - base] = make_float2(v1.y, v1.z); }
                }
                __syncthreads();  // one wave per workgroup: orders the LDS writes before the reads
                if (mine) {
                    const uint32_t steps = (n + spread - 1u) / spread;
#pragma unroll 4
                    for (uint32_t k = 0; k < steps; k++) {
                        const uint32_t i = k * spread + slice;
                        const uint32_t ic = min(i, n - 1u);
                        const float4 q = s_pos[ic];
                        const float2 uj = s_vel[ic];
                        const float ex = me.x - q.x, ey = me.y - q.y, ez = me.z - q.z;
#if FNX_VFC_FMA
                        const float r2 = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
#else
                        const float r2 = ex * ex + ey * ey + ez * ez;
#endif
                        const float t = H2 - r2;
                        const float w = (r2 < H2 && i < n) ? term1 * (t * t * t) : 0.0f;  // +0 terms leave the sums unchanged
                        S += w;
#if FNX_VFC_FMA
                        ax = __builtin_fmaf(q.w, w, ax);
                        ay = __builtin_fmaf(uj.x, w, ay);
                        az = __builtin_fmaf(uj.y, w, az);
#else
                        ax += q.w * w;
                        ay += uj.x * w;
                        az += uj.y * w;
#endif
                        if (WATCH) cnt_n += (r2 < H2 && i < n) ? 1.0f : 0.0f;
                    }
                }
                __syncthreads();
            }
            todo &= ~__ballot(mine);
        }
        for (uint32_t step = group; step < 64u; step <<= 1) {  // add the slices' partial sums (wave-uniform trip count)
            S += __shfl_xor(S, (int)step);
            ax += __shfl_xor(ax, (int)step);
            ay += __shfl_xor(ay, (int)step);
            az += __shfl_xor(az, (int)step);
            if (WATCH) cnt_n += __shfl_xor(cnt_n, (int)step);
        }
        if (WATCH && valid && slice == 0 && cnt_n > knn_k) atomicOr(knn_flags, 4u);  // radius(x = hidden, y = visual) list > K
        if (valid && slice == 0) {
            const uint32_t v = __float_as_uint(me.w);
            sum_w[v] = S;
            wvel[3 * v + 0] = ax;
            wvel[3 * v + 1] = ay;
            wvel[3 * v + 2] = az;
            const float Sc = fmaxf(S, eps);
            const float ox = me.x + ax * secs / Sc, oy = me.y + ay * secs / Sc, oz = me.z + az * secs / Sc;
            out[3 * v + 0] = ox;
            out[3 * v + 1] = oy;
            out[3 * v + 2] = oz;
            if (out_div) {  // the same positions in render units (x / scale_factor), where the rasteriser reads them;
                // torch divides a tensor by a scalar as x * (1 / s) with the reciprocal rounded to fp32: so does this
                const float inv = 1.0f / divisor;
                out_div[3 * v + 0] = ox * inv;
                out_div[3 * v + 1] = oy * inv;
                out_div[3 * v + 2] = oz * inv;
            }
        }
    }
}

// per visual-grid slot, everything the hidden<-visual backward needs from the visual particle stored there,
// with the divisions done once per visual particle instead of once per pair:
//   G = g / Sc (Sc = max(S, eps)),  c2 = [S > eps] secs (g . wvel) / Sc^2
__global__ void __launch_bounds__(256)
slot_visual_payload_kernel(const float4 *__restrict__ rec, int V, const float *__restrict__ sum_w,
                           const float *__restrict__ wvel, const float *__restrict__ g, const float *__restrict__ g2,
                           float scale2, float secs, float eps, float4 *__restrict__ a0) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= V) return;
    const uint32_t v = __float_as_uint(rec[s].w);
    const float S = sum_w[v], Sc = fmaxf(S, eps);
    float gx = g[3 * v], gy = g[3 * v + 1], gz = g[3 * v + 2];
    if (g2) {  // upstream gradient = g + scale2 * g2 (a second term that arrives separately, e.g. the distance loss's)
        gx = gx + scale2 * g2[3 * v];
        gy = gy + scale2 * g2[3 * v + 1];
        gz = gz + scale2 * g2[3 * v + 2];
    }
    float c2 = 0.f;
    if (S > eps) c2 = secs * (gx * wvel[3 * v] + gy * wvel[3 * v + 1] + gz * wvel[3 * v + 2]) / (Sc * Sc);
    a0[s] = make_float4(gx / Sc, gy / Sc, gz / Sc, c2);
}

// squared distance from a point at fractional position f (in cell units, [0,1)) inside its cell to the
// neighbouring cell at offset d along one axis
__device__ __forceinline__ float axis_gap2(float f, int d, float cell) {
    const float g = d < 0 ? f : (d > 0 ? 1.0f - f : 0.0f);
    return (g * cell) * (g * cell);
}

// One wave per hidden particle; the grid holds the VISUAL points (~65 per bucket).  Candidates are tested
// 64 at a time; the ~15 % that lie within H are compacted (ballot + popcount) into a per-wave LDS ring and
// the gradient terms are evaluated on full waves of survivors, instead of running the heavy part of the
// loop with most lanes masked off.  Cells of the 27-neighbourhood that lie entirely beyond H are skipped.
//   d hidden_j = sum_v [ w G_v + (secs (G_v . u_j) - c2_v) dW/dr2 2 (hidden_j - visual_v) ]
__global__ void __launch_bounds__(256)
visual_backward_kernel(const float *__restrict__ hidden, const float *__restrict__ hidden_prev, int N, float inv_cell,
                       float cell, float H2, float term1, float secs, uint32_t mask,
                       const uint32_t *__restrict__ start, const float4 *__restrict__ rec,
                       const float4 *__restrict__ a0, float *__restrict__ dL_dhidden) {
    __shared__ float4 s_e[4][128];   // (ex, ey, ez, r2) of the survivors
    __shared__ uint32_t s_s[4][128];  // their grid slots
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    if (j >= N) return;  // whole wave
    const float hx = hidden[3 * j], hy = hidden[3 * j + 1], hz = hidden[3 * j + 2];
    const float ux = (hx - hidden_prev[3 * j]) / secs, uy = (hy - hidden_prev[3 * j + 1]) / secs,
                uz = (hz - hidden_prev[3 * j + 2]) / secs;
    const int3 c = cell_of(hx, hy, hz, inv_cell);
    const float fx = hx * inv_cell - (float)c.x, fy = hy * inv_cell - (float)c.y, fz = hz * inv_cell - (float)c.z;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    float ax = 0.f, ay = 0.f, az = 0.f;
    uint32_t head = 0, tail = 0;  // ring positions (wave-uniform)
    auto drain = [&](uint32_t n) {  // evaluate n <= 64 survivors starting at head
        if ((uint32_t)lane < n) {
            const uint32_t q = (head + lane) & 127u;
            const float4 e = s_e[w][q];
            const float4 G = a0[s_s[w][q]];
            const float t = H2 - e.w;
            const float wgt = term1 * (t * t * t);
            const float dW = -3.0f * term1 * (t * t);
            const float dLdw = secs * (G.x * ux + G.y * uy + G.z * uz) - G.w;
            const float k = dLdw * dW * 2.0f;
            ax += wgt * G.x + k * e.x;
            ay += wgt * G.y + k * e.y;
            az += wgt * G.z + k * e.z;
        }
        head += n;
    };
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                // the margin keeps every cell whose closest point could be within H despite fp32 rounding of f
                if (axis_gap2(fx, dx, cell) + axis_gap2(fy, dy, cell) + axis_gap2(fz, dz, cell) > H2 * 1.001f) continue;
                const uint32_t h = cell_hash(make_int3(c.x + dx, c.y + dy, c.z + dz), mask);
                const uint32_t s0 = start[h], s1 = start[h + 1];
                for (uint32_t base = s0; base < s1; base += 64) {
                    const uint32_t sl = base + lane;
                    bool pass = false;
                    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (sl < s1) {
                        const float4 q = rec[sl];
                        e.x = hx - q.x;
                        e.y = hy - q.y;
                        e.z = hz - q.z;
                        e.w = e.x * e.x + e.y * e.y + e.z * e.z;
                        pass = e.w < H2;
                    }
                    const unsigned long long m = __ballot(pass);
                    if (pass) {
                        const uint32_t q = (tail + (uint32_t)__popcll(m & lt_mask)) & 127u;
                        s_e[w][q] = e;
                        s_s[w][q] = sl;
                    }
                    tail += (uint32_t)__popcll(m);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's LDS writes before its reads
                    if (tail - head >= 64u) drain(64u);
                }
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    drain(tail - head);
    ax = wave_sum63(ax);
    ay = wave_sum63(ay);
    az = wave_sum63(az);
    if (lane == 63) {
        dL_dhidden[3 * j + 0] = ax;
        dL_dhidden[3 * j + 1] = ay;
        dL_dhidden[3 * j + 2] = az;
    }
}

// The capped backward: one wave per hidden particle j over the VISUAL grid; the pair (v, j) exists iff j <= cutv[v].
// Plain lanes-over-candidates walk (the mode is for parity with capped reference runs, not a tuned path).
__global__ void __launch_bounds__(256)
visual_backward_kcap_kernel(const float *__restrict__ hidden, const float *__restrict__ hidden_prev, int N,
                            float inv_cell, float H2, float term1, float secs, uint32_t mask,
                            const uint32_t *__restrict__ start, const float4 *__restrict__ rec,
                            const float4 *__restrict__ a0, const uint32_t *__restrict__ cutv,
                            float *__restrict__ dL_dhidden) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    if (j >= N) return;  // whole wave
    const float hx = hidden[3 * j], hy = hidden[3 * j + 1], hz = hidden[3 * j + 2];
    const float ux = (hx - hidden_prev[3 * j]) / secs, uy = (hy - hidden_prev[3 * j + 1]) / secs,
                uz = (hz - hidden_prev[3 * j + 2]) / secs;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for_neighbours<64>(lane, hx, hy, hz, inv_cell, H2, mask, start, rec,
                       [&](uint32_t s, uint32_t v, float ex, float ey, float ez, float r2) {
                           if ((uint32_t)j > cutv[v]) return;
                           const float4 G = a0[s];
                           const float t = H2 - r2;
                           const float wgt = term1 * (t * t * t);
                           const float dW = -3.0f * term1 * (t * t);
                           const float k = (secs * (G.x * ux + G.y * uy + G.z * uz) - G.w) * dW * 2.0f;
                           ax += wgt * G.x + k * ex;
                           ay += wgt * G.y + k * ey;
                           az += wgt * G.z + k * ez;
                       });
    ax = wave_sum63(ax);
    ay = wave_sum63(ay);
    az = wave_sum63(az);
    if (lane == 63) {
        dL_dhidden[3 * j + 0] = ax;
        dL_dhidden[3 * j + 1] = ay;
        dL_dhidden[3 * j + 2] = az;
    }
}

// Cell-centric form of the same backward.  The hidden particles of one cell (a work item of the HIDDEN grid,
// fnx_grid_cell_items) see the same 27 visual buckets, so a 4-wave workgroup takes an item: lanes = candidates
// (~1 750 visual slots of the neighbourhood, flattened; position + payload loaded once per CELL instead of once per
// hidden particle, with no start -> record -> payload dependent chain per bucket), and every candidate is tested
// against up to kBwdGroup hidden particles whose positions / velocities sit in scalar registers.  Per-lane sums
// for each hidden particle are reduced across the wave once at the end, the four waves' partials through LDS.
constexpr int kBwdGroup = 8;
#ifndef FNX_BWD_WAVES
#define FNX_BWD_WAVES 4
#endif
constexpr int kBwdWaves = FNX_BWD_WAVES;  // waves per cell item
__global__ void __launch_bounds__(64 * kBwdWaves)
visual_backward_cells_kernel(float inv_cell, float H2, float term1, float secs, const float4 *__restrict__ hrec,
                             const uint2 *__restrict__ hitems, const uint32_t *__restrict__ n_hitems,
                             const float *__restrict__ hidden_prev, uint32_t vmask,
                             const uint32_t *__restrict__ vstart, const float4 *__restrict__ vrec,
                             const float4 *__restrict__ a0, float *__restrict__ dL_dhidden) {
    __shared__ uint32_t s_first[28], s_s0[28];
    __shared__ float s_part[kBwdWaves][kBwdGroup][3];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n_work = *n_hitems;
    for (uint32_t item = blockIdx.x; item < n_work; item += gridDim.x) {
        const uint2 it = hitems[item];
        const bool valid = (uint32_t)lane < it.y;
        // every wave holds the item's hidden particles (lane = particle): position, velocity, id
        const float4 me = hrec[valid ? it.x + lane : it.x];
        const uint32_t jid = __float_as_uint(me.w);
        const float ux = (me.x - hidden_prev[3 * jid]) / secs, uy = (me.y - hidden_prev[3 * jid + 1]) / secs,
                    uz = (me.z - hidden_prev[3 * jid + 2]) / secs;
        const int3 c = cell_of(me.x, me.y, me.z, inv_cell);
        unsigned long long todo = __ballot(valid);
        while (todo) {  // one round per distinct cell among the lanes: one, unless cells collide in the bucket
            const int lead = __ffsll((long long)todo) - 1;
            const int3 c0 = make_int3(__shfl(c.x, lead), __shfl(c.y, lead), __shfl(c.z, lead));
            const bool mine = valid && c.x == c0.x && c.y == c0.y && c.z == c0.z;
            // the 27 visual buckets of the neighbourhood: ranges by lanes 0..26 of wave 0, prefix, into LDS
            __syncthreads();  // the previous round / item is done with s_first / s_s0 / s_part
            if (w == 0) {
                uint32_t s0 = 0, cnt = 0;
                if (lane < 27) {
                    const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
                    const uint32_t h = cell_hash(make_int3(c0.x + dx, c0.y + dy, c0.z + dz), vmask);
                    s0 = vstart[h];
                    cnt = vstart[h + 1] - s0;
                }
                uint32_t inc = cnt;
                for (int off = 1; off < 32; off <<= 1) {
                    const uint32_t t = (uint32_t)__shfl_up((int)inc, off);
                    if (lane >= off) inc += t;
                }
                if (lane < 27) {
                    s_first[lane] = inc - cnt;
                    s_s0[lane] = s0;
                }
                if (lane == 26) s_first[27] = inc;
            }
            __syncthreads();
            const uint32_t total = s_first[27];
            unsigned long long left = __ballot(mine);
            while (left) {  // groups of kBwdGroup hidden particles of this cell
                float hx[kBwdGroup], hy[kBwdGroup], hz[kBwdGroup], vx[kBwdGroup], vy[kBwdGroup], vz[kBwdGroup];
                uint32_t hj[kBwdGroup];
                int ng = 0;
#pragma unroll
                for (int k = 0; k < kBwdGroup; k++) {
                    const bool have = left != 0ull;  // wave-uniform
                    const int src = have ? __ffsll((long long)left) - 1 : 0;
                    if (have) {
                        left &= left - 1ull;
                        ng = k + 1;
                    }
                    // wave-uniform values (scalar registers): particle k of the group
                    hx[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(me.x), src));
                    hy[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(me.y), src));
                    hz[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(me.z), src));
                    vx[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ux), src));
                    vy[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uy), src));
                    vz[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uz), src));
                    hj[k] = (uint32_t)__builtin_amdgcn_readlane((int)jid, src);
                }
                float ax[kBwdGroup], ay[kBwdGroup], az[kBwdGroup];
#pragma unroll
                for (int k = 0; k < kBwdGroup; k++) ax[k] = ay[k] = az[k] = 0.f;
                // wave w takes every kBwdWaves-th 64-candidate chunk, two chunks per step (their loads in flight together)
                for (uint32_t base = (uint32_t)w * 64u; base < total; base += 128u * kBwdWaves) {
                    float4 q[2], G[2];
                    bool in[2];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint32_t i = base + (uint32_t)u * (64u * kBwdWaves) + lane;
                        in[u] = i < total;
                        uint32_t b = 0;  // bucket of flat index i: last b with s_first[b] <= i
#pragma unroll
                        for (int step = 16; step >= 1; step >>= 1)
                            if (b + step < 27u && s_first[b + step] <= i) b += step;
                        const uint32_t slot = in[u] ? s_s0[b] + (i - s_first[b]) : 0u;
                        q[u] = vrec[slot];
                        G[u] = a0[slot];
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++) {
#pragma unroll
                        for (int k = 0; k < kBwdGroup; k++) {
                            if (k < ng) {  // wave-uniform
                                // fused multiply-adds, spelled out (the library is built with -ffp-contract=off): the kernel is
                                // bound by its VALU instructions (82 % busy), 37 per (candidate, hidden particle) without them
                                const float ex = hx[k] - q[u].x, ey = hy[k] - q[u].y, ez = hz[k] - q[u].z;
                                const float r2 = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                                if (in[u] && r2 < H2) {
                                    const float t = H2 - r2, t2 = t * t;
                                    const float wgt = term1 * (t2 * t);
                                    const float dW2 = (-6.0f * term1) * t2;  // 2 dW/dr2
                                    const float dot = __builtin_fmaf(G[u].z, vz[k], __builtin_fmaf(G[u].y, vy[k], G[u].x * vx[k]));
                                    const float kk = __builtin_fmaf(secs, dot, -G[u].w) * dW2;
                                    ax[k] = __builtin_fmaf(kk, ex, __builtin_fmaf(wgt, G[u].x, ax[k]));
                                    ay[k] = __builtin_fmaf(kk, ey, __builtin_fmaf(wgt, G[u].y, ay[k]));
                                    az[k] = __builtin_fmaf(kk, ez, __builtin_fmaf(wgt, G[u].z, az[k]));
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < kBwdGroup; k++) {
                    if (k < ng) {
                        const float sx = wave_sum63(ax[k]), sy = wave_sum63(ay[k]), sz = wave_sum63(az[k]);
                        if (lane == 63) {
                            s_part[w][k][0] = sx;
                            s_part[w][k][1] = sy;
                            s_part[w][k][2] = sz;
                        }
                    }
                }
                __syncthreads();
                if (w == 0 && lane < 3 * ng) {
                    const int k = lane / 3, d = lane - 3 * k;
                    uint32_t j = hj[0];
#pragma unroll
                    for (int q = 1; q < kBwdGroup; q++) j = (k == q) ? hj[q] : j;
                    float sum = 0.f;
#pragma unroll
                    for (int ww = 0; ww < kBwdWaves; ww++) sum += s_part[ww][k][d];
                    dL_dhidden[3 * j + d] = sum;
                }
                __syncthreads();  // s_part is rewritten by the next group
            }
            todo &= ~__ballot(mine);
        }
    }
}

thread_local char g_err[512] = "";
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int hip_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FNX_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return FNX_OK;
}
inline float poly6_term1(float H) {  // gm_dynamics.py:130: 315 / (64 pi H^9), evaluated in double
    const double h = (double)H;
    return (float)(315.0 / (64.0 * M_PI * std::pow(h, 9.0)));
}

}  // namespace

extern "C" {

int fnx_physics_abi_version(void) { return FNX_PHYSICS_ABI_VERSION; }
const char *fnx_physics_last_error(void) { return g_err; }
size_t fnx_grid_bytes(int N) { return grid_bytes(N); }

int fnx_grid_build(const float *xyz, int N, float cell, char *grid, fnx_stream_t stream) {
    if (N < 0 || !grid || cell <= 0.f || (N > 0 && !xyz)) return fail(FNX_ERR_INVALID_ARG, "grid_build: bad argument");
    hipStream_t s = (hipStream_t)stream;
    GridView g = carve(grid, N);
    hipLaunchKernelGGL(zero_u32_kernel, dim3((g.M + 255) / 256), dim3(256), 0, s, g.count, (size_t)g.M);
    const float inv = 1.0f / cell;
    if (N > 0)
        hipLaunchKernelGGL(grid_count_kernel, dim3((N + 255) / 256), dim3(256), 0, s, xyz, N, inv, g.M - 1, g.count);
    hipLaunchKernelGGL(grid_scan_kernel, dim3(1), dim3(1024), 0, s, g.M, g.count, g.start, g.cursor);
    if (N > 0)
        hipLaunchKernelGGL(grid_fill_kernel, dim3((N + 255) / 256), dim3(256), 0, s, xyz, N, inv, g.M - 1, g.start,
                           g.cursor, g.rec);
    return hip_check("grid_build");
}

int fnx_density_forward(const float *xyz, int N, const float *imass, float H, float p0, const char *grid,
                        float *p_ratio, fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (N < 0 || !xyz || !imass || !grid || !p_ratio) return fail(FNX_ERR_INVALID_ARG, "density_forward: bad argument");
    GridView g = carve(const_cast<char *>(grid), N);
    hipLaunchKernelGGL(density_forward_kernel, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, xyz, N, imass,
                       1.0f / H, H * H, poly6_term1(H), p0, g.M - 1, g.start, g.rec, p_ratio);
    return hip_check("density_forward");
}

int fnx_density_backward(const float *xyz, int N, const float *imass, float H, float p0, const char *grid,
                         const float *dL_dp_ratio, float *dL_dxyz, fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (N < 0 || !xyz || !imass || !grid || !dL_dp_ratio || !dL_dxyz)
        return fail(FNX_ERR_INVALID_ARG, "density_backward: bad argument");
    GridView g = carve(const_cast<char *>(grid), N);
    hipLaunchKernelGGL(density_backward_kernel, dim3((N + kDPerWg - 1) / kDPerWg), dim3(256), 0, (hipStream_t)stream, xyz, N,
                       imass, 1.0f / H, H * H, poly6_term1(H), p0, g.M - 1, g.start, g.rec, dL_dp_ratio, 1.0f, dL_dxyz);
    return hip_check("density_backward");
}

int fnx_knn_cut(const float *queries, int Nq, int N_points, float H, int K, const char *points_grid, uint32_t *cut,
                fnx_stream_t stream) {
    if (Nq == 0) return FNX_OK;
    if (Nq < 0 || N_points < 0 || K < 1 || !queries || !points_grid || !cut)
        return fail(FNX_ERR_INVALID_ARG, "knn_cut: bad argument");
    GridView g = carve(const_cast<char *>(points_grid), N_points);
    hipLaunchKernelGGL(knn_cut_kernel, dim3((Nq + 256 / kCutLanes - 1) / (256 / kCutLanes)), dim3(256), 0,
                       (hipStream_t)stream, queries, Nq, 1.0f / H, H * H, g.M - 1, g.start, g.rec, (uint32_t)K, cut);
    return hip_check("knn_cut");
}

int fnx_density_forward_kcap(const float *xyz, int N, const float *imass, float H, float p0, const char *grid,
                             const uint32_t *cut, float *p_ratio, fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (N < 0 || !xyz || !imass || !grid || !cut || !p_ratio)
        return fail(FNX_ERR_INVALID_ARG, "density_forward_kcap: bad argument");
    GridView g = carve(const_cast<char *>(grid), N);
    hipLaunchKernelGGL(density_forward_kcap_kernel, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, xyz, N, imass,
                       1.0f / H, H * H, poly6_term1(H), p0, g.M - 1, g.start, g.rec, cut, p_ratio);
    return hip_check("density_forward_kcap");
}

int fnx_density_backward_kcap(const float *xyz, int N, const float *imass, float H, float p0, const char *grid,
                              const uint32_t *cut, const float *dL_dp_ratio, float *dL_dxyz, fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (N < 0 || !xyz || !imass || !grid || !cut || !dL_dp_ratio || !dL_dxyz)
        return fail(FNX_ERR_INVALID_ARG, "density_backward_kcap: bad argument");
    GridView g = carve(const_cast<char *>(grid), N);
    hipLaunchKernelGGL(density_backward_kcap_kernel, dim3((N + kDPerWg - 1) / kDPerWg), dim3(256), 0, (hipStream_t)stream,
                       xyz, N, imass, 1.0f / H, H * H, poly6_term1(H), p0, g.M - 1, g.start, g.rec, cut, dL_dp_ratio,
                       dL_dxyz);
    return hip_check("density_backward_kcap");
}

// K-cap watch (include/fnx_physics.h): per host thread, armed until cleared
thread_local uint32_t *t_knn_flags = nullptr;
thread_local int t_knn_k = 0;
int fnx_knn_watch(uint32_t *flags, int K) {
    if (flags && K < 1) return fail(FNX_ERR_INVALID_ARG, "knn_watch: K must be >= 1");
    t_knn_flags = flags;
    t_knn_k = flags ? K : 0;
    return FNX_OK;
}

int fnx_physical_stage(const float *x_nn, int N, float scale_factor, const float *x_est, const float *x_prev,
                       const float *imass, const float *buoyancy, const float *force, float buoyancy_max_y, float H,
                       float p0, float secs, float lam_e, float lam_g, float lam_n, char *est_grid, int build_est_grid,
                       char *guess_grid, float *scratch, float *terms, float *loss, float *grad, fnx_stream_t stream) {
    if (N <= 0 || !x_nn || !x_est || !x_prev || !imass || !buoyancy || !force || !est_grid || !guess_grid ||
        !scratch || !terms || !loss || !grad || H <= 0.f || secs == 0.f)
        return fail(FNX_ERR_INVALID_ARG, "physical_stage: bad argument");
    hipStream_t s = (hipStream_t)stream;
    float *x = scratch, *xg = scratch + 3 * (size_t)N, *d1 = scratch + 6 * (size_t)N, *d2 = d1 + N;
    float *dx_est = d2 + N, *dg = dx_est + 3 * (size_t)N;
    const int nb_e = (N + 255) / 256, nb_d = (N + kDPerWg - 1) / kDPerWg;
    float *part_e = dg + 3 * (size_t)N, *part_g = part_e + nb_e, *part_n = part_g + nb_d;  // total <= 15 N + 64 floats
    hipLaunchKernelGGL(stage_points_kernel, dim3(nb_e), dim3(256), 0, s, x_nn, N, scale_factor, x_est, x_prev,
                       buoyancy, force, buoyancy_max_y, secs, x, xg, part_e);
    const float inv = 1.0f / H, H2 = H * H, t1 = poly6_term1(H);
    if (lam_g > 0.f) {
        if (build_est_grid)
            if (int rc = fnx_grid_build(x, N, H, est_grid, stream)) return rc;
        GridView g = carve(est_grid, N);
        if (t_knn_flags)
            hipLaunchKernelGGL(density_term_kernel<true>, dim3(nb_d), dim3(256), 0, s, x, N, imass, inv, H2, t1, p0,
                               g.M - 1, g.start, g.rec, d1, part_g, t_knn_flags, (float)t_knn_k, 1u);
        else
            hipLaunchKernelGGL(density_term_kernel<false>, dim3(nb_d), dim3(256), 0, s, x, N, imass, inv, H2, t1, p0,
                               g.M - 1, g.start, g.rec, d1, part_g, (uint32_t *)nullptr, 0.f, 0u);
        hipLaunchKernelGGL(density_backward_kernel, dim3(nb_d), dim3(256), 0, s, x, N, imass, inv, H2, t1, p0,
                           g.M - 1, g.start, g.rec, d1, 2.0f * lam_g / (float)N, dx_est);
    }
    if (lam_n > 0.f) {
        if (int rc = fnx_grid_build(xg, N, H, guess_grid, stream)) return rc;
        GridView g = carve(guess_grid, N);
        if (t_knn_flags)
            hipLaunchKernelGGL(density_term_kernel<true>, dim3(nb_d), dim3(256), 0, s, xg, N, imass, inv, H2, t1, p0,
                               g.M - 1, g.start, g.rec, d2, part_n, t_knn_flags, (float)t_knn_k, 2u);
        else
            hipLaunchKernelGGL(density_term_kernel<false>, dim3(nb_d), dim3(256), 0, s, xg, N, imass, inv, H2, t1, p0,
                               g.M - 1, g.start, g.rec, d2, part_n, (uint32_t *)nullptr, 0.f, 0u);
        hipLaunchKernelGGL(density_backward_kernel, dim3(nb_d), dim3(256), 0, s, xg, N, imass, inv, H2, t1, p0,
                           g.M - 1, g.start, g.rec, d2, 2.0f * lam_n / (float)N, dg);
    }
    hipLaunchKernelGGL(stage_combine_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, scale_factor, x, x_est, dx_est,
                       dg, buoyancy, buoyancy_max_y, secs, lam_e, lam_g, lam_n, part_e, terms, loss, grad);
    return hip_check("physical_stage");
}

int fnx_pbf_predict(const float *xyz, float *velocity, float *buoyancy, float *force, float *estimate_xyz, float *counts,
                    int N, const float *gravity, float alpha, float secs, float scale_max_y, float decay_rate,
                    fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (N < 0 || !xyz || !velocity || !buoyancy || !force || !estimate_xyz || !counts || !gravity)
        return fail(FNX_ERR_INVALID_ARG, "pbf_predict: bad argument");
    hipLaunchKernelGGL(pbf_predict_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, xyz, velocity,
                       buoyancy, force, estimate_xyz, counts, N, gravity[0], gravity[1], gravity[2], alpha, secs,
                       scale_max_y, decay_rate);
    return hip_check("pbf_predict");
}

int fnx_pbf_neighbor_counts(const float *xyz, int N, float H, char *grid, int *counts, fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (N < 0 || !xyz || !grid || !counts || H <= 0.f) return fail(FNX_ERR_INVALID_ARG, "pbf_neighbor_counts: bad argument");
    if (int rc = fnx_grid_build(xyz, N, H, grid, stream)) return rc;
    GridView g = carve(grid, N);
    hipLaunchKernelGGL(pbf_count_kernel, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, xyz, N, 1.0f / H, H * H,
                       g.M - 1, g.start, g.rec, counts);
    return hip_check("pbf_neighbor_counts");
}

int fnx_pbf_project(float *estimate_xyz, const float *velocity, float *force, const float *imass, const float *counts,
                    int N, float H, float p0, float k, float relaxation, float K_P, float E_P, float DQ_P, float eps,
                    char *grid, float *scratch, fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (N < 0 || !estimate_xyz || !velocity || !force || !imass || !counts || !grid || !scratch || H <= 0.f)
        return fail(FNX_ERR_INVALID_ARG, "pbf_project: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = fnx_grid_build(estimate_xyz, N, H, grid, stream)) return rc;
    GridView g = carve(grid, N);
    float *lambdas = scratch, *nlen = scratch + N, *x_new = scratch + 2 * (size_t)N;
    const float H2 = H * H, p6t = poly6_term1(H);
    const float spt = (float)(45.0 / (M_PI * std::pow((double)H, 6.0)));  // gm_dynamics.py:131
    // lamb_corr_denom = poly6(DQ_P^2 H^2) (:133), evaluated like the reference's fp32 tensor expression
    const float r2q = DQ_P * DQ_P * H * H, tq = H2 - r2q;
    const float corr_denom = (r2q < H2) ? p6t * (tq * tq * tq) : 0.0f;
    hipLaunchKernelGGL(pbf_lambda_kernel, dim3((N + 15) / 16), dim3(256), 0, s, estimate_xyz, velocity, force, imass, N,
                       1.0f / H, H, H2, p6t, spt, p0, k, relaxation, eps, g.M - 1, g.start, g.rec, lambdas, nlen);
    hipLaunchKernelGGL(pbf_delta_kernel, dim3((N + 15) / 16), dim3(256), 0, s, estimate_xyz, N, 1.0f / H, H, H2, p6t, spt,
                       p0, K_P, E_P, corr_denom, eps, g.M - 1, g.start, g.rec, lambdas, nlen, counts, x_new);
    (void)hipMemcpyAsync(estimate_xyz, x_new, (size_t)N * 12, hipMemcpyDeviceToDevice, s);
    return hip_check("pbf_project");
}

int fnx_pbf_confirm(float *xyz, const float *estimate_xyz, float *velocity, int N, float secs, float eps,
                    fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (N < 0 || !xyz || !estimate_xyz || !velocity || secs == 0.f) return fail(FNX_ERR_INVALID_ARG, "pbf_confirm: bad argument");
    hipLaunchKernelGGL(pbf_confirm_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, xyz, estimate_xyz,
                       velocity, N, secs, eps);
    return hip_check("pbf_confirm");
}

int fnx_visual_advect(float *visual, int V, const float *hidden, const float *velocity, int N, float H, float secs,
                      float eps, char *hidden_grid, float *scratch, fnx_stream_t stream) {
    if (V == 0) return FNX_OK;
    if (V < 0 || N < 0 || !visual || !hidden_grid || !scratch || (N > 0 && (!hidden || !velocity)) || H <= 0.f)
        return fail(FNX_ERR_INVALID_ARG, "visual_advect: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = fnx_grid_build(hidden, N, H, hidden_grid, stream)) return rc;
    GridView g = carve(hidden_grid, N);
    if (N > 0)
        hipLaunchKernelGGL(slot_given_velocity_kernel, dim3((N + 255) / 256), dim3(256), 0, s, g.rec, N, velocity, g.aux0);
    hipLaunchKernelGGL(visual_forward_kernel, dim3((V + 31) / 32), dim3(256), 0, s, visual, V, 1.0f / H, H * H,
                       poly6_term1(H), secs, eps, g.M - 1, g.start, g.rec, g.aux0, visual, scratch, scratch + V);
    return hip_check("visual_advect");
}

int fnx_knn_mean_dist2(const float *xyz, int N, float cell, char *grid, float *mean_dist2, fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (N < 0 || !xyz || !grid || !mean_dist2 || !(cell > 0.f)) return fail(FNX_ERR_INVALID_ARG, "knn_mean_dist2: bad argument");
    if (int rc = fnx_grid_build(xyz, N, cell, grid, stream)) return rc;
    GridView g = carve(grid, N);
    hipLaunchKernelGGL(knn_mean_dist2_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, xyz, N, 1.0f / cell,
                       cell, g.M - 1, g.start, g.rec, mean_dist2);
    return hip_check("knn_mean_dist2");
}

// ---- the same loss on per-bucket LINKED LISTS (fnx_distance_loss_lists) ---------------------------------------
// The grid version above is five dependent launches (zero, count, scan by one workgroup, fill, search with eight lanes
// per point = 25 000 waves for 200 k points); next to the rasteriser's blend forward those waves take registers the
// forward's workgroups are waiting for, and the branch cost the iteration ~60 us wherever it was placed.  Here a bucket
// is a list threaded through the points: two launches, ONE thread per point, and a bucket table large enough that a
// list holds about one point.
//   build   old = atomicExch(head[bucket], call stamp | i); node i = (x, y, z, next = old's index if old carries this
//           call's stamp).  The table is never cleared: an entry of an earlier call fails the stamp test.
//   search  the eight buckets a point's threshold ball can touch (the 2 x 2 x 2 cells on the point's side of its own
//           cell: eight different buckets, see cell_hash), their lists walked side by side (eight loads in flight);
//           value and gradient as above; per-workgroup sums, added up in workgroup order by the workgroup that arrives
//           last (deterministic given the lists; the order inside a list is the arrival order of the atomics, as the
//           slot order of the grid version is).
// `table`: caller-owned, persistent, zero-filled once (fnx_distance_table_bytes): header | head u64[M] | nodes float4[N] |
// partial sums.
struct DistTable {
    uint32_t *hdr;  // [0] call stamp, [1] arrival counter
    unsigned long long *head;
    float4 *node;
    float *partial;
    uint32_t M;
};
inline uint32_t dist_buckets_for(int N) {
    uint32_t m = 4096;
    while (m < (1u << 20) && m < 2u * (uint32_t)(N > 0 ? N : 1)) m <<= 1;
    return m;
}
inline size_t dist_table_bytes(int N) {
    const size_t n = (size_t)(N > 0 ? N : 0);
    size_t off = align_up(256);
    off = align_up(off + (size_t)dist_buckets_for(N) * 8);
    off = align_up(off + n * 16);
    off = align_up(off + ((n + 255) / 256 + 1) * 4);
    return off + kAlign;
}
inline DistTable carve_dist(char *blob, int N) {
    char *b = (char *)(((uintptr_t)blob + kAlign - 1) / kAlign * kAlign);
    const size_t n = (size_t)(N > 0 ? N : 0);
    DistTable t;
    t.M = dist_buckets_for(N);
    size_t off = 0;
    t.hdr = (uint32_t *)(b + off);            off = align_up(off + 256);
    t.head = (unsigned long long *)(b + off); off = align_up(off + (size_t)t.M * 8);
    t.node = (float4 *)(b + off);             off = align_up(off + n * 16);
    t.partial = (float *)(b + off);
    return t;
}
constexpr uint32_t kDistNone = 0xFFFFFFFFu;
// Publish one workgroup's partial sum for the workgroup that arrives last, WITHOUT a release fence: __threadfence() is
// `buffer_wbl2` -- it writes back everything the XCD's L2 holds dirty (these kernels have just stored 2.4 MB of gradients),
// once per workgroup, 782 times per launch: 10 of the 29 us of the list-mode kernel (round 5, FNX_LAB_VERLET=1).  The
// partial goes out as a device-scope atomic exchange instead (performed at the coherent level), its return is waited for,
// then the arrival counter is bumped; the last workgroup invalidates its L2 (acquire fence) and reads the partials.
__device__ __forceinline__ bool publish_partial_and_arrive(float *partial, float value, uint32_t *arrived, uint32_t n_wg) {
    const uint32_t old = atomicExch(reinterpret_cast<uint32_t *>(partial), __float_as_uint(value));
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(old) : "memory");
    return atomicAdd(arrived, 1u) == n_wg - 1u;
}

__global__ void __launch_bounds__(256)
distance_build_kernel(const float *__restrict__ xyz, int N, float inv_cell, uint32_t mask, const uint32_t *__restrict__ hdr,
                      unsigned long long *__restrict__ head, float4 *__restrict__ node, const uint32_t *__restrict__ need) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    if (need && need[0] == 0u) return;  // Verlet form: the pair lists of an earlier call still hold
    const uint32_t stamp = hdr[0] + 1u;  // never 0: a zero-filled table entry belongs to no call
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const uint32_t h = cell_hash(cell_of(x, y, z, inv_cell), mask);
    const unsigned long long old = atomicExch(&head[h], ((unsigned long long)stamp << 32) | (uint32_t)i);
    const uint32_t next = ((uint32_t)(old >> 32) == stamp && (uint32_t)old < (uint32_t)N) ? (uint32_t)old : kDistNone;
    node[i] = make_float4(x, y, z, __uint_as_float(next));
}

__global__ void __launch_bounds__(256)
distance_lists_kernel(int N, float inv_cell, float thr, uint32_t mask, uint32_t *__restrict__ hdr,
                      const unsigned long long *__restrict__ head, const float4 *__restrict__ node,
                      float *__restrict__ partial, float *__restrict__ grad, float *__restrict__ loss_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t stamp = hdr[0] + 1u;
    float acc = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    if (i < N) {
        const float4 me = node[i];
        const float px = me.x, py = me.y, pz = me.z;
        const int3 c = cell_of(px, py, pz, inv_cell);
        const float fx = px * inv_cell - (float)c.x, fy = py * inv_cell - (float)c.y, fz = pz * inv_cell - (float)c.z;
        const int sx = fx < 0.5f ? -1 : 1, sy = fy < 0.5f ? -1 : 1, sz = fz < 0.5f ? -1 : 1;
        const float thr2 = thr * thr;
        uint32_t cur[8];
        {
            unsigned long long hd[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                hd[k] = head[cell_hash(make_int3(c.x + ((k & 1) ? sx : 0), c.y + ((k & 2) ? sy : 0), c.z + ((k & 4) ? sz : 0)), mask)];
#pragma unroll
            for (int k = 0; k < 8; k++)
                cur[k] = ((uint32_t)(hd[k] >> 32) == stamp && (uint32_t)hd[k] < (uint32_t)N) ? (uint32_t)hd[k] : kDistNone;
        }
        for (int guard = 0; guard < N; guard++) {  // one node of every list per round (a list cannot be longer than N)
            bool any = false;
            float4 q[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cur[k] != kDistNone) {
                    q[k] = node[cur[k]];
                    any = true;
                }
            }
            if (!any) break;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (cur[k] == kDistNone) continue;
                const bool self = cur[k] == (uint32_t)i;
                const uint32_t nx = __float_as_uint(q[k].w);
                cur[k] = nx < (uint32_t)N ? nx : kDistNone;
                if (self) continue;
                const float ex = px - q[k].x, ey = py - q[k].y, ez = pz - q[k].z;
                const float r2 = ex * ex + ey * ey + ez * ez;
                if (!(r2 < thr2)) continue;
                const float d = sqrtf(r2);
                const float t = thr - d;
                if (!(t > 0.f)) continue;
                acc += t * t;
                if (d > 0.f) {
                    const float kk = -4.0f * t / d;
                    ax += kk * ex;
                    ay += kk * ey;
                    az += kk * ez;
                }
            }
        }
        if (grad) {
            grad[3 * i + 0] = ax;
            grad[3 * i + 1] = ay;
            grad[3 * i + 2] = az;
        }
    }
    __shared__ float s_w[4];
    __shared__ uint32_t s_last;
    acc = wave_sum63(acc);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = publish_partial_and_arrive(&partial[blockIdx.x], (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]), &hdr[1], gridDim.x) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    // the workgroup that arrives last: the loss (partial sums in workgroup order), the next call's stamp
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float tot = 0.f;
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += 256) tot += partial[b];
    tot = wave_sum63(tot);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        loss_out[0] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        hdr[1] = 0u;
        hdr[0] = stamp;  // the next call stamps its entries stamp + 1
    }
}

int fnx_distance_loss_partials(int N) { return N > 0 ? (N + 31) / 32 : 0; }

int fnx_distance_loss(const float *xyz, int N, float threshold, char *grid, float *partials, float *grad,
                      fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (N < 0 || !xyz || !grid || !partials || !(threshold > 0.f))
        return fail(FNX_ERR_INVALID_ARG, "distance_loss: bad argument");
    // table of at most 32768 buckets (one round of the single-workgroup scan); far cells that share a bucket are
    // rejected by the distance test, a bucket holds N / M points on average
    uint32_t M = 4096;
    while (M < 32768u && M < (uint32_t)N / 4u) M <<= 1;
    hipStream_t s = (hipStream_t)stream;
    GridView g = carve(grid, N, M);
    const float inv = 1.0f / (2.0f * threshold);
    hipLaunchKernelGGL(zero_u32_kernel, dim3((M + 255) / 256), dim3(256), 0, s, g.count, (size_t)M);
    hipLaunchKernelGGL(grid_count_kernel, dim3((N + 255) / 256), dim3(256), 0, s, xyz, N, inv, M - 1, g.count);
    hipLaunchKernelGGL(grid_scan_kernel, dim3(1), dim3(1024), 0, s, M, g.count, g.start, g.cursor);
    hipLaunchKernelGGL(grid_fill_kernel, dim3((N + 255) / 256), dim3(256), 0, s, xyz, N, inv, M - 1, g.start, g.cursor,
                       g.rec);
    hipLaunchKernelGGL(distance_loss_kernel, dim3((N + 31) / 32), dim3(256), 0, s, xyz, N, inv, threshold, M - 1,
                       g.start, g.rec, partials, grad);
    return hip_check("distance_loss");
}

// One wave that sleeps: delays whatever is enqueued behind it on its stream without taking more than a wave slot.
__global__ void __launch_bounds__(64)
delay_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

int fnx_stream_delay(float microseconds, fnx_stream_t stream) {
    if (!(microseconds > 0.f)) return FNX_OK;
    // wall_clock64 ticks at 100 MHz on this family
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)(microseconds * 100.0f));
    return hip_check("stream_delay");
}

size_t fnx_distance_table_bytes(int N) { return dist_table_bytes(N); }

int fnx_distance_loss_lists(const float *xyz, int N, float threshold, char *table, float *grad, float *loss_out,
                            fnx_stream_t stream) {
    if (N < 0 || !loss_out || (N > 0 && (!xyz || !table)) || !(threshold > 0.f))
        return fail(FNX_ERR_INVALID_ARG, "distance_loss_lists: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<uint32_t *>(loss_out), (size_t)1);
        return hip_check("distance_loss_lists");
    }
    DistTable t = carve_dist(table, N);
    const float inv = 1.0f / (2.0f * threshold);
    const int nb = (N + 255) / 256;
    hipLaunchKernelGGL(distance_build_kernel, dim3(nb), dim3(256), 0, s, xyz, N, inv, t.M - 1, t.hdr, t.head, t.node,
                       (const uint32_t *)nullptr);
    hipLaunchKernelGGL(distance_lists_kernel, dim3(nb), dim3(256), 0, s, N, inv, threshold, t.M - 1, t.hdr, t.head, t.node,
                       t.partial, grad, loss_out);
    return hip_check("distance_loss_lists");
}

// ---------------------------------------------------------------------------------------------
// The same loss with VERLET pair lists (round 5).  The rendered positions of consecutive optimiser steps differ by ~1e-4
// (lr), the threshold is 2e-3: the set of pairs closer than `threshold` hardly changes from call to call, yet the search
// above pays ~13 dependent random loads per point every call -- and what the branch costs the iteration is exactly that
// memory traffic beside the rasteriser's emit / blend forward.  Here a call that finds its lists still valid
// reads, per point, its <= K stored neighbour indices (coalesced) and their current positions -- no hash table at all.
//   * lists: for every point the indices of the points within R = threshold + skin of it AT BUILD TIME (ref[] holds the
//     build-time positions), K slots per point, slot-major (nbr[k N + i]).
//   * validity: a pair that is not in the lists was >= R apart at build time; while no point has moved further than
//     skin / 2 from its ref[] it is still >= threshold apart, i.e. contributes nothing.  distance_verlet_check_kernel
//     tests exactly that every call (plus: state built for this N and threshold, no list overflowed) and raises `need`;
//   * rebuild, inside the same launch sequence (no host decision, graph-capturable): distance_build_kernel fills the
//     stamped bucket lists (it returns at once when `need` is 0), and distance_verlet_kernel runs its FULL form -- the 27
//     cells around the point (cell = 2 threshold >= R), collecting the lists and evaluating the loss on its way.
// Both forms add the same terms (the pairs closer than `threshold`), in an order that depends on the form: equal within
// fp32 summation order, like the two grid forms above.  A list that overflows K leaves the state invalid: every call then
// takes the full form (correct, slow; the counters say so).
enum { VL_VALID = 0, VL_NEED = 1, VL_CALLS = 2, VL_REBUILDS = 3, VL_OVERFLOW = 4, VL_N = 5, VL_THR = 6, VL_K = 7,
       VL_ARRIVED = 8, VL_OVERFLOW_NOW = 9, VL_SKIN = 10 };
struct VerletState {
    uint32_t *hdr;
    float4 *ref;
    uint32_t *cnt;
    uint32_t *nbr;
};
inline size_t verlet_bytes(int N, int K) {
    const size_t n = (size_t)(N > 0 ? N : 0);
    size_t off = align_up(256);
    off = align_up(off + n * 16);
    off = align_up(off + n * 4);
    off = align_up(off + n * 4 * (size_t)(K > 0 ? K : 0));
    return off + kAlign;
}
inline VerletState carve_verlet(char *blob, int N, int K) {
    char *b = (char *)(((uintptr_t)blob + kAlign - 1) / kAlign * kAlign);
    const size_t n = (size_t)(N > 0 ? N : 0);
    VerletState v;
    size_t off = 0;
    v.hdr = (uint32_t *)(b + off); off = align_up(off + 256);
    v.ref = (float4 *)(b + off);   off = align_up(off + n * 16);
    v.cnt = (uint32_t *)(b + off); off = align_up(off + n * 4);
    v.nbr = (uint32_t *)(b + off);
    return v;
}

__global__ void __launch_bounds__(256)
distance_verlet_check_kernel(const float *__restrict__ xyz, int N, const float4 *__restrict__ ref, uint32_t *__restrict__ hdr,
                             float lim2, uint32_t thr_bits, uint32_t skin_bits, uint32_t K) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (i == 0) {
        hdr[VL_CALLS] += 1u;
        bad = hdr[VL_VALID] == 0u || hdr[VL_N] != (uint32_t)N || hdr[VL_THR] != thr_bits || hdr[VL_SKIN] != skin_bits ||
              hdr[VL_K] != K;
    }
    if (i < N) {
        const float4 r = ref[i];
        const float dx = xyz[3 * i] - r.x, dy = xyz[3 * i + 1] - r.y, dz = xyz[3 * i + 2] - r.z;
        bad |= !(dx * dx + dy * dy + dz * dz <= lim2);  // (a NaN position asks for a rebuild too)
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) hdr[VL_NEED] = 1u;
}

__global__ void __launch_bounds__(256)
distance_verlet_kernel(const float *__restrict__ xyz, int N, float inv_cell, float thr, float R, uint32_t mask,
                       uint32_t *__restrict__ thdr, const unsigned long long *__restrict__ head,
                       const float4 *__restrict__ node, float *__restrict__ partial, uint32_t *__restrict__ hdr,
                       float4 *__restrict__ ref, uint32_t *__restrict__ cnt, uint32_t *__restrict__ nbr, int K,
                       uint32_t thr_bits, uint32_t skin_bits, float *__restrict__ grad, float *__restrict__ loss_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool full = hdr[VL_NEED] != 0u;  // uniform over the launch: written by the check kernel in front of it
    const float thr2 = thr * thr;
    float acc = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    auto pair_term = [&](float ex, float ey, float ez, float r2) {
        if (!(r2 < thr2)) return;
        const float d = sqrtf(r2);
        const float t = thr - d;
        if (!(t > 0.f)) return;
        acc += t * t;
        if (d > 0.f) {
            const float kk = -4.0f * t / d;
            ax += kk * ex;
            ay += kk * ey;
            az += kk * ez;
        }
    };
    if (i < N) {
        if (full) {
            const uint32_t stamp = thdr[0] + 1u;
            const float4 me = node[i];
            const float px = me.x, py = me.y, pz = me.z;
            const int3 c = cell_of(px, py, pz, inv_cell);
            const float R2 = R * R;
            uint32_t found = 0;
            for (int dz = -1; dz <= 1; dz++) {  // nine bucket lists side by side per layer of cells
                uint32_t cur[9];
                {
                    unsigned long long hd[9];
#pragma unroll
                    for (int k = 0; k < 9; k++)
                        hd[k] = head[cell_hash(make_int3(c.x + (k % 3) - 1, c.y + (k / 3) - 1, c.z + dz), mask)];
#pragma unroll
                    for (int k = 0; k < 9; k++)
                        cur[k] = ((uint32_t)(hd[k] >> 32) == stamp && (uint32_t)hd[k] < (uint32_t)N) ? (uint32_t)hd[k] : kDistNone;
                }
                for (int guard = 0; guard < N; guard++) {
                    bool any = false;
                    float4 q[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (cur[k] != kDistNone) {
                            q[k] = node[cur[k]];
                            any = true;
                        }
                    }
                    if (!any) break;
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        if (cur[k] == kDistNone) continue;
                        const uint32_t j = cur[k];
                        const uint32_t nx = __float_as_uint(q[k].w);
                        cur[k] = nx < (uint32_t)N ? nx : kDistNone;
                        if (j == (uint32_t)i) continue;
                        const float ex = px - q[k].x, ey = py - q[k].y, ez = pz - q[k].z;
                        const float r2 = ex * ex + ey * ey + ez * ez;
                        if (!(r2 < R2)) continue;
                        if (found < (uint32_t)K) nbr[(size_t)found * N + i] = j;
                        found++;
                        pair_term(ex, ey, ez, r2);
                    }
                }
            }
            cnt[i] = min(found, (uint32_t)K);
            if (found > (uint32_t)K) atomicAdd(&hdr[VL_OVERFLOW_NOW], 1u);
            ref[i] = make_float4(px, py, pz, 0.f);
        } else {
            const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
            const uint32_t n = cnt[i];
            for (uint32_t k0 = 0; k0 < n; k0 += 4) {  // four neighbours' gathers in flight at a time
                uint32_t j[4];
                float qx[4], qy[4], qz[4];
#pragma unroll
                for (int k = 0; k < 4; k++) j[k] = k0 + k < n ? nbr[(size_t)(k0 + k) * N + i] : (uint32_t)i;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const size_t jj = (size_t)j[k];
                    qx[k] = xyz[3 * jj];
                    qy[k] = xyz[3 * jj + 1];
                    qz[k] = xyz[3 * jj + 2];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (j[k] == (uint32_t)i) continue;
                    const float ex = px - qx[k], ey = py - qy[k], ez = pz - qz[k];
                    pair_term(ex, ey, ez, ex * ex + ey * ey + ez * ez);
                }
            }
        }
        if (grad) {
            grad[3 * i + 0] = ax;
            grad[3 * i + 1] = ay;
            grad[3 * i + 2] = az;
        }
    }
    __shared__ float s_w[4];
    __shared__ uint32_t s_last;
    acc = wave_sum63(acc);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = publish_partial_and_arrive(&partial[blockIdx.x], (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]), &hdr[VL_ARRIVED],
                                            gridDim.x) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    // the workgroup that arrives last: the loss (partial sums in workgroup order), the state's bookkeeping
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float tot = 0.f;
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += 256) tot += partial[b];
    tot = wave_sum63(tot);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        loss_out[0] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        hdr[VL_ARRIVED] = 0u;
        if (full) {
            const uint32_t over = __hip_atomic_load(&hdr[VL_OVERFLOW_NOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hdr[VL_OVERFLOW] = over;
            hdr[VL_OVERFLOW_NOW] = 0u;
            hdr[VL_VALID] = over == 0u ? 1u : 0u;
            hdr[VL_N] = (uint32_t)N;
            hdr[VL_THR] = thr_bits;
            hdr[VL_SKIN] = skin_bits;
            hdr[VL_K] = (uint32_t)K;
            hdr[VL_REBUILDS] += 1u;
            thdr[0] = thdr[0] + 1u;  // the bucket table: the next build stamps its entries one higher
        }
        hdr[VL_NEED] = 0u;
    }
}

size_t fnx_distance_verlet_bytes(int N, int K) { return verlet_bytes(N, K); }

int fnx_distance_loss_verlet(const float *xyz, int N, float threshold, float skin, char *table, char *state, int K,
                             float *grad, float *loss_out, fnx_stream_t stream) {
    if (N < 0 || !loss_out || (N > 0 && (!xyz || !table || !state)) || !(threshold > 0.f) || !(skin > 0.f) ||
        skin > threshold || K < 1 || K > 64)
        return fail(FNX_ERR_INVALID_ARG, "distance_loss_verlet: bad argument (0 < skin <= threshold, 1 <= K <= 64)");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<uint32_t *>(loss_out), (size_t)1);
        return hip_check("distance_loss_verlet");
    }
    DistTable t = carve_dist(table, N);
    VerletState v = carve_verlet(state, N, K);
    const float inv = 1.0f / (2.0f * threshold);  // cell = 2 threshold >= R: the 27 cells around a point hold its R-ball
    const int nb = (N + 255) / 256;
    uint32_t thr_bits, skin_bits;
    std::memcpy(&thr_bits, &threshold, 4);
    std::memcpy(&skin_bits, &skin, 4);
    const float lim = 0.5f * skin;
    hipLaunchKernelGGL(distance_verlet_check_kernel, dim3(nb), dim3(256), 0, s, xyz, N, v.ref, v.hdr, lim * lim, thr_bits,
                       skin_bits, (uint32_t)K);
    hipLaunchKernelGGL(distance_build_kernel, dim3(nb), dim3(256), 0, s, xyz, N, inv, t.M - 1, t.hdr, t.head, t.node,
                       (const uint32_t *)(v.hdr + VL_NEED));
    hipLaunchKernelGGL(distance_verlet_kernel, dim3(nb), dim3(256), 0, s, xyz, N, inv, threshold, threshold + skin, t.M - 1,
                       t.hdr, t.head, t.node, t.partial, v.hdr, v.ref, v.cnt, v.nbr, K, thr_bits, skin_bits, grad, loss_out);
    return hip_check("distance_loss_verlet");
}

int fnx_adam_step(float *x, int n, const float *g0, float s0, const float *g1, float s1, const float *g2, float s2,
                  float inv_batch, float *exp_avg, float *exp_avg_sq, float *step, float lr, double beta1_d,
                  double beta2_d, float eps, float *grad_out, float *scaled_out, float scale, unsigned int *arrived,
                  fnx_stream_t stream) {
    const float beta1 = (float)beta1_d, beta2 = (float)beta2_d;
    if (n < 0 || (n > 0 && (!x || !exp_avg || !exp_avg_sq)) || !step || !arrived || !(g0 || g1 || g2))
        return fail(FNX_ERR_INVALID_ARG, "adam_step: bad argument");
    hipStream_t s = (hipStream_t)stream;
    // n == 0 still advances the step count: one (empty) workgroup
    hipLaunchKernelGGL(adam_step_kernel, dim3(n > 0 ? (n + 255) / 256 : 1), dim3(256), 0, s, x, n, g0, s0, g1, s1, g2, s2,
                       inv_batch, exp_avg, exp_avg_sq, step, lr, beta1, beta2, (float)(1.0 - (double)beta1_d),
                       (float)(1.0 - (double)beta2_d), eps, grad_out, scaled_out, scale, arrived);
    return hip_check("adam_step");
}

int fnx_adam_step_grid(float *x, int N, const float *g0, float s0, const float *g1, float s1, const float *g2, float s2,
                       float inv_batch, float *exp_avg, float *exp_avg_sq, float *step, float lr, double beta1_d,
                       double beta2_d, float eps, float *grad_out, float *scaled_out, float scale, unsigned int *arrived,
                       float cell, char *grid, const float *prev, float secs, fnx_stream_t stream) {
    const float beta1 = (float)beta1_d, beta2 = (float)beta2_d;
    if (N <= 0 || !x || !exp_avg || !exp_avg_sq || !step || !arrived || !(g0 || g1 || g2) || !scaled_out || !grid ||
        cell <= 0.f || (prev && !(secs != 0.f)))
        return fail(FNX_ERR_INVALID_ARG, "adam_step_grid: bad argument");
    hipStream_t s = (hipStream_t)stream;
    GridView g = carve(grid, N);
    const float inv = 1.0f / cell;
    hipLaunchKernelGGL(adam_count_scan_kernel, dim3((N + 1023) / 1024), dim3(1024), 0, s, x, N, g0, s0, g1, s1, g2, s2,
                       inv_batch, exp_avg, exp_avg_sq, step, lr, beta1, beta2, (float)(1.0 - (double)beta1_d),
                       (float)(1.0 - (double)beta2_d), eps, grad_out, scaled_out, scale, arrived, inv, g.M, g.count,
                       g.start, g.cursor);
    hipLaunchKernelGGL(grid_fill_velocity_kernel, dim3((N + 255) / 256), dim3(256), 0, s, scaled_out, N, inv, g.M, g.start,
                       g.cursor, g.rec, prev, secs, g.aux0, g.count);
    return hip_check("adam_step_grid");
}

int fnx_visual_interp_forward(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                              float H, float secs, float eps, const char *hidden_grid, float *out, float *sum_w,
                              float *wvel, fnx_stream_t stream) {
    if (V == 0) return FNX_OK;
    if (V < 0 || N < 0 || !visual || !hidden_grid || !out || !sum_w || !wvel || (N > 0 && (!hidden || !hidden_prev)))
        return fail(FNX_ERR_INVALID_ARG, "visual_interp_forward: bad argument");
    GridView g = carve(const_cast<char *>(hidden_grid), N);
    if (N > 0)
        hipLaunchKernelGGL(slot_velocity_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, g.rec, N,
                           hidden_prev, secs, g.aux0);
    hipLaunchKernelGGL(visual_forward_kernel, dim3((V + 31) / 32), dim3(256), 0, (hipStream_t)stream, visual, V,
                       1.0f / H, H * H, poly6_term1(H), secs, eps, g.M - 1, g.start, g.rec, g.aux0, out, sum_w, wvel);
    return hip_check("visual_interp_forward");
}

int fnx_visual_interp_forward_kcap(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                   float H, float secs, float eps, const char *hidden_grid, const uint32_t *cutv,
                                   float *out, float *sum_w, float *wvel, fnx_stream_t stream) {
    if (V == 0) return FNX_OK;
    if (V < 0 || N < 0 || !visual || !hidden_grid || !cutv || !out || !sum_w || !wvel ||
        (N > 0 && (!hidden || !hidden_prev)))
        return fail(FNX_ERR_INVALID_ARG, "visual_interp_forward_kcap: bad argument");
    GridView g = carve(const_cast<char *>(hidden_grid), N);
    if (N > 0)
        hipLaunchKernelGGL(slot_velocity_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, g.rec, N,
                           hidden_prev, secs, g.aux0);
    hipLaunchKernelGGL(visual_forward_kcap_kernel, dim3((V + 31) / 32), dim3(256), 0, (hipStream_t)stream, visual, V,
                       1.0f / H, H * H, poly6_term1(H), secs, eps, g.M - 1, g.start, g.rec, g.aux0, cutv, out, sum_w,
                       wvel);
    return hip_check("visual_interp_forward_kcap");
}

int fnx_visual_interp_backward_kcap(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                    float H, float secs, float eps, const char *visual_grid, const uint32_t *cutv,
                                    const float *sum_w, const float *wvel, const float *dL_dout, float *dL_dhidden,
                                    fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (V < 0 || N < 0 || !hidden || !hidden_prev || !visual_grid || !dL_dhidden ||
        (V > 0 && (!visual || !cutv || !sum_w || !wvel || !dL_dout)))
        return fail(FNX_ERR_INVALID_ARG, "visual_interp_backward_kcap: bad argument");
    GridView g = carve(const_cast<char *>(visual_grid), V);
    if (V > 0)
        hipLaunchKernelGGL(slot_visual_payload_kernel, dim3((V + 255) / 256), dim3(256), 0, (hipStream_t)stream, g.rec,
                           V, sum_w, wvel, dL_dout, (const float *)nullptr, 0.0f, secs, eps, g.aux0);
    hipLaunchKernelGGL(visual_backward_kcap_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, hidden,
                       hidden_prev, N, 1.0f / H, H * H, poly6_term1(H), secs, g.M - 1, g.start, g.rec, g.aux0, cutv,
                       dL_dhidden);
    return hip_check("visual_interp_backward_kcap");
}

size_t fnx_grid_cell_items_bytes(int N) { return 64 + ((size_t)(N > 0 ? N : 0) + (size_t)(N > 0 ? N : 0) / 64 + 1) * 8; }

int fnx_grid_cell_items(const char *grid, int N, char *items, fnx_stream_t stream) {
    if (N < 0 || !grid || !items) return fail(FNX_ERR_INVALID_ARG, "grid_cell_items: bad argument");
    GridView g = carve(const_cast<char *>(grid), N);
    uint32_t *n_items = (uint32_t *)items;
    hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n_items, (size_t)16);
    if (N > 0)
        hipLaunchKernelGGL(grid_cell_items_kernel, dim3((g.M + 255) / 256), dim3(256), 0, (hipStream_t)stream, g.M,
                           g.start, (uint2 *)(items + 64), n_items);
    return hip_check("grid_cell_items");
}

int fnx_visual_interp_forward_cells(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                    float H, float secs, float eps, const char *hidden_grid, const char *visual_grid,
                                    const char *visual_items, float *out, float *sum_w, float *wvel,
                                    fnx_stream_t stream) {
    return fnx_visual_interp_forward_cells_div(visual, V, hidden, hidden_prev, N, H, secs, eps, hidden_grid, visual_grid,
                                               visual_items, out, sum_w, wvel, nullptr, 1.0f, stream);
}

int fnx_visual_interp_forward_cells_div(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                        float H, float secs, float eps, const char *hidden_grid, const char *visual_grid,
                                        const char *visual_items, float *out, float *sum_w, float *wvel, float *out_div,
                                        float divisor, fnx_stream_t stream) {
    return fnx_visual_interp_forward_cells_vel(visual, V, hidden, hidden_prev, N, H, secs, eps, hidden_grid, visual_grid,
                                               visual_items, out, sum_w, wvel, out_div, divisor, 0, stream);
}

int fnx_visual_interp_forward_cells_vel(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                        float H, float secs, float eps, const char *hidden_grid, const char *visual_grid,
                                        const char *visual_items, float *out, float *sum_w, float *wvel, float *out_div,
                                        float divisor, int velocity_ready, fnx_stream_t stream) {
    if (V == 0) return FNX_OK;
    if (V < 0 || N < 0 || !visual || !hidden_grid || !visual_grid || !visual_items || !out || !sum_w || !wvel ||
        (N > 0 && (!hidden || !hidden_prev)))
        return fail(FNX_ERR_INVALID_ARG, "visual_interp_forward_cells: bad argument");
    GridView g = carve(const_cast<char *>(hidden_grid), N);
    GridView gv = carve(const_cast<char *>(visual_grid), V);
    if (N > 0 && !velocity_ready)
        hipLaunchKernelGGL(slot_velocity_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, g.rec, N,
                           hidden_prev, secs, g.aux0);
    // one-wave workgroups that stride over the items; 20 of them fit a CU (8 KiB of LDS each)
    const size_t bound = (size_t)V + (size_t)V / 64 + 1;
    const unsigned wgs = (unsigned)(bound < 5120 ? bound : 5120);
    if (t_knn_flags)
        hipLaunchKernelGGL(visual_forward_cells_kernel<true>, dim3(wgs), dim3(64), 0, (hipStream_t)stream, V, 1.0f / H, H * H,
                           poly6_term1(H), secs, eps, gv.rec, (const uint2 *)(visual_items + 64),
                           (const uint32_t *)visual_items, g.M - 1, g.start, g.rec, g.aux0, out, sum_w, wvel, out_div, divisor,
                           t_knn_flags, (float)t_knn_k);
    else
        hipLaunchKernelGGL(visual_forward_cells_kernel<false>, dim3(wgs), dim3(64), 0, (hipStream_t)stream, V, 1.0f / H, H * H,
                           poly6_term1(H), secs, eps, gv.rec, (const uint2 *)(visual_items + 64),
                           (const uint32_t *)visual_items, g.M - 1, g.start, g.rec, g.aux0, out, sum_w, wvel, out_div, divisor,
                           (uint32_t *)nullptr, 0.f);
    return hip_check("visual_interp_forward_cells");
}

int fnx_visual_interp_backward(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                               float H, float secs, float eps, const char *visual_grid, const float *sum_w,
                               const float *wvel, const float *dL_dout, float *dL_dhidden, fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (V < 0 || N < 0 || !hidden || !hidden_prev || !visual_grid || !dL_dhidden ||
        (V > 0 && (!visual || !sum_w || !wvel || !dL_dout)))
        return fail(FNX_ERR_INVALID_ARG, "visual_interp_backward: bad argument");
    GridView g = carve(const_cast<char *>(visual_grid), V);
    if (V > 0)
        hipLaunchKernelGGL(slot_visual_payload_kernel, dim3((V + 255) / 256), dim3(256), 0, (hipStream_t)stream, g.rec,
                           V, sum_w, wvel, dL_dout, (const float *)nullptr, 0.0f, secs, eps, g.aux0);
    hipLaunchKernelGGL(visual_backward_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, hidden,
                       hidden_prev, N, 1.0f / H, H, H * H, poly6_term1(H), secs, g.M - 1, g.start, g.rec, g.aux0,
                       dL_dhidden);
    return hip_check("visual_interp_backward");
}

int fnx_visual_interp_backward_cells(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                     float H, float secs, float eps, const char *visual_grid, const char *hidden_grid,
                                     const char *hidden_items, const float *sum_w, const float *wvel,
                                     const float *dL_dout, float *dL_dhidden, fnx_stream_t stream) {
    return fnx_visual_interp_backward_cells_sum(visual, V, hidden, hidden_prev, N, H, secs, eps, visual_grid, hidden_grid,
                                                hidden_items, sum_w, wvel, dL_dout, nullptr, 0.0f, dL_dhidden, stream);
}

int fnx_visual_interp_backward_cells_sum(const float *visual, int V, const float *hidden, const float *hidden_prev, int N,
                                         float H, float secs, float eps, const char *visual_grid,
                                         const char *hidden_grid, const char *hidden_items, const float *sum_w,
                                         const float *wvel, const float *dL_dout, const float *dL_dout2, float scale2,
                                         float *dL_dhidden, fnx_stream_t stream) {
    if (N == 0) return FNX_OK;
    if (V < 0 || N < 0 || !hidden || !hidden_prev || !visual_grid || !hidden_grid || !hidden_items || !dL_dhidden ||
        (V > 0 && (!visual || !sum_w || !wvel || !dL_dout)))
        return fail(FNX_ERR_INVALID_ARG, "visual_interp_backward_cells: bad argument");
    GridView g = carve(const_cast<char *>(visual_grid), V);
    GridView gh = carve(const_cast<char *>(hidden_grid), N);
    if (V > 0)
        hipLaunchKernelGGL(slot_visual_payload_kernel, dim3((V + 255) / 256), dim3(256), 0, (hipStream_t)stream, g.rec,
                           V, sum_w, wvel, dL_dout, dL_dout2, scale2, secs, eps, g.aux0);
    // workgroups stride over the items (their number is only known on the device)
    const size_t bound = (size_t)N + (size_t)N / 64 + 1;
    const unsigned wgs = (unsigned)(bound < 4096 ? bound : 4096);
    hipLaunchKernelGGL(visual_backward_cells_kernel, dim3(wgs), dim3(64 * kBwdWaves), 0, (hipStream_t)stream, 1.0f / H, H * H,
                       poly6_term1(H), secs, gh.rec, (const uint2 *)(hidden_items + 64), (const uint32_t *)hidden_items,
                       hidden_prev, g.M - 1, g.start, g.rec, g.aux0, dL_dhidden);
    return hip_check("visual_interp_backward_cells");
}

}  // extern "C"
