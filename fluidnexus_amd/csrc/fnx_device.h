// Device-side helpers shared by the gfx950 rasteriser kernels.
//
// Numerics contract (DESIGN.md "Numerics"): this library is compiled with -ffp-contract=off, so
// every fp32 expression below is evaluated exactly as written (IEEE add/mul/div/sqrt); fused
// multiply-adds appear only where spelled __builtin_fmaf.  That makes the per-Gaussian stage and
// the forward blend bit-reproducible against a scalar CPU evaluation of the same expressions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#define FNX_TILE_X 16  // ch3/cuda_rasterizer/config.h:16
#define FNX_TILE_Y 16  // ch3/cuda_rasterizer/config.h:17
#define FNX_TILE_PIX (FNX_TILE_X * FNX_TILE_Y)

namespace fnx {

// Slice of a per-view array: `stride_bytes` between consecutive views (see ViewBatch in fnx_state.h).
// (pointer arithmetic on the pointer itself, not through an integer: a pointer rebuilt from a uintptr_t loses its
// address space and every access through it becomes a FLAT instruction, which counts against BOTH vmcnt and lgkmcnt --
// each LDS wait then also drains the global prefetches in flight)
template <class T>
__device__ __forceinline__ T *view_at(T *p, size_t stride_bytes, int v) {
    using Byte = typename std::conditional<std::is_const<T>::value, const char, char>::type;
    return reinterpret_cast<T *>(reinterpret_cast<Byte *>(p) + stride_bytes * (size_t)v);
}

// Workgroup barrier that orders LDS traffic only: waits for this wave's outstanding LDS operations, not for
// its global stores (a plain __syncthreads() also drains those, i.e. an L2 round trip per barrier).
// Only for phases whose global stores are not read again by the workgroup.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 3x3 matrix stored as columns (c0,c1,c2), element m[c][r]; the product keeps the k = 0,1,2
// left-to-right summation order of the maths library the reference uses (glm mat3 operator*).
struct M3 {
    float m[3][3];
};
__device__ __forceinline__ M3 m3_mul(const M3 &A, const M3 &B) {
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
    return R;
}
__device__ __forceinline__ M3 m3_t(const M3 &A) {
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
    return R;
}
__device__ __forceinline__ M3 m3_cols(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1,
                                      float c2) {
    M3 R;
    R.m[0][0] = a0; R.m[0][1] = a1; R.m[0][2] = a2;
    R.m[1][0] = b0; R.m[1][1] = b1; R.m[1][2] = b2;
    R.m[2][0] = c0; R.m[2][1] = c1; R.m[2][2] = c2;
    return R;
}

// Row-vector 4x4 transforms (matrices arrive transposed: SURVEY A.1; ch3 auxiliary.h:54-71).
__device__ __forceinline__ float3 xform4x3(const float3 p, const float *m) {
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(const float3 p, const float *m) {
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

// NDC -> pixel centre, evaluated in fp64 then rounded (ch3 auxiliary.h:41-43 uses double literals).
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// Tile rectangle of a splat (ch3 auxiliary.h:45-52): float divide, truncation toward zero, clamp.
__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int &x0, int &y0, int &x1,
                                          int &y1) {
    x0 = min(gx, max(0, (int)((px - radius) / FNX_TILE_X)));
    y0 = min(gy, max(0, (int)((py - radius) / FNX_TILE_Y)));
    x1 = min(gx, max(0, (int)((px + radius + FNX_TILE_X - 1) / FNX_TILE_X)));
    y1 = min(gy, max(0, (int)((py + radius + FNX_TILE_Y - 1) / FNX_TILE_Y)));
}

// exp() with a fixed instruction sequence (Cody-Waite reduction + degree-6 polynomial, <= 1 ulp on
// [-87, 0]) so the blend is reproducible on any IEEE machine; the parity oracle evaluates the same
// sequence on the CPU.  v_rndne_f32 + 8 v_fma_f32 + a few VALU ops.
__device__ __forceinline__ float exp_fixed(float x) {
    // branch-free (selects only) so that several evaluations can be interleaved by the scheduler;
    // the arithmetic on [-87, 88] is the fixed sequence, x < -87 gives exactly 0, x > 88 saturates
    const float xc = fminf(x, 88.0f);
    const float t = xc * 1.44269504088896341f;
    const float n = __builtin_rintf(t);
    float r = __builtin_fmaf(n, -0.693145751953125f, xc);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = __builtin_fmaf(p, r2, r) + 1.0f;
    const int ni = (int)n;
    const float v = y * __uint_as_float((uint32_t)(ni + 127) << 23);
    return (x < -87.0f) ? 0.0f : v;
}

// The same sequence without the two range guards, for callers that only USE the result when -87 <= x <= 0 (the blend
// kernels discard every lane whose power is positive or below the splat's threshold >= -ln(255) - margin): on that
// range it returns exactly exp_fixed(x); outside it returns garbage that must not be consumed.
__device__ __forceinline__ float exp_fixed_in_range(float x) {
    const float t = x * 1.44269504088896341f;
    const float n = __builtin_rintf(t);
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = __builtin_fmaf(p, r2, r) + 1.0f;
    return y * __uint_as_float((uint32_t)((int)n + 127) << 23);
}

// Footprint of a splat for culling, computed once per splat in preprocess.  A pixel can receive a
// contribution (alpha >= 1/255) only if power >= -ln(255 o); that ellipse's axis-aligned bounding
// box has half extents sqrt(2 tau c / det), sqrt(2 tau a / det) (conic (a, b, c), det = ac - b^2).
// The extents are inflated by 1e-3 relative + 0.01 px and tau by 1e-4 relative + 1e-3, orders of
// magnitude above fp32 evaluation error, so culled (pixel, splat) pairs are exactly pairs the
// un-culled loop would have skipped.  Outputs: thr = power threshold below which exp() need not be
// evaluated; (ex, ey) half extents, ex < 0 meaning "never contributes" (o < 1/255: alpha = min(.99,
// o G) with G <= 1 stays below 1/255), +huge meaning "cannot be culled".
__device__ __forceinline__ void splat_footprint(float a, float b, float c, float o, float &thr, float &ex, float &ey) {
    thr = -87.0f;  // below -87 the fixed exp is exactly 0, i.e. alpha < 1/255: never a contribution
    ex = ey = 3.0e38f;
    if (o < 1.0f / 255.0f) {
        ex = ey = -1.0f;
        return;
    }
    const float tau = __logf(255.0f * o) * 1.0001f + 1.0e-3f;
    if (!(tau > 0.0f)) return;
    thr = fmaxf(-tau, -87.0f);
    const float det = a * c - b * b;
    if (!(det > 0.0f)) return;
    const float k = 2.0f * tau / det;
    const float fx = sqrtf(k * c) * 1.001f + 0.01f, fy = sqrtf(k * a) * 1.001f + 0.01f;
    if (!(fx >= 0.0f) || !(fy >= 0.0f)) return;  // NaN guard
    ex = fx;
    ey = fy;
}

// Pixel of its 16x16 tile that lane `lane` of wave `w` of a blend workgroup (forward and backward) owns: a wave is an
// 8x8 quadrant (w & 1, w >> 1), a row of 16 lanes one 4x4 block of it (block_mask_exact's bit order), lane & 15 the
// pixel of the block, row-major.
__device__ __forceinline__ int blend_pixel_x(int w, int lane) { return (w & 1) * 8 + ((lane >> 4) & 1) * 4 + (lane & 3); }
__device__ __forceinline__ int blend_pixel_y(int w, int lane) { return (w >> 1) * 8 + (lane >> 5) * 4 + ((lane >> 2) & 3); }

// 4x4-pixel blocks of the 16x16 tile at (tile_x0, tile_y0) that can hold a contributing pixel of the splat.  A pixel
// can contribute only where power = -q/2 >= thr (thr = -ln(255 o) with margins, see splat_footprint), so a block over
// which the minimum of the conic's quadratic form q exceeds -2 thr (plus a further 0.1 % + 0.01, far above the fp32
// error of the evaluation near the threshold) holds no contributing pixel.  Bit 4 q + b <-> block (b & 1, b >> 1) of
// quadrant q = (q & 1, q >> 1), i.e. the block whose first pixel is (8 (q & 1) + 4 (b & 1), 8 (q >> 1) + 4 (b >> 1)).
__device__ __forceinline__ uint32_t block_mask_exact(float x, float y, float a, float b, float c, float thr, float ex,
                                                     float ey, float tile_x0, float tile_y0) {
    if (ex < 0.0f) return 0u;
    // The minimum of q(dx, dy) = a dx^2 + 2 b dx dy + c dy^2 over a block is 0 if the centre lies in it; otherwise it
    // lies on an edge that FACES the centre (q is convex with its minimum at the centre: from any other point of the
    // block a step towards the centre stays inside and lowers q).  Per column of blocks that is the vertical line
    // nearer to the centre, per row the nearer horizontal line (for a column that straddles the centre the nearer
    // line is no facing edge, but a point of the block all the same: it cannot undercut the minimum).  So a block
    // costs two 1-D quadratics with clamped minimisers: a clamp (v_med3), two fused multiply-adds and a min each.
    // Culling arithmetic is free to fuse: its margins dwarf the rounding.
    const float lim = (-2.0f * thr) * 1.001f + 0.01f;
    const float nbc = -b / c, nba = -b / a, b2 = 2.0f * b;
    const float ox = tile_x0 - x, oy = tile_y0 - y;
    // in_x / in_y: 0 if the column / row of blocks straddles the centre, else huge; their sum is the "quadratic" of a
    // block that holds the centre (0: kept, lim > 0) without a branch
    float X0[4], X1[4], aXX[4], bX[4], sX[4], in_x[4], Y0[4], Y1[4], cYY[4], bY[4], sY[4], in_y[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {  // block column / row i: its first and last pixel line, and the one nearer the centre
        X0[i] = ox + (float)(4 * i);
        X1[i] = ox + (float)(4 * i + 3);
        in_x[i] = fmaxf(X0[i], -X1[i]) <= 0.0f ? 0.0f : 3.0e38f;
        const float LX = X0[i] > 0.0f ? X0[i] : X1[i];  // right of the centre: the left line; else the right line
        aXX[i] = a * LX * LX;
        bX[i] = b2 * LX;
        sX[i] = nbc * LX;  // minimiser in dy along the vertical line
        Y0[i] = oy + (float)(4 * i);
        Y1[i] = oy + (float)(4 * i + 3);
        in_y[i] = fmaxf(Y0[i], -Y1[i]) <= 0.0f ? 0.0f : 3.0e38f;
        const float LY = Y0[i] > 0.0f ? Y0[i] : Y1[i];
        cYY[i] = c * LY * LY;
        bY[i] = b2 * LY;
        sY[i] = nba * LY;  // minimiser in dx along the horizontal line
    }
    uint32_t m = 0u;
#pragma unroll
    for (int cy = 0; cy < 4; cy++) {
#pragma unroll
        for (int cx = 0; cx < 4; cx++) {
            const float dy = __builtin_amdgcn_fmed3f(sX[cx], Y0[cy], Y1[cy]);
            const float qv = __builtin_fmaf(dy, __builtin_fmaf(c, dy, bX[cx]), aXX[cx]);
            const float dx = __builtin_amdgcn_fmed3f(sY[cy], X0[cx], X1[cx]);
            const float qh = __builtin_fmaf(dx, __builtin_fmaf(a, dx, bY[cy]), cYY[cy]);
            const float q = fminf(fminf(qv, qh), in_x[cx] + in_y[cy]);
            m |= (q > lim) ? 0u : 1u << (4 * ((cy >> 1) * 2 + (cx >> 1)) + (cy & 1) * 2 + (cx & 1));
        }
    }
    return m;
}

// Real-SH basis constants (ch3 auxiliary.h:22-39).
__device__ static const float kSH0 = 0.28209479177387814f;
__device__ static const float kSH1 = 0.4886025119029199f;
__device__ static const float kSH2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                         -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float kSH3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                         0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                         -0.5900435899266435f};

}  // namespace fnx
