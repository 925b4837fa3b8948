/*
 * raster_oracle.c -- TEST INFRASTRUCTURE ONLY (the parity checker, never the product).
 *
 * A scalar, single-threaded, fp32 CPU restatement of the tile-based differentiable
 * 3D-Gaussian rasteriser that FluidNexus vendors in
 *   FluidDynamics/submodules/gaussian_rasterization_ch{3,1}/cuda_rasterizer/
 * (ch1 and ch3 differ only in NUM_CHANNELS, config.h:15).  Every function cites the
 * reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 *
 * PARITY STATUS: "parity unpinned" for this file.  The reference's rasteriser is CUDA-only
 * (needs nvcc, cub, cooperative_groups and a GPU), cannot be built or run in the build
 * container, and the reference ships no tests, golden vectors or fixtures for it.  The
 * restatement is pinned only by (i) line-by-line review against the .cu sources, (ii)
 * closed-form / structural known-answer tests, (iii) an independent fp64 torch-autograd
 * formulation of the same maths (oracle/torch_ref.py) that checks the hand-written backward,
 * and (iv) golden vectors from the importable Python parts of the reference (SH, cameras).
 *
 * Numerics contract (so that the HIP product path can be BIT-EXACT against this file):
 *  - built with -ffp-contract=off: every expression is evaluated as written, no FMA fusion
 *    (nvcc's own contraction choices are not reproducible without nvcc; "as written" is the
 *    neutral reading of the source).  fmaf() appears only inside fnx_oracle_expf.
 *  - sqrtf, division: IEEE correctly rounded (as CUDA without -use_fast_math, ch3/setup.py:9-28).
 *  - expf: CUDA's libdevice expf is not reproducible off-NVIDIA; both this oracle and the HIP
 *    kernels use the fixed Cody-Waite + degree-6 polynomial below (<= 1 ulp measured against
 *    double exp on [-87, 0]; tests/test_oracle_kat.py checks it).
 *  - backward sums over pixels: the reference accumulates with fp32 atomicAdd in arbitrary
 *    order (backward.cu:503,524-533; run-to-run nondeterministic).  The oracle accumulates the
 *    same fp32 addends in double and rounds once: the value every ordering approximates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16 /* config.h:16 */
#define BLOCK_Y 16 /* config.h:17 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)

/* auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* glm-style column-major 3x3: m[col][row] (glm::mat3 constructor fills columns). */
typedef struct {
    float m[3][3];
} mat3;

/* glm detail/type_mat3x3.inl:486-519: Result[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2] */
static mat3 mat3_mul(mat3 A, mat3 B) {
    mat3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
    return R;
}
static mat3 mat3_transpose(mat3 A) {
    mat3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
    return R;
}
static mat3 mat3_cols(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) {
    mat3 R = {{{a0, a1, a2}, {b0, b1, b2}, {c0, c1, c2}}};
    return R;
}

/* Fixed-algorithm expf shared (by construction, not by source) with the HIP kernels. */
float fnx_oracle_expf(float x) {
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float t = x * 1.44269504088896341f;
    float n = rintf(t); /* round-half-even, == v_rndne_f32 */
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    int32_t ni = (int32_t)n; /* in [-126, 127] */
    union {
        uint32_t u;
        float f;
    } s;
    s.u = (uint32_t)(ni + 127) << 23;
    return y * s.f;
}

/* Number of OpenMP threads the pixel/Gaussian loops may use (1 = scalar port). */
void fnx_oracle_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
int fnx_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* auxiliary.h:41-43: double literals => evaluated in fp64, rounded to fp32 on return */
static float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:45-52 */
static void getRect(float px, float py, int max_radius, int gx, int gy, uint32_t *rmin, uint32_t *rmax) {
    rmin[0] = (uint32_t)imin(gx, imax(0, (int)((px - max_radius) / BLOCK_X)));
    rmin[1] = (uint32_t)imin(gy, imax(0, (int)((py - max_radius) / BLOCK_Y)));
    rmax[0] = (uint32_t)imin(gx, imax(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rmax[1] = (uint32_t)imin(gy, imax(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* auxiliary.h:54-71 */
static void transformPoint4x3(const float *p, const float *m, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void transformPoint4x4(const float *p, const float *m, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* auxiliary.h:124-147: the local edit culls only on view-space z <= 0.2 */
static int in_frustum(const float *p_orig, const float *viewmatrix, float *p_view) {
    transformPoint4x3(p_orig, viewmatrix, p_view);
    return !(p_view[2] <= 0.2f);
}

/* forward.cu:20-67 */
static void computeColorFromSH(int idx, int deg, int max_coeffs, const float *means, const float *campos,
                               const float *shs, uint8_t *clamped, float *out) {
    float dir[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] = dir[0] / len;
    dir[1] = dir[1] / len;
    dir[2] = dir[2] / len;
    const float *sh = shs + (size_t)idx * max_coeffs * 3;
    float x = dir[0], y = dir[1], z = dir[2];
    for (int c = 0; c < 3; c++) {
#define SH(k) sh[(k) * 3 + c]
        float result = SH_C0 * SH(0);
        if (deg > 0) {
            result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                result = result + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) +
                         SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + SH_C2[3] * xz * SH(7) +
                         SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    result = result + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                             SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                             SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                             SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                             SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        result += 0.5f;
        clamped[3 * idx + c] = (result < 0);
        out[c] = result > 0.0f ? result : 0.0f; /* glm::max(result, 0.0f) */
    }
}

/* forward.cu:113-145 (quaternion NOT normalised: local edit at :121) */
static void computeCov3D(const float *scale, float mod, const float *rot, float *cov3D) {
    mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * scale[0];
    S.m[1][1] = mod * scale[1];
    S.m[2][2] = mod * scale[2];
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R = mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                       2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                       2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    mat3 M = mat3_mul(S, R);
    mat3 Sigma = mat3_mul(mat3_transpose(M), M);
    cov3D[0] = Sigma.m[0][0];
    cov3D[1] = Sigma.m[0][1];
    cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1];
    cov3D[4] = Sigma.m[1][2];
    cov3D[5] = Sigma.m[2][2];
}

/* forward.cu:70-108 */
static void computeCov2D(const float *mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                         const float *cov3D, const float *viewmatrix, float *cov) {
    float t[3];
    transformPoint4x3(mean, viewmatrix, t);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t[0] / t[2];
    const float tytz = t[1] / t[2];
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    mat3 J = mat3_cols(focal_x / t[2], 0.0f, -(focal_x * t[0]) / (t[2] * t[2]), 0.0f, focal_y / t[2],
                       -(focal_y * t[1]) / (t[2] * t[2]), 0, 0, 0);
    mat3 Wm = mat3_cols(viewmatrix[0], viewmatrix[4], viewmatrix[8], viewmatrix[1], viewmatrix[5], viewmatrix[9],
                        viewmatrix[2], viewmatrix[6], viewmatrix[10]);
    mat3 T = mat3_mul(Wm, J);
    mat3 Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    mat3 c2 = mat3_mul(mat3_mul(mat3_transpose(T), mat3_transpose(Vrk)), T);
    c2.m[0][0] += 0.3f;
    c2.m[1][1] += 0.3f;
    cov[0] = c2.m[0][0];
    cov[1] = c2.m[0][1];
    cov[2] = c2.m[1][1];
}

/*
 * Stage 1: preprocessCUDA<C> (forward.cu:148-244) + InclusiveSum (rasterizer_impl.cu:259).
 * All per-Gaussian output arrays are caller-allocated and must be zero-initialised (the
 * reference leaves culled entries uninitialised; zero is this oracle's convention).
 * Pass NULL for absent optional inputs (the reference's nullptr, rasterize_points.cu:95-101).
 * Returns num_rendered = point_offsets[P-1].
 */
int64_t fnx_oracle_preprocess(int C, int P, int D, int M, const float *means3D, const float *scales,
                              float scale_modifier, const float *rotations, const float *opacities, const float *shs,
                              const float *cov3D_precomp, const float *colors_precomp, const float *viewmatrix,
                              const float *projmatrix, const float *cam_pos, int W, int H, float tan_fovx,
                              float tan_fovy, int32_t *radii, float *means2D, float *depths, float *cov3Ds, float *rgb,
                              float *conic_opacity, uint8_t *clamped, uint32_t *tiles_touched,
                              uint32_t *point_offsets) {
    /* rasterizer_impl.cu:207-208 */
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        const float *p_orig = means3D + 3 * idx;
        float p_view[3];
        if (!in_frustum(p_orig, viewmatrix, p_view)) continue;
        float p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
        const float *cov3D;
        if (cov3D_precomp != NULL) {
            cov3D = cov3D_precomp + idx * 6;
        } else {
            computeCov3D(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + idx * 6);
            cov3D = cov3Ds + idx * 6;
        }
        float cov[3];
        computeCov2D(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, cov);
        float det = (cov[0] * cov[2] - cov[1] * cov[1]);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
        float mid = 0.5f * (cov[0] + cov[2]);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float px = ndc2Pix(p_proj[0], W), py = ndc2Pix(p_proj[1], H);
        uint32_t rmin[2], rmax[2];
        getRect(px, py, (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (colors_precomp == NULL) {
            float res[3];
            computeColorFromSH(idx, D, M, means3D, cam_pos, shs, clamped, res);
            rgb[idx * C + 0] = res[0];
            rgb[idx * C + 1] = res[1];
            rgb[idx * C + 2] = res[2];
        }
        depths[idx] = p_view[2];
        radii[idx] = (int32_t)my_radius;
        means2D[2 * idx] = px;
        means2D[2 * idx + 1] = py;
        conic_opacity[4 * idx + 0] = conic[0];
        conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
    }
    uint32_t run = 0;
    for (int idx = 0; idx < P; idx++) {
        run += tiles_touched[idx];
        point_offsets[idx] = run;
    }
    return P > 0 ? (int64_t)point_offsets[P - 1] : 0;
}

/* rasterizer_impl.cu:52-63 (checkFrustum / markVisible) */
void fnx_oracle_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                             uint8_t *present) {
    (void)projmatrix;
    for (int idx = 0; idx < P; idx++) {
        float pv[3];
        present[idx] = (uint8_t)in_frustum(means3D + 3 * idx, viewmatrix, pv);
    }
}

typedef struct {
    uint64_t key;
    uint32_t val;
    uint32_t pos;
} kv_t;
static int kv_cmp(const void *a, const void *b) {
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);
}

/*
 * Stage 2: duplicateWithKeys (rasterizer_impl.cu:67-104) + stable SortPairs (:285-290, ties keep
 * emission order) + identifyTileRanges (:109-128; ranges zeroed first, :292).
 * keys_sorted[R], point_list[R], ranges[2*T] are caller-allocated.
 */
void fnx_oracle_bin(int P, int W, int H, const float *means2D, const float *depths, const uint32_t *point_offsets,
                    const int32_t *radii, int64_t R, uint64_t *keys_sorted, uint32_t *point_list, uint32_t *ranges) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    if (R <= 0) return;
    kv_t *kv = (kv_t *)malloc(sizeof(kv_t) * (size_t)R);
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
            uint32_t rmin[2], rmax[2];
            getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
            uint32_t dbits;
            memcpy(&dbits, &depths[idx], 4);
            for (uint32_t y = rmin[1]; y < rmax[1]; y++)
                for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * (uint32_t)gx + x);
                    key <<= 32;
                    key |= dbits;
                    kv[off].key = key;
                    kv[off].val = (uint32_t)idx;
                    kv[off].pos = off;
                    off++;
                }
        }
    }
    qsort(kv, (size_t)R, sizeof(kv_t), kv_cmp);
    for (int64_t i = 0; i < R; i++) {
        keys_sorted[i] = kv[i].key;
        point_list[i] = kv[i].val;
    }
    free(kv);
    for (int64_t i = 0; i < R; i++) {
        uint32_t currtile = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0)
            ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(keys_sorted[i - 1] >> 32);
            if (currtile != prevtile) {
                ranges[2 * prevtile + 1] = (uint32_t)i;
                ranges[2 * currtile] = (uint32_t)i;
            }
        }
        if (i == R - 1) ranges[2 * currtile + 1] = (uint32_t)R;
    }
}

/*
 * Stage 3: renderCUDA<C> forward (forward.cu:249-373).  Per pixel the batched tile loop is
 * equivalent to a sequential walk of the tile's range that stops when `done`.
 */
void fnx_oracle_render(int C, int W, int H, const uint32_t *ranges, const uint32_t *point_list, const float *means2D,
                       const float *features, const float *conic_opacity, const float *depths, const float *bg,
                       float *final_T, uint32_t *n_contrib, float *out_color, float *out_depth,
                       uint32_t *examined /* optional diagnostics: list entries walked per pixel */) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int ty = tile / gx, tx = tile % gx;
        for (int py = ty * BLOCK_Y; py < imin((ty + 1) * BLOCK_Y, H); py++)
            for (int px = tx * BLOCK_X; px < imin((tx + 1) * BLOCK_X, W); px++) {
                const uint32_t pix_id = (uint32_t)W * py + px;
                const float pixf[2] = {(float)px, (float)py};
                float T = 1.0f;
                uint32_t contributor = 0, last_contributor = 0;
                float Cacc[4] = {0, 0, 0, 0};
                float Dm = 15.0f; /* forward.cu:295 median-depth default */
                for (uint32_t k = r0; k < r1; k++) {
                    contributor++;
                    const uint32_t id = point_list[k];
                    float dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
                    const float *con_o = conic_opacity + 4 * id;
                    float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
                    if (power > 0.0f) continue;
                    float alpha = fminf(0.99f, con_o[3] * fnx_oracle_expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break; /* done = true */
                    for (int ch = 0; ch < C; ch++) Cacc[ch] += features[id * C + ch] * alpha * T;
                    if (T > 0.5f && test_T < 0.5) Dm = depths[id];
                    T = test_T;
                    last_contributor = contributor;
                }
                final_T[pix_id] = T;
                n_contrib[pix_id] = last_contributor;
                if (examined) examined[pix_id] = contributor;
                for (int ch = 0; ch < C; ch++) out_color[(size_t)ch * H * W + pix_id] = Cacc[ch] + T * bg[ch];
                out_depth[pix_id] = Dm;
            }
    }
}

/* backward.cu:20-132 */
static void computeColorFromSH_bwd(int idx, int deg, int max_coeffs, const float *means, const float *campos,
                                   const float *shs, const uint8_t *clamped, const float *dL_dcolor, float *dL_dmeans,
                                   float *dL_dshs) {
    float dir_orig[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
    float x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
    const float *sh = shs + (size_t)idx * max_coeffs * 3;
    float *dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
    float dL_dRGB[3];
    for (int c = 0; c < 3; c++) dL_dRGB[c] = dL_dcolor[3 * idx + c] * (clamped[3 * idx + c] ? 0.0f : 1.0f);
    float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
#define SH(k, c) sh[(k) * 3 + (c)]
#define DSH(k, c) dL_dsh[(k) * 3 + (c)]
    for (int c = 0; c < 3; c++) DSH(0, c) = SH_C0 * dL_dRGB[c];
    if (deg > 0) {
        float d1 = -SH_C1 * y, d2 = SH_C1 * z, d3 = -SH_C1 * x;
        for (int c = 0; c < 3; c++) {
            DSH(1, c) = d1 * dL_dRGB[c];
            DSH(2, c) = d2 * dL_dRGB[c];
            DSH(3, c) = d3 * dL_dRGB[c];
            dRGBdx[c] = -SH_C1 * SH(3, c);
            dRGBdy[c] = -SH_C1 * SH(1, c);
            dRGBdz[c] = SH_C1 * SH(2, c);
        }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            float d4 = SH_C2[0] * xy, d5 = SH_C2[1] * yz, d6 = SH_C2[2] * (2.f * zz - xx - yy), d7 = SH_C2[3] * xz,
                  d8 = SH_C2[4] * (xx - yy);
            for (int c = 0; c < 3; c++) {
                DSH(4, c) = d4 * dL_dRGB[c];
                DSH(5, c) = d5 * dL_dRGB[c];
                DSH(6, c) = d6 * dL_dRGB[c];
                DSH(7, c) = d7 * dL_dRGB[c];
                DSH(8, c) = d8 * dL_dRGB[c];
                dRGBdx[c] += SH_C2[0] * y * SH(4, c) + SH_C2[2] * 2.f * -x * SH(6, c) + SH_C2[3] * z * SH(7, c) +
                             SH_C2[4] * 2.f * x * SH(8, c);
                dRGBdy[c] += SH_C2[0] * x * SH(4, c) + SH_C2[1] * z * SH(5, c) + SH_C2[2] * 2.f * -y * SH(6, c) +
                             SH_C2[4] * 2.f * -y * SH(8, c);
                dRGBdz[c] += SH_C2[1] * y * SH(5, c) + SH_C2[2] * 2.f * 2.f * z * SH(6, c) + SH_C2[3] * x * SH(7, c);
            }
            if (deg > 2) {
                float d9 = SH_C3[0] * y * (3.f * xx - yy), d10 = SH_C3[1] * xy * z,
                      d11 = SH_C3[2] * y * (4.f * zz - xx - yy), d12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy),
                      d13 = SH_C3[4] * x * (4.f * zz - xx - yy), d14 = SH_C3[5] * z * (xx - yy),
                      d15 = SH_C3[6] * x * (xx - 3.f * yy);
                for (int c = 0; c < 3; c++) {
                    DSH(9, c) = d9 * dL_dRGB[c];
                    DSH(10, c) = d10 * dL_dRGB[c];
                    DSH(11, c) = d11 * dL_dRGB[c];
                    DSH(12, c) = d12 * dL_dRGB[c];
                    DSH(13, c) = d13 * dL_dRGB[c];
                    DSH(14, c) = d14 * dL_dRGB[c];
                    DSH(15, c) = d15 * dL_dRGB[c];
                    dRGBdx[c] += (SH_C3[0] * SH(9, c) * 3.f * 2.f * xy + SH_C3[1] * SH(10, c) * yz +
                                  SH_C3[2] * SH(11, c) * -2.f * xy + SH_C3[3] * SH(12, c) * -3.f * 2.f * xz +
                                  SH_C3[4] * SH(13, c) * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * SH(14, c) * 2.f * xz +
                                  SH_C3[6] * SH(15, c) * 3.f * (xx - yy));
                    dRGBdy[c] += (SH_C3[0] * SH(9, c) * 3.f * (xx - yy) + SH_C3[1] * SH(10, c) * xz +
                                  SH_C3[2] * SH(11, c) * (-3.f * yy + 4.f * zz - xx) +
                                  SH_C3[3] * SH(12, c) * -3.f * 2.f * yz + SH_C3[4] * SH(13, c) * -2.f * xy +
                                  SH_C3[5] * SH(14, c) * -2.f * yz + SH_C3[6] * SH(15, c) * -3.f * 2.f * xy);
                    dRGBdz[c] += (SH_C3[1] * SH(10, c) * xy + SH_C3[2] * SH(11, c) * 4.f * 2.f * yz +
                                  SH_C3[3] * SH(12, c) * 3.f * (2.f * zz - xx - yy) +
                                  SH_C3[4] * SH(13, c) * 4.f * 2.f * xz + SH_C3[5] * SH(14, c) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    float dL_ddir[3] = {dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
                        dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
                        dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]};
    /* dnormvdv(float3), auxiliary.h:95-105 */
    const float *v = dir_orig, *dv = dL_ddir;
    float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    float o0 = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
    float o1 = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
    float o2 = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
    dL_dmeans[3 * idx + 0] += o0;
    dL_dmeans[3 * idx + 1] += o1;
    dL_dmeans[3 * idx + 2] += o2;
}

/* backward.cu:137-263 */
static void computeCov2D_bwd(int idx, const float *means, const float *cov3Ds, float h_x, float h_y, float tan_fovx,
                             float tan_fovy, const float *view_matrix, const float *dL_dconics, float *dL_dmeans,
                             float *dL_dcov) {
    const float *cov3D = cov3Ds + 6 * idx;
    const float *mean = means + 3 * idx;
    float dL_dconic[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
    float t[3];
    transformPoint4x3(mean, view_matrix, t);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t[0] / t[2];
    const float tytz = t[1] / t[2];
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    const float x_grad_mul = txtz < -limx || txtz > limx ? 0 : 1;
    const float y_grad_mul = tytz < -limy || tytz > limy ? 0 : 1;
    mat3 J = mat3_cols(h_x / t[2], 0.0f, -(h_x * t[0]) / (t[2] * t[2]), 0.0f, h_y / t[2],
                       -(h_y * t[1]) / (t[2] * t[2]), 0, 0, 0);
    mat3 Wm = mat3_cols(view_matrix[0], view_matrix[4], view_matrix[8], view_matrix[1], view_matrix[5],
                        view_matrix[9], view_matrix[2], view_matrix[6], view_matrix[10]);
    mat3 Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    mat3 T = mat3_mul(Wm, J);
    mat3 cov2D = mat3_mul(mat3_mul(mat3_transpose(T), mat3_transpose(Vrk)), T);
    float a = cov2D.m[0][0] += 0.3f;
    float b = cov2D.m[0][1];
    float c = cov2D.m[1][1] += 0.3f;
    float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define Tm(i, j) T.m[i][j]
#define Vm(i, j) Vrk.m[i][j]
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * dL_dconic[0] + 2 * b * c * dL_dconic[1] + (denom - a * c) * dL_dconic[2]);
        dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2 * a * b * dL_dconic[1] + (denom - a * c) * dL_dconic[0]);
        dL_db = denom2inv * 2 * (b * c * dL_dconic[0] - (denom + 2 * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);
        dL_dcov[6 * idx + 0] = (Tm(0, 0) * Tm(0, 0) * dL_da + Tm(0, 0) * Tm(1, 0) * dL_db + Tm(1, 0) * Tm(1, 0) * dL_dc);
        dL_dcov[6 * idx + 3] = (Tm(0, 1) * Tm(0, 1) * dL_da + Tm(0, 1) * Tm(1, 1) * dL_db + Tm(1, 1) * Tm(1, 1) * dL_dc);
        dL_dcov[6 * idx + 5] = (Tm(0, 2) * Tm(0, 2) * dL_da + Tm(0, 2) * Tm(1, 2) * dL_db + Tm(1, 2) * Tm(1, 2) * dL_dc);
        dL_dcov[6 * idx + 1] = 2 * Tm(0, 0) * Tm(0, 1) * dL_da + (Tm(0, 0) * Tm(1, 1) + Tm(0, 1) * Tm(1, 0)) * dL_db +
                               2 * Tm(1, 0) * Tm(1, 1) * dL_dc;
        dL_dcov[6 * idx + 2] = 2 * Tm(0, 0) * Tm(0, 2) * dL_da + (Tm(0, 0) * Tm(1, 2) + Tm(0, 2) * Tm(1, 0)) * dL_db +
                               2 * Tm(1, 0) * Tm(1, 2) * dL_dc;
        dL_dcov[6 * idx + 4] = 2 * Tm(0, 2) * Tm(0, 1) * dL_da + (Tm(0, 1) * Tm(1, 2) + Tm(0, 2) * Tm(1, 1)) * dL_db +
                               2 * Tm(1, 1) * Tm(1, 2) * dL_dc;
    } else {
        for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
    }
    float dL_dT00 = 2 * (Tm(0, 0) * Vm(0, 0) + Tm(0, 1) * Vm(0, 1) + Tm(0, 2) * Vm(0, 2)) * dL_da +
                    (Tm(1, 0) * Vm(0, 0) + Tm(1, 1) * Vm(0, 1) + Tm(1, 2) * Vm(0, 2)) * dL_db;
    float dL_dT01 = 2 * (Tm(0, 0) * Vm(1, 0) + Tm(0, 1) * Vm(1, 1) + Tm(0, 2) * Vm(1, 2)) * dL_da +
                    (Tm(1, 0) * Vm(1, 0) + Tm(1, 1) * Vm(1, 1) + Tm(1, 2) * Vm(1, 2)) * dL_db;
    float dL_dT02 = 2 * (Tm(0, 0) * Vm(2, 0) + Tm(0, 1) * Vm(2, 1) + Tm(0, 2) * Vm(2, 2)) * dL_da +
                    (Tm(1, 0) * Vm(2, 0) + Tm(1, 1) * Vm(2, 1) + Tm(1, 2) * Vm(2, 2)) * dL_db;
    float dL_dT10 = 2 * (Tm(1, 0) * Vm(0, 0) + Tm(1, 1) * Vm(0, 1) + Tm(1, 2) * Vm(0, 2)) * dL_dc +
                    (Tm(0, 0) * Vm(0, 0) + Tm(0, 1) * Vm(0, 1) + Tm(0, 2) * Vm(0, 2)) * dL_db;
    float dL_dT11 = 2 * (Tm(1, 0) * Vm(1, 0) + Tm(1, 1) * Vm(1, 1) + Tm(1, 2) * Vm(1, 2)) * dL_dc +
                    (Tm(0, 0) * Vm(1, 0) + Tm(0, 1) * Vm(1, 1) + Tm(0, 2) * Vm(1, 2)) * dL_db;
    float dL_dT12 = 2 * (Tm(1, 0) * Vm(2, 0) + Tm(1, 1) * Vm(2, 1) + Tm(1, 2) * Vm(2, 2)) * dL_dc +
                    (Tm(0, 0) * Vm(2, 0) + Tm(0, 1) * Vm(2, 1) + Tm(0, 2) * Vm(2, 2)) * dL_db;
#undef Tm
#undef Vm
    float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
    float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
    float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
    float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
    float tz = 1.f / t[2];
    float tz2 = tz * tz;
    float tz3 = tz2 * tz;
    float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t[0]) * tz3 * dL_dJ02 +
                   (2 * h_y * t[1]) * tz3 * dL_dJ12;
    /* transformVec4x3Transpose, auxiliary.h:82-89 */
    const float *m = view_matrix;
    dL_dmeans[3 * idx + 0] = m[0] * dL_dtx + m[1] * dL_dty + m[2] * dL_dtz;
    dL_dmeans[3 * idx + 1] = m[4] * dL_dtx + m[5] * dL_dty + m[6] * dL_dtz;
    dL_dmeans[3 * idx + 2] = m[8] * dL_dtx + m[9] * dL_dty + m[10] * dL_dtz;
}

static float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* backward.cu:267-327 (no quaternion-normalisation Jacobian: local edit at :326) */
static void computeCov3D_bwd(int idx, const float *scale, float mod, const float *rot, const float *dL_dcov3Ds,
                             float *dL_dscales, float *dL_drots) {
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R = mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                       2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                       2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    S.m[0][0] = s[0];
    S.m[1][1] = s[1];
    S.m[2][2] = s[2];
    mat3 M = mat3_mul(S, R);
    const float *d = dL_dcov3Ds + 6 * idx;
    mat3 dL_dSigma = mat3_cols(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2],
                               0.5f * d[4], d[5]);
    /* 2.0f * M * dL_dSigma: (2.0f * M) first, glm scalar*mat is per-element */
    mat3 M2;
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) M2.m[c][rr] = 2.0f * M.m[c][rr];
    mat3 dL_dM = mat3_mul(M2, dL_dSigma);
    mat3 Rt = mat3_transpose(R);
    mat3 dL_dMt = mat3_transpose(dL_dM);
    dL_dscales[3 * idx + 0] = dot3(Rt.m[0], dL_dMt.m[0]);
    dL_dscales[3 * idx + 1] = dot3(Rt.m[1], dL_dMt.m[1]);
    dL_dscales[3 * idx + 2] = dot3(Rt.m[2], dL_dMt.m[2]);
    for (int k = 0; k < 3; k++) {
        dL_dMt.m[0][k] *= s[0];
        dL_dMt.m[1][k] *= s[1];
        dL_dMt.m[2][k] *= s[2];
    }
#define D(i, j) dL_dMt.m[i][j]
    float qx = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
    float qy = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) -
               4 * x * (D(2, 2) + D(1, 1));
    float qz = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) -
               4 * y * (D(2, 2) + D(0, 0));
    float qw = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) -
               4 * z * (D(1, 1) + D(0, 0));
#undef D
    dL_drots[4 * idx + 0] = qx;
    dL_drots[4 * idx + 1] = qy;
    dL_drots[4 * idx + 2] = qz;
    dL_drots[4 * idx + 3] = qw;
}

/*
 * Stage 4a: renderCUDA<C> backward (backward.cu:384-536), per pixel back-to-front.
 * Outputs (caller zero-initialised): dL_dmean2D[P*3] (z never written), dL_dconic[P*4]
 * (index 2 never written), dL_dopacity[P], dL_dcolor[P*C].
 */
void fnx_oracle_render_backward(int C, int P, int W, int H, const uint32_t *ranges, const uint32_t *point_list,
                                const float *bg, const float *means2D, const float *conic_opacity,
                                const float *colors, const float *final_Ts, const uint32_t *n_contrib,
                                const float *dL_dpixels, float *dL_dmean2D, float *dL_dconic, float *dL_dopacity,
                                float *dL_dcolors) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    /* total instances = end of the last non-empty range */
    uint32_t R = 0;
    for (int t = 0; t < gx * gy; t++)
        if (ranges[2 * t + 1] > R) R = ranges[2 * t + 1];
    const int NV = 6 + C; /* mean2D x,y | conic x,y,w | opacity | colour[C] */
    /* per-instance partial sums in double (one tile owns each slot => deterministic), merged in list order */
    double *part = (double *)calloc((size_t)R * NV + 1, sizeof(double));
    const float ddelx_dx = (float)(0.5 * W); /* backward.cu:444-445 */
    const float ddely_dy = (float)(0.5 * H);
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int ty = tile / gx, tx = tile % gx;
        for (int py = ty * BLOCK_Y; py < imin((ty + 1) * BLOCK_Y, H); py++)
            for (int px = tx * BLOCK_X; px < imin((tx + 1) * BLOCK_X, W); px++) {
                const uint32_t pix_id = (uint32_t)W * py + px;
                const float pixf[2] = {(float)px, (float)py};
                const float T_final = final_Ts[pix_id];
                float T = T_final;
                uint32_t contributor = r1 - r0;
                const uint32_t last_contributor = n_contrib[pix_id];
                float accum_rec[4] = {0, 0, 0, 0}, dL_dpixel[4] = {0, 0, 0, 0}, last_color[4] = {0, 0, 0, 0};
                for (int i = 0; i < C; i++) dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
                float last_alpha = 0;
                for (uint32_t k = r1; k-- > r0;) {
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    const uint32_t id = point_list[k];
                    const float dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
                    const float *con_o = conic_opacity + 4 * id;
                    const float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float G = fnx_oracle_expf(power);
                    const float alpha = fminf(0.99f, con_o[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    double *acc = part + (size_t)k * NV;
                    float dL_dalpha = 0.0f;
                    for (int ch = 0; ch < C; ch++) {
                        const float c = colors[id * C + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        const float dL_dchannel = dL_dpixel[ch];
                        dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                        acc[6 + ch] += (double)(dchannel_dcolor * dL_dchannel);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    float bg_dot_dpixel = 0;
                    for (int i = 0; i < C; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                    const float dL_dG = con_o[3] * dL_dalpha;
                    const float gdx = G * dx;
                    const float gdy = G * dy;
                    const float dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
                    const float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];
                    acc[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                    acc[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                    acc[2] += (double)(-0.5f * gdx * dx * dL_dG);
                    acc[3] += (double)(-0.5f * gdx * dy * dL_dG);
                    acc[4] += (double)(-0.5f * gdy * dy * dL_dG);
                    acc[5] += (double)(G * dL_dalpha);
                }
            }
    }
    double *tot = (double *)calloc((size_t)P * NV + 1, sizeof(double));
    for (uint32_t k = 0; k < R; k++) {
        const uint32_t id = point_list[k];
        for (int v = 0; v < NV; v++) tot[(size_t)id * NV + v] += part[(size_t)k * NV + v];
    }
    for (int i = 0; i < P; i++) {
        const double *t = tot + (size_t)i * NV;
        dL_dmean2D[3 * i + 0] = (float)t[0];
        dL_dmean2D[3 * i + 1] = (float)t[1];
        dL_dconic[4 * i + 0] = (float)t[2];
        dL_dconic[4 * i + 1] = (float)t[3];
        dL_dconic[4 * i + 3] = (float)t[4];
        dL_dopacity[i] = (float)t[5];
        for (int ch = 0; ch < C; ch++) dL_dcolors[(size_t)i * C + ch] = (float)t[6 + ch];
    }
    free(part);
    free(tot);
}

/*
 * Stage 4b: BACKWARD::preprocess = computeCov2DCUDA (backward.cu:137-263) then
 * preprocessCUDA<C> (backward.cu:332-381).  cov3Ds = the precomputed or the forward-computed
 * covariances (rasterizer_impl.cu:390).  Outputs caller zero-initialised.
 */
void fnx_oracle_preprocess_backward(int P, int D, int M, const float *means3D, const int32_t *radii, const float *shs,
                                    const uint8_t *clamped, const float *scales, const float *rotations,
                                    float scale_modifier, const float *cov3Ds, const float *viewmatrix,
                                    const float *projmatrix, int W, int H, float tan_fovx, float tan_fovy,
                                    const float *campos, const float *dL_dmean2D, const float *dL_dconic,
                                    float *dL_dmean3D, const float *dL_dcolor, float *dL_dcov3D, float *dL_dsh,
                                    float *dL_dscale, float *dL_drot) {
    const float focal_y = H / (2.0f * tan_fovy); /* rasterizer_impl.cu:360-361 */
    const float focal_x = W / (2.0f * tan_fovx);
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        computeCov2D_bwd(idx, means3D, cov3Ds, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix, dL_dconic,
                         dL_dmean3D, dL_dcov3D);
    }
    const float *proj = projmatrix;
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float *m = means3D + 3 * idx;
        float m_hom[4];
        transformPoint4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        const float g0 = dL_dmean2D[3 * idx], g1 = dL_dmean2D[3 * idx + 1];
        float dmx = (proj[0] * m_w - proj[3] * mul1) * g0 + (proj[1] * m_w - proj[3] * mul2) * g1;
        float dmy = (proj[4] * m_w - proj[7] * mul1) * g0 + (proj[5] * m_w - proj[7] * mul2) * g1;
        float dmz = (proj[8] * m_w - proj[11] * mul1) * g0 + (proj[9] * m_w - proj[11] * mul2) * g1;
        dL_dmean3D[3 * idx + 0] += dmx;
        dL_dmean3D[3 * idx + 1] += dmy;
        dL_dmean3D[3 * idx + 2] += dmz;
        if (shs) computeColorFromSH_bwd(idx, D, M, means3D, campos, shs, clamped, dL_dcolor, dL_dmean3D, dL_dsh);
        if (scales) computeCov3D_bwd(idx, scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov3D, dL_dscale, dL_drot);
    }
}
