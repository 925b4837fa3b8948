// Fused L1 + SSIM image loss for gfx950 (C ABI: include/fnx_losses.h).
// One 256-thread workgroup per 32x32 output tile: the 42x42 halo of both images is staged in LDS once (1.72x the
// tile, against 2.64x for a 16x16 tile), the 11-tap Gaussian runs as a row pass then a column pass, both out of LDS
// with the taps register-blocked (a thread produces 8 neighbouring row sums / 4 neighbouring column sums from one
// sliding window of LDS reads), and the SSIM map, its three partial derivatives and the L1 term come out of the
// same pass -- one read of each image instead of the reference's five depthwise convolutions plus ~15 elementwise
// kernels.  Every pixel's sums are accumulated tap 0..10 in order, row pass first: the values do not depend on the
// tiling.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>

#include "../../include/fnx_losses.h"
#include "../../include/fnx_raster.h"

namespace {

constexpr int TS = 32, R = 5, K = 11, HS = TS + 2 * R;  // tile, radius, taps, halo tile = 42
constexpr int RSEG = 8, CSEG = 4;  // outputs per thread in the row / column pass (TS / CSEG * TS == 256 threads)

struct Win {
    float g[K];
};

// normalised 1-D Gaussian, sigma = 1.5, fp32 like loss_utils.py:21-23
Win make_window() {
    Win w;
    float s = 0.f;
    for (int i = 0; i < K; i++) {
        w.g[i] = (float)std::exp(-((double)((i - K / 2) * (i - K / 2))) / (2.0 * 1.5 * 1.5));
        s += w.g[i];
    }
    for (int i = 0; i < K; i++) w.g[i] = w.g[i] / s;
    return w;
}

typedef float f2 __attribute__((ext_vector_type(2)));

// Pixel (y, x) of channel c (or the 3-channel mean): the address is clamped into the image and the value selected
// afterwards, so that the loads of an unrolled staging loop are unconditional and can all be in flight together.
// GREY: 0 = per channel, 1 = both images are 3-channel and their grey means are compared, 2 = as 1 but the TARGET is
// already its grey mean, one plane per image (it is constant across a frame's iterations: a third of the loads less).
// MEAN: form the mean of three planes here; else one plane (plane c of a per-channel image, or a pre-averaged target).
template <bool MEAN>
__device__ __forceinline__ float load_px(const float *__restrict__ im, int H, int W, int c, int y, int x) {
    const bool in = x >= 0 && y >= 0 && x < W && y < H;  // zero padding outside
    const size_t o = (size_t)(in ? y : 0) * W + (in ? x : 0), hw = (size_t)H * W;
    float v;
    if (MEAN) v = ((im[o] + im[hw + o]) + im[2 * hw + o]) * (1.0f / 3.0f);  // torch.mean over the 3 channels
    else v = im[(size_t)c * hw + o];
    return in ? v : 0.f;
}

// The two Gaussian passes run on packed pairs (v_pk_fma_f32: two lanes of work per instruction) with the
// multiply-adds fused; the image-loss values are compared with the reference's conv2d at a tolerance (its own
// accumulation order is the library's), not bit for bit.
template <int GREY>
__global__ void __launch_bounds__(256)
l1_ssim_forward_kernel(const float *__restrict__ img, const float *__restrict__ gt, int C, int H, int W, Win win,
                       float *__restrict__ partials, float *__restrict__ dmaps) {
#pragma clang fp contract(fast)
    __shared__ f2 s_p[HS][HS + 1];                              // (x, y) of the halo tile
    __shared__ f2 s_mu[HS][TS + 1], s_sq[HS][TS + 1];           // row sums of (x, y) and (x^2, y^2)
    __shared__ float s_xy[HS][TS + 1];                          // row sums of x y
    __shared__ float s_red[2][4];
    const int tid = threadIdx.x;
    const int Ce = GREY ? 1 : C;
    const int n = blockIdx.z / Ce, c = blockIdx.z - n * Ce, x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    img += (size_t)n * C * H * W;  // image n of the batch
    gt += (size_t)n * (GREY == 2 ? 1 : C) * H * W;
    dmaps += (size_t)n * 3 * Ce * H * W;
    constexpr int NLOAD = (HS * HS + 255) / 256;
    f2 rp[NLOAD];
#pragma unroll
    for (int u = 0; u < NLOAD; u++) {  // all global loads first, then the LDS stores
        const int i = min(tid + u * 256, HS * HS - 1), ly = i / HS, lx = i - ly * HS;
        rp[u].x = load_px<GREY != 0>(img, H, W, c, y0 + ly - R, x0 + lx - R);
        rp[u].y = load_px<GREY == 1>(gt, H, W, GREY == 2 ? 0 : c, y0 + ly - R, x0 + lx - R);
    }
#pragma unroll
    for (int u = 0; u < NLOAD; u++) {
        const int i = tid + u * 256, ly = i / HS, lx = i - ly * HS;
        if (i < HS * HS) s_p[ly][lx] = rp[u];
    }
    __syncthreads();
    for (int i = tid; i < HS * (TS / RSEG); i += 256) {  // row pass: RSEG neighbouring outputs of one halo row
        const int ly = i / (TS / RSEG), lx = (i - ly * (TS / RSEG)) * RSEG;
        f2 mu[RSEG], sq[RSEG];
        float xy[RSEG];
#pragma unroll
        for (int j = 0; j < RSEG; j++) {
            mu[j] = sq[j] = f2{0.f, 0.f};
            xy[j] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < RSEG + K - 1; t++) {  // input t feeds output j with tap k = t - j: k ascends with t
            const f2 p = s_p[ly][lx + t];
            const f2 pp = p * p;
            const float pxy = p.x * p.y;
#pragma unroll
            for (int j = 0; j < RSEG; j++) {
                const int k = t - j;
                if (k >= 0 && k < K) {
                    const float g = win.g[k];
                    mu[j] += p * g;
                    sq[j] += pp * g;
                    xy[j] += pxy * g;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RSEG; j++) {
            s_mu[ly][lx + j] = mu[j];
            s_sq[ly][lx + j] = sq[j];
            s_xy[ly][lx + j] = xy[j];
        }
    }
    __syncthreads();
    const int tx = tid & (TS - 1), ty = (tid / TS) * CSEG;  // column pass: CSEG outputs below one another
    f2 mu[CSEG], sq[CSEG];
    float xy[CSEG];
#pragma unroll
    for (int j = 0; j < CSEG; j++) {
        mu[j] = sq[j] = f2{0.f, 0.f};
        xy[j] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < CSEG + K - 1; t++) {
        const f2 hm = s_mu[ty + t][tx], hs = s_sq[ty + t][tx];
        const float hx = s_xy[ty + t][tx];
#pragma unroll
        for (int j = 0; j < CSEG; j++) {
            const int k = t - j;
            if (k >= 0 && k < K) {
                const float g = win.g[k];
                mu[j] += hm * g;
                sq[j] += hs * g;
                xy[j] += hx * g;
            }
        }
    }
    float l1 = 0.f, sm = 0.f;
    const int px = x0 + tx;
#pragma unroll
    for (int j = 0; j < CSEG; j++) {
        const int py = y0 + ty + j;
        if (px < W && py < H) {
            const float mu1 = mu[j].x, mu2 = mu[j].y, exx = sq[j].x, eyy = sq[j].y, exy = xy[j];
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = exx - mu1_sq, s2 = eyy - mu2_sq, s12 = exy - mu12;
            const float A = 2.f * mu12 + C1, B = 2.f * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s1 + s2 + C2;
            const float inv = __builtin_amdgcn_rcpf(Cc * D);  // 1 ulp
            const float smap = (A * B) * inv;
            // partials of the map w.r.t. mu1 (E[x^2], E[xy] held fixed), E[x^2] and E[xy]
            const float dmu1 = (2.f * mu2 * (B - A)) * inv - smap * (2.f * mu1 * (D - Cc)) * inv;
            const float dexx = -smap * __builtin_amdgcn_rcpf(D);
            const float dexy = 2.f * A * inv;
            const size_t hw = (size_t)H * W, o = (size_t)c * hw + (size_t)py * W + px;
            dmaps[o] = dmu1;
            dmaps[(size_t)Ce * hw + o] = dexx;
            dmaps[2 * (size_t)Ce * hw + o] = dexy;
            sm += smap;
            const f2 ctr = s_p[ty + j + R][tx + R];
            l1 += fabsf(ctr.x - ctr.y);
        }
    }
    // workgroup sums (shuffle within waves, then 4 partials)
    for (int off = 32; off >= 1; off >>= 1) {
        l1 += __shfl_xor(l1, off);
        sm += __shfl_xor(sm, off);
    }
    if ((tid & 63) == 0) {
        s_red[0][tid >> 6] = l1;
        s_red[1][tid >> 6] = sm;
    }
    __syncthreads();
    if (tid == 0) {
        const int b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[2 * b] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
        partials[2 * b + 1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
    }
}

// Optional tail of the backward kernel (workgroup 0 only): the per-image means and the scalar loss from the forward's
// tile partials.  Only logging consumes them, so the reduction rides inside the backward launch instead of being a
// launch of its own between the forward and the backward.  A wave per image, fixed summation order.
struct CombineArgs {
    const float *partials;  // NULL: no combine
    int N, nt;
    float inv_count, w_l1, w_dssim;
    float *per_image, *loss;
};
constexpr int kCombineMaxImages = 64;
__device__ void combine_in_workgroup(const CombineArgs &a, float *s_term /* [kCombineMaxImages] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int n = w; n < a.N; n += 4) {
        const float2 *p = reinterpret_cast<const float2 *>(a.partials) + (size_t)n * a.nt;
        float x = 0.f, y = 0.f;
        for (int i = lane; i < a.nt; i += 64) {
            const float2 v = p[i];
            x += v.x;
            y += v.y;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            x += __shfl_xor(x, off);
            y += __shfl_xor(y, off);
        }
        const float l1 = x * a.inv_count, ss = y * a.inv_count;
        if (lane == 0) {
            a.per_image[2 * n] = l1;
            a.per_image[2 * n + 1] = ss;
            s_term[n] = a.w_l1 * l1 + a.w_dssim * (1.0f - ss);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int n = 0; n < a.N; n++) total += s_term[n];
        a.loss[0] = total;
    }
}

template <int GREY>
__global__ void __launch_bounds__(256)
l1_ssim_backward_kernel(const float *__restrict__ img, const float *__restrict__ gt, int C, int H, int W, Win win,
                        const float *__restrict__ dmaps, const float *__restrict__ g_l1,
                        const float *__restrict__ g_ssim, int g_stride, float scale_l1, float scale_ssim,
                        float *__restrict__ dL_dimg, const CombineArgs comb) {
#pragma clang fp contract(fast)
    __shared__ f2 s_d01[HS][HS + 1];  // (d map / d mu1, d map / d E[x^2]) of the halo tile
    __shared__ float s_d2[HS][HS + 1];  // d map / d E[xy]
    __shared__ f2 s_h01[HS][TS + 1];
    __shared__ float s_h2[HS][TS + 1];
    __shared__ float s_term[kCombineMaxImages];
    if (comb.partials && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) combine_in_workgroup(comb, s_term);
    const int tid = threadIdx.x;
    const int Ce = GREY ? 1 : C;
    const int n = blockIdx.z / Ce, c = blockIdx.z - n * Ce, x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    const size_t hw = (size_t)H * W;
    img += (size_t)n * C * hw;
    gt += (size_t)n * (GREY == 2 ? 1 : C) * hw;
    dmaps += (size_t)n * 3 * Ce * hw;
    dL_dimg += (size_t)n * C * hw;
    g_l1 += (size_t)n * g_stride;
    g_ssim += (size_t)n * g_stride;
    constexpr int NLOAD = (HS * HS + 255) / 256;
    float rd[NLOAD][3];
#pragma unroll
    for (int u = 0; u < NLOAD; u++) {  // all global loads first, then the LDS stores
        const int i = min(tid + u * 256, HS * HS - 1), ly = i / HS, lx = i - ly * HS;
        const int x = x0 + lx - R, y = y0 + ly - R;
        const bool in = x >= 0 && y >= 0 && x < W && y < H;
        const size_t o = (size_t)c * hw + (size_t)(in ? y : 0) * W + (in ? x : 0);
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const float v = dmaps[(size_t)m * Ce * hw + o];
            rd[u][m] = in ? v : 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < NLOAD; u++) {
        const int i = tid + u * 256, ly = i / HS, lx = i - ly * HS;
        if (i < HS * HS) {
            s_d01[ly][lx] = f2{rd[u][0], rd[u][1]};
            s_d2[ly][lx] = rd[u][2];
        }
    }
    __syncthreads();
    for (int i = tid; i < HS * (TS / RSEG); i += 256) {  // row pass, RSEG outputs per thread (see the forward)
        const int ly = i / (TS / RSEG), lx = (i - ly * (TS / RSEG)) * RSEG;
        f2 a01[RSEG];
        float a2[RSEG];
#pragma unroll
        for (int j = 0; j < RSEG; j++) {
            a01[j] = f2{0.f, 0.f};
            a2[j] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < RSEG + K - 1; t++) {
            const f2 d01 = s_d01[ly][lx + t];
            const float d2 = s_d2[ly][lx + t];
#pragma unroll
            for (int j = 0; j < RSEG; j++) {
                const int k = t - j;
                if (k >= 0 && k < K) {
                    const float g = win.g[k];
                    a01[j] += d01 * g;
                    a2[j] += d2 * g;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RSEG; j++) {
            s_h01[ly][lx + j] = a01[j];
            s_h2[ly][lx + j] = a2[j];
        }
    }
    __syncthreads();
    const int tx = tid & (TS - 1), ty = (tid / TS) * CSEG;
    f2 v01[CSEG];
    float v2[CSEG];
#pragma unroll
    for (int j = 0; j < CSEG; j++) {
        v01[j] = f2{0.f, 0.f};
        v2[j] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < CSEG + K - 1; t++) {
        const f2 h01 = s_h01[ty + t][tx];
        const float h2 = s_h2[ty + t][tx];
#pragma unroll
        for (int j = 0; j < CSEG; j++) {
            const int k = t - j;
            if (k >= 0 && k < K) {
                const float g = win.g[k];
                v01[j] += h01 * g;
                v2[j] += h2 * g;
            }
        }
    }
    const int px = x0 + tx;
    const float inv_cnt = 1.0f / (float)((size_t)Ce * hw);
    const float gl1 = g_l1[0] * scale_l1 * inv_cnt, gss = g_ssim[0] * scale_ssim * inv_cnt;
#pragma unroll
    for (int j = 0; j < CSEG; j++) {
        const int py = y0 + ty + j;
        if (px >= W || py >= H) continue;
        const float x = load_px<GREY != 0>(img, H, W, c, py, px), y = load_px<GREY == 1>(gt, H, W, GREY == 2 ? 0 : c, py, px);
        const float d = x - y;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        float grad = gl1 * sgn + gss * (v01[j].x + 2.f * x * v01[j].y + y * v2[j]);
        const size_t o = (size_t)py * W + px;
        if (GREY) {
            grad = grad * (1.0f / 3.0f);
            dL_dimg[o] = grad;
            dL_dimg[hw + o] = grad;
            dL_dimg[2 * hw + o] = grad;
        } else {
            dL_dimg[(size_t)c * hw + o] = grad;
        }
    }
}

// per_image[n] = (mean |x - y|, mean ssim_map) of image n; loss = sum_n (w_l1 * l1_n + w_dssim * (1 - ssim_n)).
// One workgroup of 16 waves, a wave per image (images n = w, w + 16, ...): fixed summation order (tile index
// ascending per lane, a butterfly over the lanes, images in order) -> deterministic.
__global__ void __launch_bounds__(1024)
image_loss_combine_kernel(const float *__restrict__ partials, int N, int nt, float inv_count, float w_l1, float w_dssim,
                          float *__restrict__ per_image, float *__restrict__ loss) {
    extern __shared__ float s_term[];  // N per-image loss terms
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int n = w; n < N; n += 16) {
        const float2 *p = reinterpret_cast<const float2 *>(partials) + (size_t)n * nt;
        float a = 0.f, b = 0.f;
        for (int i = lane; i < nt; i += 64) {
            const float2 v = p[i];
            a += v.x;
            b += v.y;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            a += __shfl_xor(a, off);
            b += __shfl_xor(b, off);
        }
        const float l1 = a * inv_count, ss = b * inv_count;
        if (lane == 0) {
            per_image[2 * n] = l1;
            per_image[2 * n + 1] = ss;
            s_term[n] = w_l1 * l1 + w_dssim * (1.0f - ss);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int n = 0; n < N; n++) total += s_term[n];
        loss[0] = total;
    }
}

template <typename... A>
void launch_l1_ssim_backward(int grey, dim3 grid, hipStream_t stream, A... args) {
    if (grey == 2) hipLaunchKernelGGL(l1_ssim_backward_kernel<2>, grid, dim3(256), 0, stream, args...);
    else if (grey) hipLaunchKernelGGL(l1_ssim_backward_kernel<1>, grid, dim3(256), 0, stream, args...);
    else hipLaunchKernelGGL(l1_ssim_backward_kernel<0>, grid, dim3(256), 0, stream, args...);
}

// ---------------------------------------------------------------------------------------------
// Visual-particle stage (train_visual_particle.py:133-222): the leaves are the raw colour / opacity / scales /
// rotation of the fluid Gaussians.  One kernel activates them into the leading rows of the concatenated
// [fluid | background] arrays the rasteriser reads (gm_dynamics.py getters: identity x3, sigmoid, exp, normalize;
// pipe_dynamics.py:88-148); one kernel turns the rasteriser's gradients with respect to those rows, plus the
// view-independent terms of the loss (consistency MSE with the previous frame per attribute, scale-ratio
// regulariser, tvp:161-194), into the gradients of the raw leaves -- instead of ~90 elementwise / reduction launches.
__global__ void __launch_bounds__(256)
level2_activate_kernel(const float *__restrict__ rc, const float *__restrict__ ro, const float *__restrict__ rs,
                       const float *__restrict__ rr, int n, float *__restrict__ colors, float *__restrict__ opacity,
                       float *__restrict__ scales, float *__restrict__ rot) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float c = rc[i];
    colors[3 * i] = colors[3 * i + 1] = colors[3 * i + 2] = c;
    opacity[i] = 1.0f / (1.0f + expf(-ro[i]));
#pragma unroll
    for (int k = 0; k < 3; k++) scales[3 * i + k] = expf(rs[3 * i + k]);
    const float4 q = *reinterpret_cast<const float4 *>(rr + 4 * (size_t)i);
    const float inv = 1.0f / fmaxf(sqrtf((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w)), 1e-12f);  // F.normalize
    *reinterpret_cast<float4 *>(rot + 4 * (size_t)i) = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
}

struct Level2Args {
    const float *raw[4], *prev[4], *g[4];  // colour [n,1] (g: [n,3]), opacity [n,1], scales [n,3], rotation [n,4]
    float *d[4];
    float lam[4], lam_reg, reg_thr, reg_count, scale;
    int n, n_prev;
};

__global__ void __launch_bounds__(256) level2_backward_kernel(const Level2Args a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const bool has_prev = i < a.n_prev;
    const float np = (float)a.n_prev;
    // d mse(raw[:n_prev], prev) / d raw = 2 (raw - prev) / (n_prev * dim); counted reg_count (= local views) times
    if (a.d[0]) {
        const float r = a.raw[0][i];
        float g = (a.g[0][3 * i] + a.g[0][3 * i + 1]) + a.g[0][3 * i + 2];
        if (has_prev) g += a.reg_count * a.lam[0] * 2.0f * (r - a.prev[0][i]) / np;
        a.d[0][i] = g * a.scale;
    }
    if (a.d[1]) {
        const float r = a.raw[1][i], o = 1.0f / (1.0f + expf(-r));
        float g = a.g[1][i] * (o * (1.0f - o));
        if (has_prev) g += a.reg_count * a.lam[1] * 2.0f * (r - a.prev[1][i]) / np;
        a.d[1][i] = g * a.scale;
    }
    if (a.d[2]) {
        float r[3], s[3], g[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            r[k] = a.raw[2][3 * i + k];
            s[k] = expf(r[k]);
            g[k] = a.g[2][3 * i + k] * s[k];
            if (has_prev) g[k] += a.reg_count * a.lam[2] * 2.0f * (r[k] - a.prev[2][3 * i + k]) / (3.0f * np);
        }
        if (a.lam_reg > 0.0f) {  // lam_reg * mean_i relu(max(s) / min(s) - thr): first maximal / minimal index, as torch
            int hi = 0, lo = 0;
#pragma unroll
            for (int k = 1; k < 3; k++) {
                if (s[k] > s[hi]) hi = k;
                if (s[k] < s[lo]) lo = k;
            }
            if (s[hi] / s[lo] - a.reg_thr > 0.0f) {
                const float w = a.reg_count * a.lam_reg / (float)a.n;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float dr = 0.0f;
                    if (k == hi) dr += 1.0f / s[lo];
                    if (k == lo) dr -= s[hi] / (s[lo] * s[lo]);
                    g[k] += w * dr * s[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) a.d[2][3 * i + k] = g[k] * a.scale;
    }
    if (a.d[3]) {
        const float4 q = *reinterpret_cast<const float4 *>(a.raw[3] + 4 * (size_t)i);
        const float4 gq = *reinterpret_cast<const float4 *>(a.g[3] + 4 * (size_t)i);
        const float nrm = fmaxf(sqrtf((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w)), 1e-12f), inv = 1.0f / nrm;
        const float ux = q.x * inv, uy = q.y * inv, uz = q.z * inv, uw = q.w * inv;
        const float dot = (ux * gq.x + uy * gq.y) + (uz * gq.z + uw * gq.w);
        float g[4] = {(gq.x - ux * dot) * inv, (gq.y - uy * dot) * inv, (gq.z - uz * dot) * inv, (gq.w - uw * dot) * inv};
        const float r[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (has_prev) g[k] += a.reg_count * a.lam[3] * 2.0f * (r[k] - a.prev[3][4 * i + k]) / (4.0f * np);
            a.d[3][4 * i + k] = g[k] * a.scale;
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
bool args_ok(int C, int H, int W, int grey) { return C > 0 && H > 0 && W > 0 && grey >= 0 && grey <= 2 && (!grey || C == 3); }

}  // namespace

extern "C" {
int fnx_losses_abi_version(void) { return 1; }
const char *fnx_losses_last_error(void) { return g_err; }
int fnx_l1_ssim_tiles(int C, int H, int W, int grey) {
    return (grey ? 1 : C) * ((H + TS - 1) / TS) * ((W + TS - 1) / TS);
}
int fnx_l1_ssim_forward_batch(const float *img, const float *gt, int N, int C, int H, int W, int grey, float *partials,
                              float *dmaps, fnx_stream_t stream) {
    if (N < 1 || !args_ok(C, H, W, grey) || !img || !gt || !partials || !dmaps)
        return fail(FNX_ERR_INVALID_ARG, "l1_ssim_forward: bad argument (grey needs C == 3)");
    static const Win win = make_window();
    dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, N * (grey ? 1 : C));
    if (grey == 2)
        hipLaunchKernelGGL(l1_ssim_forward_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, img, gt, C, H, W, win,
                           partials, dmaps);
    else if (grey)
        hipLaunchKernelGGL(l1_ssim_forward_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, img, gt, C, H, W, win,
                           partials, dmaps);
    else
        hipLaunchKernelGGL(l1_ssim_forward_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, img, gt, C, H, W, win,
                           partials, dmaps);
    return hip_check("l1_ssim_forward");
}
int fnx_l1_ssim_backward_batch(const float *img, const float *gt, int N, int C, int H, int W, int grey,
                               const float *dmaps, const float *g_l1, const float *g_ssim, float *dL_dimg,
                               fnx_stream_t stream) {
    if (N < 1 || !args_ok(C, H, W, grey) || !img || !gt || !dmaps || !g_l1 || !g_ssim || !dL_dimg)
        return fail(FNX_ERR_INVALID_ARG, "l1_ssim_backward: bad argument");
    static const Win win = make_window();
    dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, N * (grey ? 1 : C));
    launch_l1_ssim_backward(grey, grid, (hipStream_t)stream, img, gt, C, H, W, win,
                            dmaps, g_l1, g_ssim, 1, 1.0f, 1.0f, dL_dimg, CombineArgs{nullptr, 0, 0, 0.f, 0.f, 0.f, nullptr, nullptr});
    return hip_check("l1_ssim_backward");
}
int fnx_image_loss_forward(const float *img, const float *gt, int N, int C, int H, int W, int grey, float w_l1,
                           float w_dssim, float *partials, float *dmaps, float *per_image, float *loss,
                           fnx_stream_t stream) {
    if (!per_image || !loss) return fail(FNX_ERR_INVALID_ARG, "image_loss_forward: bad argument");
    int rc = fnx_l1_ssim_forward_batch(img, gt, N, C, H, W, grey, partials, dmaps, stream);
    if (rc) return rc;
    const int nt = fnx_l1_ssim_tiles(C, H, W, grey);
    const float inv = 1.0f / (float)((size_t)(grey ? 1 : C) * H * W);
    hipLaunchKernelGGL(image_loss_combine_kernel, dim3(1), dim3(1024), (size_t)N * sizeof(float), (hipStream_t)stream,
                       partials, N, nt, inv, w_l1, w_dssim, per_image, loss);
    return hip_check("image_loss_forward");
}
int fnx_image_loss_backward(const float *img, const float *gt, int N, int C, int H, int W, int grey, float w_l1,
                            float w_dssim, const float *dmaps, const float *g_loss, float *dL_dimg,
                            fnx_stream_t stream) {
    if (N < 1 || !args_ok(C, H, W, grey) || !img || !gt || !dmaps || !g_loss || !dL_dimg)
        return fail(FNX_ERR_INVALID_ARG, "image_loss_backward: bad argument");
    static const Win win = make_window();
    dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, N * (grey ? 1 : C));
    // d loss / d l1_n = w_l1, d loss / d ssim_n = -w_dssim; the upstream scalar multiplies both
    launch_l1_ssim_backward(grey, grid, (hipStream_t)stream, img, gt, C, H, W, win,
                            dmaps, g_loss, g_loss, 0, w_l1, -w_dssim, dL_dimg,
                       CombineArgs{nullptr, 0, 0, 0.f, 0.f, 0.f, nullptr, nullptr});
    return hip_check("image_loss_backward");
}
int fnx_image_loss_value_and_grad(const float *img, const float *gt, int N, int C, int H, int W, int grey, float w_l1,
                                  float w_dssim, float *partials, float *dmaps, float *per_image, float *loss,
                                  const float *g_loss, float *dL_dimg, fnx_stream_t stream) {
    if (N < 1 || N > kCombineMaxImages || !per_image || !loss || !g_loss || !dL_dimg)
        return fail(FNX_ERR_INVALID_ARG, "image_loss_value_and_grad: bad argument (1 <= N <= %d)", kCombineMaxImages);
    int rc = fnx_l1_ssim_forward_batch(img, gt, N, C, H, W, grey, partials, dmaps, stream);
    if (rc) return rc;
    static const Win win = make_window();
    const int nt = fnx_l1_ssim_tiles(C, H, W, grey);
    const float inv = 1.0f / (float)((size_t)(grey ? 1 : C) * H * W);
    dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, N * (grey ? 1 : C));
    launch_l1_ssim_backward(grey, grid, (hipStream_t)stream, img, gt, C, H, W, win,
                            dmaps, g_loss, g_loss, 0, w_l1, -w_dssim, dL_dimg,
                       CombineArgs{partials, N, nt, inv, w_l1, w_dssim, per_image, loss});
    return hip_check("image_loss_value_and_grad");
}
int fnx_l1_ssim_forward(const float *img, const float *gt, int C, int H, int W, int grey, float *partials,
                        float *dmaps, fnx_stream_t stream) {
    return fnx_l1_ssim_forward_batch(img, gt, 1, C, H, W, grey, partials, dmaps, stream);
}
int fnx_l1_ssim_backward(const float *img, const float *gt, int C, int H, int W, int grey, const float *dmaps,
                         const float *g_l1, const float *g_ssim, float *dL_dimg, fnx_stream_t stream) {
    return fnx_l1_ssim_backward_batch(img, gt, 1, C, H, W, grey, dmaps, g_l1, g_ssim, dL_dimg, stream);
}
int fnx_level2_activate(const float *raw_color, const float *raw_opacity, const float *raw_scales, const float *raw_rotation,
                        int n, float *colors, float *opacity, float *scales, float *rotations, fnx_stream_t stream) {
    if (n < 0 || (n > 0 && (!raw_color || !raw_opacity || !raw_scales || !raw_rotation || !colors || !opacity || !scales ||
                            !rotations)))
        return fail(FNX_ERR_INVALID_ARG, "level2_activate: bad argument");
    if (n == 0) return FNX_OK;
    hipLaunchKernelGGL(level2_activate_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw_color,
                       raw_opacity, raw_scales, raw_rotation, n, colors, opacity, scales, rotations);
    return hip_check("level2_activate");
}
int fnx_level2_backward(const float *const raw[4], const float *const prev[4], const float *const g[4], float *const d[4],
                        int n, int n_prev, const float lambdas[4], float lambda_reg, float reg_threshold, float reg_count,
                        float scale, fnx_stream_t stream) {
    if (n < 0 || n_prev < 0 || n_prev > n || !raw || !prev || !g || !d || !lambdas)
        return fail(FNX_ERR_INVALID_ARG, "level2_backward: bad argument");
    Level2Args a;
    for (int k = 0; k < 4; k++) {
        if (d[k] && (!raw[k] || !g[k] || (n_prev > 0 && !prev[k])))
            return fail(FNX_ERR_INVALID_ARG, "level2_backward: attribute %d has an output but no input", k);
        a.raw[k] = raw[k];
        a.prev[k] = prev[k];
        a.g[k] = g[k];
        a.d[k] = d[k];
        a.lam[k] = lambdas[k];
    }
    a.lam_reg = lambda_reg;
    a.reg_thr = reg_threshold;
    a.reg_count = reg_count;
    a.scale = scale;
    a.n = n;
    a.n_prev = n_prev;
    if (n == 0) return FNX_OK;
    hipLaunchKernelGGL(level2_backward_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return hip_check("level2_backward");
}
}
