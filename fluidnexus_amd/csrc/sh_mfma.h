// SH colours of a view batch on the MATRIX CORES (round 5; north_star: "MFMA used only for the dense SH-colour contraction",
// ch3 forward.cu:20-67).  Included by raster_forward.hip inside namespace fnx.  The production kernel of the FAST blend
// arithmetic (fnx_raster_opts_t.blend_math = 1); the exact arithmetic keeps sh_colors_views_kernel, whose expression
// order is the reference's.  FNX_LAB_SH_MFMA=0 / 1 in the environment pins either (tests, bench.py --sh-degree).
//
// colour[g][v][c] = sum_k basis_k(direction of Gaussian g seen from camera v) * sh[g][k][c] is, per Gaussian, a
// (V x 16) . (16 x 3) product whose BOTH operands belong to that Gaussian -- nothing is shared between Gaussians, so the
// matrix instruction whose shape fits is the batched outer product v_mfma_f32_4x4x1_16B_f32: 16 independent 4 x 4
// blocks per instruction.  Block b of a wave (lanes 4 b .. 4 b + 3) = one Gaussian, rows = four views, columns = (R, G, B,
// unused), K runs over the 16 coefficients with one instruction each.  Lane 4 b + i computes the basis of (Gaussian b,
// view i) and loads coefficient column i of Gaussian b: nothing is computed or read twice within a group of four views,
// and a wave's loads cover 16 Gaussians' coefficient rows instead of 64 threads striding through 192 bytes each.
// Measured on config 3's Gaussians x 5 views, degree 3 (profiles/r05_sh_bench.json): per-splat stage 102 -> 79 us, i.e. the
// contract's algorithmic bytes at 0.88 of HBM peak (0.68 with the scalar kernel); the matrix pipe itself is ~3 % busy --
// the kernel stays bound by reading the coefficients, what the instruction buys is the layout.
// Colours agree with the scalar kernel to fp32 rounding (2.4e-7: the matrix pipe fuses multiply and add and takes sign and
// constant inside the basis value), not bit for bit -- inside the fast mode's stated tolerance.
__global__ void __launch_bounds__(256)
sh_colors_views_mfma_kernel(int P, int D, int M, int V, const float *__restrict__ means3D, const float *__restrict__ campos,
                            const float *__restrict__ shs, uint8_t *__restrict__ clamped, float *__restrict__ rgb,
                            size_t geom_stride) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int g = wave * 16 + (lane >> 2), i = lane & 3;
    const bool live = g < P;
    const float *src = shs + (size_t)(live ? g : 0) * M * 3;
    float bk[16];  // B operand: coefficient k, channel i of this lane's Gaussian (channel 3 does not exist: zero)
#pragma unroll
    for (int k = 0; k < 16; k++) bk[k] = (live && i < 3 && k < M) ? src[3 * k + i] : 0.f;
    const float mx = live ? means3D[3 * g] : 0.f, my = live ? means3D[3 * g + 1] : 0.f, mz = live ? means3D[3 * g + 2] : 1.f;
    for (int v0 = 0; v0 < V; v0 += 4) {
        const int v = min(v0 + i, V - 1);
        const float dx = mx - campos[3 * v], dy = my - campos[3 * v + 1], dz = mz - campos[3 * v + 2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        const float x = dx / len, y = dy / len, z = dz / len;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        float a[16];  // A operand: the real-SH basis of (Gaussian, view v0 + i), signs and constants included
        a[0] = kSH0;
        a[1] = D > 0 ? -kSH1 * y : 0.f;
        a[2] = D > 0 ? kSH1 * z : 0.f;
        a[3] = D > 0 ? -kSH1 * x : 0.f;
        a[4] = D > 1 ? kSH2[0] * xy : 0.f;
        a[5] = D > 1 ? kSH2[1] * yz : 0.f;
        a[6] = D > 1 ? kSH2[2] * (2.0f * zz - xx - yy) : 0.f;
        a[7] = D > 1 ? kSH2[3] * xz : 0.f;
        a[8] = D > 1 ? kSH2[4] * (xx - yy) : 0.f;
        a[9] = D > 2 ? kSH3[0] * y * (3.0f * xx - yy) : 0.f;
        a[10] = D > 2 ? kSH3[1] * xy * z : 0.f;
        a[11] = D > 2 ? kSH3[2] * y * (4.0f * zz - xx - yy) : 0.f;
        a[12] = D > 2 ? kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) : 0.f;
        a[13] = D > 2 ? kSH3[4] * x * (4.0f * zz - xx - yy) : 0.f;
        a[14] = D > 2 ? kSH3[5] * z * (xx - yy) : 0.f;
        a[15] = D > 2 ? kSH3[6] * x * (xx - 3.0f * yy) : 0.f;
        v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; k++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k], bk[k], acc, 0, 0, 0);
        // acc[r] = colour of view v0 + r, channel i, of this lane's Gaussian
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (live && i < 3 && v0 + r < V) {
                const float res = acc[r] + 0.5f;
                view_at(clamped, geom_stride, v0 + r)[3 * (size_t)g + i] = res < 0;
                view_at(rgb, geom_stride, v0 + r)[3 * (size_t)g + i] = res > 0.0f ? res : 0.0f;
            }
        }
    }
}
