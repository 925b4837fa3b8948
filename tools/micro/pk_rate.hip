// Developer microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
#define N_IT 4096
__global__ void k_scalar(float *out, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < N_IT; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pk(float *out, float a, float b) {
    float2v x[8];
    float2v av = {a, a}, bv = {b, b};
    for (int i = 0; i < 8; i++) x[i] = (float2v){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    for (int it = 0; it < N_IT; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(av), "v"(bv));
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pk_muladd(float *out, float a, float b) {
    float2v x[8];
    float2v av = {a, a}, bv = {b, b};
    for (int i = 0; i < 8; i++) x[i] = (float2v){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    for (int it = 0; it < N_IT / 2; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(av));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(bv));
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *d; hipMalloc(&d, 256 * 4 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
    for (int which = 0; which < 3; which++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(threads), 0, 0, d, 1.0001f, 0.5f);
            if (which == 1) hipLaunchKernelGGL(k_pk, dim3(blocks), dim3(threads), 0, 0, d, 1.0001f, 0.5f);
            if (which == 2) hipLaunchKernelGGL(k_pk_muladd, dim3(blocks), dim3(threads), 0, 0, d, 1.0001f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)blocks * threads / 64 * N_IT * 8;  // wave instructions
            if (rep) printf("%s: %.3f ms, %.1f G wave-instr/s, per SIMD %.2f cycles/instr @2.4GHz\n",
                            which == 0 ? "v_fma_f32" : which == 1 ? "v_pk_fma_f32" : "v_pk_mul+v_pk_add", ms, instr / ms / 1e6,
                            ms * 1e-3 * 2.4e9 * 1024 / instr);
        }
    }
    return 0;
}
