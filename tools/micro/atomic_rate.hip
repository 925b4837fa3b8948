// Developer micro-benchmark (GPU box): what fp32 global atomics cost on this chip -- throughput over a small (L2-sized) array
// at random / clustered addresses, and how long a vector LOAD issued behind three atomics takes to return (vmcnt retires in
// order) against the same load alone.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/atomic_rate.hip -o build/exp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k_rate(float *acc, const uint32_t *ids, int per_thread, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < per_thread; i++) {
        const uint32_t id = ids[(size_t)i * n + t];
        unsafeAtomicAdd(&acc[3 * (size_t)id + 0], 1.0f);
        unsafeAtomicAdd(&acc[3 * (size_t)id + 1], 1.0f);
        unsafeAtomicAdd(&acc[3 * (size_t)id + 2], 1.0f);
    }
}
__global__ void k_lat(float *acc, const uint32_t *ids, const float *other, unsigned long long *out, int with_atomics, int iters, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long tot = 0;
    float s = 0.f;
    for (int i = 0; i < iters; i++) {
        const uint32_t id = ids[(size_t)i * n + t];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (with_atomics) {
            unsafeAtomicAdd(&acc[3 * (size_t)id + 0], 1.0f);
            unsafeAtomicAdd(&acc[3 * (size_t)id + 1], 1.0f);
            unsafeAtomicAdd(&acc[3 * (size_t)id + 2], 1.0f);
        }
        const unsigned long long t0 = wall_clock64();
        const float v = other[((size_t)id * 977 + i) % (1u << 20)];
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
        tot += wall_clock64() - t0;
        s += v;
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(out, tot);
    if (s == 12345.f) out[1] = 1;
}
int main() {
    const int P = 200000, grid = 1024, block = 256, n = grid * block, per = 6;
    float *acc, *other; uint32_t *ids; unsigned long long *out;
    hipMalloc(&acc, 3 * P * 4); hipMalloc(&other, 4 << 20); hipMalloc(&ids, (size_t)per * n * 4); hipMalloc(&out, 16);
    hipMemset(acc, 0, 3 * P * 4); hipMemset(other, 0, 4 << 20);
    std::vector<uint32_t> h((size_t)per * n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++) {  // 0 random ids, 1 a workgroup's ids clustered in a window of 4096, 2 sequential
        uint32_t r = 12345;
        for (size_t i = 0; i < h.size(); i++) {
            r = r * 1664525u + 1013904223u;
            const size_t t = i % n;
            h[i] = mode == 0 ? (r >> 8) % P : mode == 1 ? ((t / 256) * 197 % (P / 4096)) * 4096 + (r >> 8) % 4096 : (uint32_t)(i % P);
        }
        hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_rate, dim3(grid), dim3(block), 0, 0, acc, ids, per, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("mode %d: %.1f M atomics in %.1f us = %.1f G/s\n", mode, 3.0 * per * n / 1e6, ms * 1e3, 3.0 * per * n / ms / 1e6);
        }
        for (int wa = 0; wa < 2; wa++) {
            hipMemset(out, 0, 16);
            hipLaunchKernelGGL(k_lat, dim3(grid), dim3(block), 0, 0, acc, ids, other, out, wa, per, n);
            unsigned long long o[2]; hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
            printf("mode %d: load %s atomics: %.2f us per load (wall clock 100 MHz)\n", mode, wa ? "behind three" : "without", o[0] / (double)(grid * block / 64) / per / 100.0);
        }
    }
    return 0;
}
