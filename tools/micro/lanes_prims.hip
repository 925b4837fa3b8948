// Developer check (GPU box): the cross-lane primitives of raster_backward_lanes.h against their definitions.
// build: hipcc --offload-arch=gfx950 -O2 -I fluidnexus_amd/csrc tools/micro/lanes_prims.hip -o build/exp/lanes_prims
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include "raster_backward_lanes_prims.h"
__global__ void k(const float *in, float *out) {
    const int lane = threadIdx.x;
    float x[4], y[4];
    for (int j = 0; j < 4; j++) x[j] = in[64 * j + lane], y[j] = x[j];
    fnx::row_scan_mul4(x);
    fnx::row_scan_add4(y);
    for (int j = 0; j < 4; j++) out[64 * j + lane] = x[j], out[256 + 64 * j + lane] = y[j];
    out[512 + lane] = fnx::row_prev(in[lane], 7.0f);
    out[576 + lane] = fnx::row_last(in[lane]);
    out[640 + lane] = fnx::rows_fold4(in[lane], in[64 + lane], in[128 + lane], in[192 + lane]);
}
int main() {
    float h[256], o[704], *d, *e;
    for (int i = 0; i < 256; i++) h[i] = 0.5f + 0.001f * ((i * 37) % 101);
    hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int j = 0; j < 4; j++)
        for (int l = 0; l < 64; l++) {
            double p = 1, s = 0;
            for (int i = l & ~15; i <= l; i++) p *= h[64 * j + i], s += h[64 * j + i];
            if (fabs(o[64 * j + l] - p) > 1e-5 * p) { if (bad++ < 5) printf("mul scan j%d lane %d: %g want %g\n", j, l, o[64 * j + l], p); }
            if (fabs(o[256 + 64 * j + l] - s) > 1e-5 * s) { if (bad++ < 10) printf("add scan j%d lane %d: %g want %g\n", j, l, o[256 + 64 * j + l], s); }
        }
    for (int l = 0; l < 64; l++) {
        const float wp = (l & 15) ? h[l - 1] : 7.0f, wl = h[(l & ~15) + 15];
        if (o[512 + l] != wp) { if (bad++ < 15) printf("row_prev lane %d: %g want %g\n", l, o[512 + l], wp); }
        if (o[576 + l] != wl) { if (bad++ < 20) printf("row_last lane %d: %g want %g\n", l, o[576 + l], wl); }
        const int r = l >> 4, e_ = l & 15, v = ((r & 1) << 1) | (r >> 1);
        double w = 0;
        for (int rr = 0; rr < 4; rr++) w += h[64 * v + 16 * rr + e_];
        if (fabs(o[640 + l] - w) > 1e-5 * w) { if (bad++ < 25) printf("rows_fold4 lane %d (row %d -> value %d): %g want %g\n", l, r, v, o[640 + l], w); }
    }
    printf(bad ? "FAILED %d\n" : "lanes primitives OK\n", bad);
    return bad != 0;
}
