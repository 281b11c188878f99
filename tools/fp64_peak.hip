// Micro-benchmark: sustained f64 FMA rate of the vector ALUs with every CU busy, at 1, 2, 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(double *out, int iters) {
    double a[8];
    for (int j = 0; j < 8; ++j) a[j] = out[threadIdx.x] + j;
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], b, c);
    }
    double s = 0;
    for (int j = 0; j < 8; ++j) s += a[j];
    if (s == 1.2345) out[threadIdx.x] = s;
}
int main() {
    double *out; hipMalloc(&out, 64 * 8); hipMemset(out, 0, 64 * 8);
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * 4 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * 64 * 32.0 * iters * blocks;
        printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s f64  (%.2f ns per wave-FMA per SIMD)\n", wps, ms, fl / ms / 1e9, ms * 1e6 / (32.0 * iters * wps));
    }
    return 0;
}
