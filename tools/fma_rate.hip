// Micro-benchmark: f64 FMA issue rate / dependent latency and LDS round trips on one wavefront (clock64 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_dep(double *out, long long *cyc, int iters) {
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a = __builtin_fma(a, b, c);
    }
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_ind(double *out, long long *cyc, int iters) {
    double a[8];
    for (int j = 0; j < 8; ++j) a[j] = out[threadIdx.x] + j;
    const double b = 1.0000001, c = 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], b, c);
    }
    long long t1 = clock64();
    double s = 0;
    for (int j = 0; j < 8; ++j) s += a[j];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_lds(double *out, long long *cyc, int iters) {
    __shared__ double buf[128];
    buf[threadIdx.x] = out[threadIdx.x];
    buf[64 + threadIdx.x] = 0;
    __syncthreads();
    double a = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        // write own slot, read neighbour's (round trip through LDS, dependent)
        buf[(it & 1) * 64 + threadIdx.x] = a + 1.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        a = buf[(it & 1) * 64 + ((threadIdx.x + 1) & 63)];
    }
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8 * 1024);
    hipMemset(out, 0, 64 * 8);
    const int iters = 10000;
    long long h[1024];
    for (int blocks : {1, 1024, 4096}) {
        hipLaunchKernelGGL(k_dep, dim3(blocks), dim3(64), 0, 0, out, cyc, iters); hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
        printf("blocks %d dep   : %.2f clk per fma\n", blocks, (double)h[0] / (iters * 16.0));
        hipLaunchKernelGGL(k_ind, dim3(blocks), dim3(64), 0, 0, out, cyc, iters); hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
        printf("blocks %d indep : %.2f clk per fma\n", blocks, (double)h[0] / (iters * 16.0));
        hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(64), 0, 0, out, cyc, iters); hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
        printf("blocks %d lds rt: %.2f clk per round trip\n", blocks, (double)h[0] / iters);
    }
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0); printf("wall clock rate kHz %d\n", clk);
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); printf("clock rate kHz %d\n", clk);
    return 0;
}
