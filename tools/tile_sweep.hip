// Micro-benchmark: pure-write rate of the lane kernels' store shape (512-B wave segments, 900 elements per instance,
// 16 single-wave workgroups per CU, a little arithmetic per store) against the interleave width TW of the batch layout.
// Build: hipcc --offload-arch=gfx950 -O3 tools/tile_sweep.hip -o tools/tile_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
// remap: the TW / 64 workgroups of a tile are the ones the dispatcher sends to one XCD (workgroup id mod 8)
__global__ void tiled8_occ(double *p, size_t batch, int P, int TW, int spin, int remap) {
    extern __shared__ double pad[];
    size_t B = blockIdx.x;
    if (remap) {
        const size_t twb = TW / 64, grp = 8 * twb;
        B = ((B / grp) * 8 + B % 8) * twb + (B % grp) / 8;
    }
    size_t b = B * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double *q = p + (b / TW) * (size_t)P * TW + (b % TW);
    double x = (double)threadIdx.x;
    if (P < 0) pad[threadIdx.x] = x;
    for (int e = 0; e < P; ++e) {
        for (int k = 0; k < spin; ++k) x = x * 1.0000001 + 0.5;
        *q = x + (double)e; q += TW;
    }
}
int main() {
    const size_t batch = 1 << 20; const int P = 900;
    const size_t n = batch * P; const double gb = n * 8.0 / 1e9;
    double *p; CK(hipMalloc(&p, n * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 1; ++rep)
    for (int tw : {64, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 20}) {
        for (int spin : {2, 3}) { const int remap = spin == 3; if (remap && (tw < 64 || tw > 65536)) continue;
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(tiled8_occ, dim3(batch / 64), dim3(64), 10240, 0, p, batch, P, tw, 2, remap);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(tiled8_occ, dim3(batch / 64), dim3(64), 10240, 0, p, batch, P, tw, 2, remap);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
            printf("TW %7d spin 2 remap %d : %7.1f GB/s\n", tw, remap, gb / ms * 1e3);
        }
    }
    return 0;
}
