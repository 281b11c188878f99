// Micro-benchmark: store rate of ONE compute unit.  G workgroups (one per CU while G <= 256) of `block` threads each
// write a private contiguous region with 8-byte stores, lanes along addresses; the aggregate rate divided by G is what a
// single CU sustains when the memory system as a whole is far from saturated.
// Build: hipcc --offload-arch=gfx950 -O3 tools/cu_write_rate.hip -o tools/cu_write_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int VEC>
__global__ void region_fill(double *p, size_t per_wg) {
    double *q = p + (size_t)blockIdx.x * per_wg;
    if (VEC == 1) {
        for (size_t i = threadIdx.x; i < per_wg; i += blockDim.x) q[i] = 1.0;
    } else {
        double2 *q2 = (double2 *)q;
        for (size_t i = threadIdx.x; i < per_wg / 2; i += blockDim.x) q2[i] = make_double2(1.0, 2.0);
    }
}
int main() {
    const size_t per_wg = (size_t)4 << 20;  // doubles: 32 MiB per workgroup
    double *p; CK(hipMalloc(&p, 1024 * per_wg * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int vec : {1, 2})
    for (int block : {64, 256, 512, 1024})
        for (int g : {1, 8, 64, 256, 512, 1024}) {
            auto run = [&] { if (vec == 1) region_fill<1><<<g, block>>>(p, per_wg); else region_fill<2><<<g, block>>>(p, per_wg); };
            run(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(a)); run(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            const double gbs = g * per_wg * 8.0 / 1e9 / (ms * 1e-3);
            printf("%2d B/lane block %4d workgroups %4d : %8.1f GB/s total, %6.1f GB/s per workgroup\n", 8 * vec, block, g, gbs, gbs / g);
        }
    return 0;
}
