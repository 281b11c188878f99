// Micro-benchmark: what pure-write HBM bandwidth does MI355X sustain for the store shapes the lane kernels use?
//   fill16   contiguous stream, 16 B per lane (dwordx4)
//   fill8    contiguous stream, 8 B per lane (dwordx2)
//   planes8  each wave writes 512-B segments (8 B per lane) to P planes `plane_stride` apart: the access
//            pattern of a batch-interleaved output with P elements per instance
//   planes16 the same with two instances per lane (16 B per lane, 1 KiB segments)
// Build: hipcc --offload-arch=gfx950 -O3 tools/write_ceiling.hip -o tools/write_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ void fill16(double2 *p, size_t n2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += s) p[i] = make_double2(1.0, 2.0);
}
__global__ void fill8(double *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) p[i] = 1.0;
}
// batch instances, P planes; instance b element e at p[e * batch + b]
__global__ void planes8(double *p, size_t batch, int P) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double *q = p + b;
    for (int e = 0; e < P; ++e) { *q = (double)e; q += batch; }
}
__global__ void planes16(double2 *p, size_t batch2, int P) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch2) return;
    double2 *q = p + b;
    for (int e = 0; e < P; ++e) { *q = make_double2((double)e, 1.0); q += batch2; }
}
// tiled: tile of TW instances, element e of instance b at (b/TW)*P*TW + e*TW + b%TW
__global__ void tiled8(double *p, size_t batch, int P, int TW) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double *q = p + (b / TW) * (size_t)P * TW + (b % TW);
    for (int e = 0; e < P; ++e) { *q = (double)e; q += TW; }
}

// non-temporal variants (stores marked streaming: no intent to re-read)
__global__ void fill16_nt(double2 *p, size_t n2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += s) { __builtin_nontemporal_store(1.0, &p[i].x); __builtin_nontemporal_store(2.0, &p[i].y); }
}
__global__ void tiled8_nt(double *p, size_t batch, int P, int TW) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double *q = p + (b / TW) * (size_t)P * TW + (b % TW);
    for (int e = 0; e < P; ++e) { __builtin_nontemporal_store((double)e, q); q += TW; }
}

// tiled8 with occupancy capped by a dynamic LDS allocation (lds_bytes per 64-lane workgroup) and an optional stretch
// of arithmetic before each store (spin FMAs), to mimic a kernel that computes what it stores
__global__ void tiled8_occ(double *p, size_t batch, int P, int TW, int spin) {
    extern __shared__ double pad[];
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double *q = p + (b / TW) * (size_t)P * TW + (b % TW);
    double x = (double)threadIdx.x;
    if (P < 0) pad[threadIdx.x] = x;  // keeps the allocation
    for (int e = 0; e < P; ++e) {
        for (int k = 0; k < spin; ++k) x = x * 1.0000001 + 0.5;
        *q = x + (double)e; q += TW;
    }
}

template <class F> float time_it(F f, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}

int main() {
    const size_t batch = 1 << 20; const int P = 900;
    const size_t n = batch * P; const double gb = n * 8.0 / 1e9;
    double *p; CK(hipMalloc(&p, n * 8));
    for (int blk : {64, 256}) {
        for (int grid : {2048, 8192, 65536}) {
            float ms = time_it([&] { fill16<<<grid, blk>>>((double2 *)p, n / 2); }, 10);
            printf("fill16 grid %6d block %3d : %7.1f GB/s\n", grid, blk, gb / ms * 1e3);
            ms = time_it([&] { fill8<<<grid, blk>>>(p, n); }, 10);
            printf("fill8  grid %6d block %3d : %7.1f GB/s\n", grid, blk, gb / ms * 1e3);
        }
        float ms = time_it([&] { planes8<<<batch / blk, blk>>>(p, batch, P); }, 10);
        printf("planes8  block %3d : %7.1f GB/s\n", blk, gb / ms * 1e3);
        ms = time_it([&] { planes16<<<batch / 2 / blk, blk>>>((double2 *)p, batch / 2, P); }, 10);
        printf("planes16 block %3d : %7.1f GB/s\n", blk, gb / ms * 1e3);
        for (int tw : {64, 256, 4096}) {
            ms = time_it([&] { tiled8<<<batch / blk, blk>>>(p, batch, P, tw); }, 10);
            printf("tiled8 TW %4d block %3d : %7.1f GB/s\n", tw, blk, gb / ms * 1e3);
            ms = time_it([&] { tiled8_nt<<<batch / blk, blk>>>(p, batch, P, tw); }, 10);
            printf("tiled8_nt TW %4d block %3d : %7.1f GB/s\n", tw, blk, gb / ms * 1e3);
        }
        if (blk == 64) {
            for (int ldsb : {0, 5 * 1024, 10 * 1024, 20 * 1024}) {  // 160 KB / ldsb workgroups (= waves) per CU
                for (int spin : {0, 2, 8}) {
                    ms = time_it([&] { hipLaunchKernelGGL(tiled8_occ, dim3(batch / 64), dim3(64), ldsb, 0, p, batch, P, 4096, spin); }, 10);
                    printf("tiled8_occ TW 4096 lds %5d spin %d : %7.1f GB/s\n", ldsb, spin, gb / ms * 1e3);
                }
            }
        }
        ms = time_it([&] { fill16_nt<<<8192, blk>>>((double2 *)p, n / 2); }, 10);
        printf("fill16_nt grid   8192 block %3d : %7.1f GB/s\n", blk, gb / ms * 1e3);
    }
    return 0;
}
