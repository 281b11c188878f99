// Micro-benchmark: write rate of many independent per-instance output streams (the instance-major rollout's X and Y
// rows): every wavefront serves 4 instances and appends `chunk` bytes to each of their two rows per round.
//   v0  one 16-byte store per lane, 16 lanes per instance (4 runs of 256 B per instruction)        [the kernel's flush]
//   v1  one instance per instruction: lanes 0..chunk/16-1 store 16 B each (one contiguous run)
//   v2  8-byte stores, one instance per instruction, 64 lanes = 512 B contiguous
//   v3  as v0 with rows and chunks aligned to 128 B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(64) void k(double *X, double *Y, long long row, int chunk_d, int rounds, long long lead) {
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    const long long b0 = (long long)blockIdx.x * 4;
    for (int c = 0; c < rounds; ++c) {
        const long long off = lead + (long long)c * chunk_d;
        if (V == 0 || V == 3) {
            double *gx = X + (b0 + g) * row + off, *gy = Y + (b0 + g) * row + off;
            for (int kk = 0; kk * 32 < chunk_d; ++kk) {
                int e = i + kk * 16; if (2 * e >= chunk_d) e = chunk_d / 2 - 1;
                *(v2d *)(gx + 2 * e) = (v2d){(double)c, (double)e};
                *(v2d *)(gy + 2 * e) = (v2d){(double)c, (double)e};
            }
        } else if (V == 1) {
            for (int q = 0; q < 4; ++q) {
                double *gx = X + (b0 + q) * row + off, *gy = Y + (b0 + q) * row + off;
                for (int e = lane; 2 * e < chunk_d; e += 64) {
                    *(v2d *)(gx + 2 * e) = (v2d){(double)c, (double)e};
                    *(v2d *)(gy + 2 * e) = (v2d){(double)c, (double)e};
                }
            }
        } else {
            for (int q = 0; q < 4; ++q) {
                double *gx = X + (b0 + q) * row + off, *gy = Y + (b0 + q) * row + off;
                for (int e = lane; e < chunk_d; e += 64) { gx[e] = (double)c; gy[e] = (double)e; }
            }
        }
    }
}

template <int V>
void run(const char *name, double *X, double *Y, int batch, long long row, int chunk_d, int rounds, long long lead) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k<V>, dim3(batch / 4), dim3(64), 0, 0, X, Y, row, chunk_d, rounds, lead);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    const double bytes = 2.0 * 8.0 * chunk_d * rounds * batch;
    printf("%-4s chunk %5d B rows %6lld B lead %3lld B: %.4f ms  %7.1f GB/s\n", name, chunk_d * 8, row * 8, lead * 8, ms, bytes / ms / 1e6);
}

int main() {
    const int batch = 16384;
    const long long rowmax = 2432 * 2;  // doubles
    double *X, *Y;
    CK(hipMalloc(&X, sizeof(double) * rowmax * batch)); CK(hipMalloc(&Y, sizeof(double) * rowmax * batch));
    for (int chunk : {96, 192, 384}) {
        const int rounds = 2400 / chunk;
        run<0>("v0", X, Y, batch, 2412, chunk, rounds, 12);
        run<3>("v3", X, Y, batch, 2432, chunk, rounds, 0);
        run<1>("v1", X, Y, batch, 2412, chunk, rounds, 12);
        run<1>("v1a", X, Y, batch, 2432, chunk, rounds, 0);
        run<2>("v2", X, Y, batch, 2412, chunk, rounds, 12);
        run<2>("v2a", X, Y, batch, 2432, chunk, rounds, 0);
    }
    return 0;
}
