// Micro-benchmark: rate of v_mfma_f64_16x16x4_f64 for ONE wavefront per SIMD, as the held-input sums of qp_wg run it:
//   mode 0  four independent accumulator chains, operands in registers
//   mode 1  the same with eight LDS reads (8 bytes per lane, constant offsets) feeding every four products
//   mode 2  mode 1 while three more wavefronts per SIMD run dependent FP64 multiply-adds
//   mode 3  mode 1 while three more wavefronts per SIMD stream LDS reads
// One workgroup per CU (256 + 768 threads), time per product of the MFMA wavefronts printed in ns and cycles at 2.4 GHz;
// then modes 0 and 1 with two and four MFMA wavefronts per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_rate.hip -o tools/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
#ifndef ROWSTRIDE
#define ROWSTRIDE 24  // doubles between the operand rows of neighbouring lanes (24: as in qp_wg for r = 12; 4-way bank conflicts)
#endif

__global__ __launch_bounds__(1024) void probe(double *out, int rounds, int mode, int mfma_waves) {
    __shared__ double buf[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += blockDim.x) buf[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double res = 0.0;
    if (wave < mfma_waves) {  // mfma_waves / 4 MFMA wavefronts per SIMD (waves are dealt round-robin to the four SIMDs)
        d4 a0 = {0, 0, 0, 0}, a1 = a0, b0 = a0, b1 = a0;
        double x = 1.0 + lane * 1e-6, y = 1.0 - lane * 1e-6;
        const double *pa = buf + (lane & 15) * ROWSTRIDE + (lane >> 4), *pb = buf + 4096 + (lane & 15) * (ROWSTRIDE / 2) + (lane >> 4);
        for (int r = 0; r < rounds; ++r) {
            if (mode == 0) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, b0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a1, 0, 0, 0);
                b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, b1, 0, 0, 0);
            } else {
                const int k0 = (r & 63) * 8;
                const double x10 = pa[k0], x11 = pa[k0 + 4], y10 = pb[k0], y11 = pb[k0 + 4];
                const double x20 = pa[k0 + 1024], x21 = pa[k0 + 1028], y20 = pb[k0 + 1024], y21 = pb[k0 + 1028];
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x10, y10, a0, 0, 0, 0);
                b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x20, y20, b0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x11, y11, a1, 0, 0, 0);
                b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x21, y21, b1, 0, 0, 0);
            }
        }
        a0 += a1 + b0 + b1;
        res = a0[0] + a0[1] + a0[2] + a0[3];
    } else if (mode == 2) {
        double s0 = lane, s1 = 1.0, s2 = 2.0, s3 = 3.0;
        for (int r = 0; r < rounds * 8; ++r) {
            s0 = s0 * 1.0000001 + 0.5; s1 = s1 * 0.9999999 + 0.25; s2 = s2 * 1.0000002 + 0.125; s3 = s3 * 0.9999998 + 0.0625;
        }
        res = s0 + s1 + s2 + s3;
    } else if (mode == 3) {
        double s = 0.0;
        for (int r = 0; r < rounds * 4; ++r) s += buf[(lane + 64 * (r & 63)) & 8191] + buf[(lane * 2 + r) & 8191];
        res = s;
    }
    out[(size_t)blockIdx.x * blockDim.x + tid] = res;
}
int main() {
    double *out; CK(hipMalloc(&out, 256 * 1024 * sizeof(double)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int rounds = 20000;
    for (int mode = 0; mode < 4; ++mode)
        for (int threads : {256, 1024}) {
            if (mode < 2 && threads == 1024 && mode == 0) {}
            probe<<<256, threads>>>(out, rounds, mode, 4); CK(hipDeviceSynchronize());
            CK(hipEventRecord(a)); probe<<<256, threads>>>(out, rounds, mode, 4); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            const double ns = ms * 1e6 / (rounds * 4.0);
            printf("mode %d threads %4d : %7.1f ns per product of a wavefront = %6.1f cycles at 2.4 GHz\n", mode, threads, ns, ns * 2.4);
        }
    for (int mode = 0; mode < 2; ++mode)
        for (int mw : {8, 16}) {  // two and four MFMA wavefronts per SIMD: time per product of ONE wavefront; the SIMD's is that / (mw / 4)
            probe<<<256, 64 * mw>>>(out, rounds, mode, mw); CK(hipDeviceSynchronize());
            CK(hipEventRecord(a)); probe<<<256, 64 * mw>>>(out, rounds, mode, mw); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            const double ns = ms * 1e6 / (rounds * 4.0);
            printf("mode %d, %d MFMA wavefronts per SIMD : %7.1f ns per product of a wavefront, %6.1f ns per product of the SIMD\n", mode, mw / 4, ns, ns / (mw / 4));
        }
    return 0;
}
