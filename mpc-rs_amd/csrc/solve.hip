// Batched solve with the condensed Hessian: X = +-H^-1 R per instance, by LU with partial pivoting on the
// augmented matrix [H | R] held in LDS, one wavefront per instance.
//
// Reference: src/control/controller.rs:93-96 (`(tmp1 + W3).try_inverse()` then `tmp2 * M^T * W4`: the gain matrix) and
// :130-132 (u_seq = gain * s, first move applied).  nalgebra's try_inverse is an LU with partial pivoting that fails on
// an exactly zero pivot; the same test (plus non-finite pivots) sets the per-instance `info` flag here, which is how
// the batched call reports the reference's Err("Matrix inversion failed") without aborting the batch.
//   solve:  R = fvec (one column),        X = -H^-1 fvec = the optimal input sequence
//   gain:   R = Gamma^T blockdiag(Q),     X = H^-1 Gamma^T W4 = the reference's gain_matrix
#include "mpcx_internal.h"

namespace mpcx {
namespace {

constexpr int kWave = 64;

struct SolveArgs {
    const double *H, *R;
    double *X;
    int *info;
    long long batch, W;
    int nrow, nrhs;
    int negate;
};

__global__ __launch_bounds__(kWave) void solve_kernel(SolveArgs p) {
    extern __shared__ double aug[];  // column-major [col][row], leading dimension ld (odd)
    const int lane = threadIdx.x;
    const long long b = blockIdx.x;
    const int nr = p.nrow, nc = p.nrow + p.nrhs, ld = nr | 1;
    const long long EH = (long long)nr * nr, ER = (long long)nr * p.nrhs;
    for (int e = lane; e < nr * nr; e += kWave) aug[(e / nr) * ld + (e % nr)] = p.H[elem_index(b, e, EH, p.W)];
    for (int e = lane; e < nr * p.nrhs; e += kWave) aug[(nr + e / nr) * ld + (e % nr)] = p.R[elem_index(b, e, ER, p.W)];
    lds_barrier();

    bool failed = false;
    for (int k = 0; k < nr; ++k) {
        // pivot: largest magnitude in column k at or below the diagonal
        double best = -1.0;
        int brow = k;
        for (int i = k + lane; i < nr; i += kWave) {
            const double a = fabs(aug[k * ld + i]);
            if (a > best || !(a == a)) { best = (a == a) ? a : INFINITY; brow = i; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off);
            const int orow = __shfl_xor(brow, off);
            if (ob > best || (ob == best && orow < brow)) { best = ob; brow = orow; }
        }
        const double piv = aug[k * ld + brow];
        if (piv == 0.0 || !isfinite(piv)) { failed = true; break; }  // wave-uniform
        if (brow != k) {
            for (int j = k + lane; j < nc; j += kWave) {
                const double t = aug[j * ld + k];
                aug[j * ld + k] = aug[j * ld + brow];
                aug[j * ld + brow] = t;
            }
        }
        lds_barrier();
        const double inv_p = 1.0 / piv;
        for (int i = k + 1 + lane; i < nr; i += kWave) aug[k * ld + i] *= inv_p;  // multipliers
        lds_barrier();
        // trailing update, right-hand sides included: lanes along the rows, columns in turn
        for (int j = k + 1; j < nc; ++j) {
            const double akj = aug[j * ld + k];
            for (int i = k + 1 + lane; i < nr; i += kWave) aug[j * ld + i] -= aug[k * ld + i] * akj;
        }
        lds_barrier();
    }
    if (!failed) {
        // back substitution on the right-hand sides
        for (int k = nr - 1; k >= 0; --k) {
            const double inv_u = 1.0 / aug[k * ld + k];
            for (int c = lane; c < p.nrhs; c += kWave) aug[(nr + c) * ld + k] *= inv_u;
            lds_barrier();
            if (p.nrhs >= 16) {  // lanes along the right-hand sides
                for (int c = lane; c < p.nrhs; c += kWave) {
                    const double x = aug[(nr + c) * ld + k];
                    for (int i = 0; i < k; ++i) aug[(nr + c) * ld + i] -= aug[k * ld + i] * x;
                }
            } else {  // lanes along the rows
                for (int c = 0; c < p.nrhs; ++c) {
                    const double x = aug[(nr + c) * ld + k];
                    for (int i = lane; i < k; i += kWave) aug[(nr + c) * ld + i] -= aug[k * ld + i] * x;
                }
            }
            lds_barrier();
        }
    }
    const double sign = p.negate ? -1.0 : 1.0;
    for (int e = lane; e < nr * p.nrhs; e += kWave) {
        const double x = aug[(nr + e / nr) * ld + (e % nr)];
        p.X[elem_index(b, e, ER, p.W)] = failed ? __longlong_as_double(0x7ff8000000000000LL) : sign * x;
    }
    if (p.info && lane == 0) p.info[b] = failed ? 1 : 0;
}

// R[c][(i, a2)] = sum_a Gamma[(i, a)][c] Q[a][a2]  (Gamma^T blockdiag(Q)), written to `out` in the batch layout
__global__ __launch_bounds__(256) void gamma_t_q_kernel(const double *Gamma, const double *Q, double *out, long long batch,
                                                        long long W, int m, int r, int f, int v) {
    const int VM = v * m, FR = f * r;
    const long long per = (long long)VM * FR;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= batch * per) return;
    // instance fastest within a tile would need W; keep it simple: idx = b * per + e
    const long long b = idx / per, e = idx - b * per;
    const int col = (int)(e / VM), c = (int)(e - (long long)col * VM);  // out is VM x FR: element (c, col)
    const int i = col / r, a2 = col - i * r;
    double s = 0.0;
    for (int a = 0; a < r; ++a) s += Gamma[elem_index(b, (long long)c * FR + i * r + a, (long long)FR * VM, W)] * Q[a2 * r + a];
    out[elem_index(b, e, per, W)] = s;
}

int launch_solve(const SolveArgs &a, hipStream_t s) {
    const int ld = a.nrow | 1;
    const size_t lds_bytes = (size_t)ld * (a.nrow + a.nrhs) * sizeof(double);
    if (lds_bytes > 160 * 1024 || a.batch > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(solve_kernel, dim3((unsigned)a.batch), dim3(kWave), lds_bytes, s, a);
    return (int)hipGetLastError();
}

}  // namespace

int launch_qp_solve(const double *H, const double *fvec, double *u_seq, int *info, long long batch, long long W, int vm,
                    hipStream_t s) {
    SolveArgs a{H, fvec, u_seq, info, batch, W, vm, 1, 1};
    return launch_solve(a, s);
}

int launch_qp_gain(const double *H, const double *Gamma, const double *Q, double *gain, int *info, long long batch,
                   long long W, int m, int r, int f, int v, hipStream_t s) {
    const long long total = batch * (long long)(v * m) * (f * r);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    {   // the augmented system must fit LDS before anything is launched
        const int nr = v * m, ld = nr | 1;
        if ((size_t)ld * (nr + f * r) * sizeof(double) > 160 * 1024) return MPCX_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(gamma_t_q_kernel, dim3((unsigned)blocks), dim3(256), 0, s, Gamma, Q, gain, batch, W, m, r, f, v);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    SolveArgs a{H, gain, gain, info, batch, W, v * m, f * r, 0};  // in place: R is read into LDS before X is written
    return launch_solve(a, s);
}

}  // namespace mpcx
