// H and fvec with a DENSE output weight W4 ((f r) x (f r)), the literal argument of the reference's
// form_lifted_matrices (src/control/controller.rs:23, :92-96):  H = Gamma^T (W4 Gamma) + W3,  fvec = Gamma^T W4 (Phi x0 - ref).
//
// Compatibility path, one workgroup per instance, instance-major data.  A dense W4 destroys the block-Toeplitz
// structure the other kernels exploit, so H is formed column by column: t = W4 Gamma[:, col] (W4 streamed from L2,
// Gamma's column generated from the Markov vectors in LDS, its structural zeros skipped), then H[:, col] = Gamma^T t.
// Work is O(v m (f r)^2) per instance; the block-weight kernels are the fast path.
#include "mpcx_internal.h"

namespace mpcx {
namespace {

constexpr int kThreads = 256;

struct DwLds {
    int A, B, C, x0, Ax, W, P, G, S, Sr, Z, T, total;
};

__host__ __device__ inline DwLds dw_lds_layout(int n, int m, int r, int f, int v) {
    DwLds L;
    int o = 0;
    const int fr = f * r, nl = f - v + 1;
    L.A = o; o += n * n;
    L.B = o; o += n * m;
    L.C = o; o += r * n;
    L.x0 = o; o += n;
    L.Ax = o; o += n;
    L.W = o; o += 2 * n * m;
    L.P = o; o += 2 * r * n;
    L.G = o; o += m * fr;
    L.S = o; o += m * nl * r;
    L.Sr = o; o += fr;  // ref, then Phi x0 - ref
    L.Z = o; o += fr;   // W4 (Phi x0 - ref)
    L.T = o; o += fr;   // W4 Gamma[:, col]
    L.total = o;
    return L;
}

__global__ __launch_bounds__(kThreads) void qp_dense_kernel(QpArgs p) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const long long b = blockIdx.x;
    const int n = p.n, m = p.m, r = p.r, f = p.f, v = p.v;
    const int FR = f * r, VM = v * m, nl = f - v + 1;
    const DwLds L = dw_lds_layout(n, m, r, f, v);
    double *sA = lds + L.A, *sB = lds + L.B, *sC = lds + L.C, *sx0 = lds + L.x0, *sAx = lds + L.Ax, *sW = lds + L.W;
    double *sP = lds + L.P, *sG = lds + L.G, *sS = lds + L.S, *sSr = lds + L.Sr, *sZ = lds + L.Z, *sT = lds + L.T;
    const double *gA = p.A + b * (long long)(n * n), *gB = p.B + b * (long long)(n * m), *gC = p.C + b * (long long)(r * n);
    for (int e = tid; e < n * n; e += kThreads) sA[e] = gA[e];
    for (int e = tid; e < n * m; e += kThreads) { const double x = gB[e]; sB[e] = x; sW[e] = x; }
    for (int e = tid; e < r * n; e += kThreads) { const double x = gC[e]; sC[e] = x; sP[e] = x; }
    if (p.fvec) {
        for (int e = tid; e < n; e += kThreads) sx0[e] = p.x0[b * n + e];
        for (int e = tid; e < FR; e += kThreads) sSr[e] = p.ref[b * (long long)FR + e];
    }
    lds_barrier();
    if (p.fvec) {
        for (int i = tid; i < n; i += kThreads) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += sA[j * n + i] * sx0[j];
            sAx[i] = s;
        }
        lds_barrier();
    }
    // recurrences: G_k = C W_k, W <- A W, P <- P A, s_k = P_k (A x0) - ref_k
    const int nG = r * m, nW = n * m, nP = p.fvec ? r * n : 0, nS = p.fvec ? r : 0;
    for (int k = 0; k < f; ++k) {
        const double *Wc = sW + (k & 1) * nW, *Pc = sP + (k & 1) * (r * n);
        double *Wn = sW + ((k + 1) & 1) * nW, *Pn = sP + ((k + 1) & 1) * (r * n);
        for (int t = tid; t < nG + nW + nP + nS; t += kThreads) {
            double s = 0.0;
            if (t < nG) {
                const int a = t % r, c = t / r;
                for (int j = 0; j < n; ++j) s += sC[j * r + a] * Wc[c * n + j];
                sG[c * FR + k * r + a] = s;
            } else if (t < nG + nW) {
                const int u = t - nG, i = u % n, c = u / n;
                for (int j = 0; j < n; ++j) s += sA[j * n + i] * Wc[c * n + j];
                Wn[u] = s;
            } else if (t < nG + nW + nP) {
                const int u = t - nG - nW, a = u % r, j = u / r;
                for (int i = 0; i < n; ++i) s += Pc[i * r + a] * sA[j * n + i];
                Pn[u] = s;
            } else {
                const int a = t - nG - nW - nP;
                for (int i = 0; i < n; ++i) s += Pc[i * r + a] * sAx[i];
                sSr[k * r + a] = s - sSr[k * r + a];
            }
        }
        lds_barrier();
    }
    for (int t = tid; t < nG; t += kThreads) {
        const int a = t % r, c = t / r;
        double run = 0.0;
        for (int j = 0; j < nl; ++j) {
            run += sG[c * FR + j * r + a];
            sS[c * nl * r + j * r + a] = run;
        }
    }
    lds_barrier();

    // column `col` of Gamma: element rho is src[rho - shift] for rho >= shift, zero above
    auto gamma_col = [&](int col, const double *&src, int &shift) {
        const int blk = col / m, c = col - blk * m;
        shift = blk * r;
        src = (blk == v - 1) ? (sS + c * nl * r) : (sG + c * FR);
    };

    if (p.fvec) {
        for (int t = tid; t < FR; t += kThreads) {  // z = W4 s
            double z = 0.0;
            for (int rho = 0; rho < FR; ++rho) z += p.W4[(long long)rho * FR + t] * sSr[rho];
            sZ[t] = z;
        }
        lds_barrier();
        double *gf = p.fvec + b * (long long)VM;
        for (int col = tid; col < VM; col += kThreads) {
            const double *src;
            int shift;
            gamma_col(col, src, shift);
            double acc = 0.0;
            for (int rho = shift; rho < FR; ++rho) acc += src[rho - shift] * sZ[rho];
            gf[col] = acc;
        }
    }
    if (p.H) {
        double *gH = p.H + b * (long long)VM * VM;
        for (int col = 0; col < VM; ++col) {
            const double *src;
            int shift;
            gamma_col(col, src, shift);
            for (int t = tid; t < FR; t += kThreads) {  // t = W4 Gamma[:, col], zeros of the column skipped
                double acc = 0.0;
                for (int rho = shift; rho < FR; ++rho) acc += p.W4[(long long)rho * FR + t] * src[rho - shift];
                sT[t] = acc;
            }
            lds_barrier();
            for (int a = tid; a < VM; a += kThreads) {  // H[a][col] = Gamma[:, a] . t
                const double *sa;
                int sha;
                gamma_col(a, sa, sha);
                double acc = 0.0;
                for (int rho = sha; rho < FR; ++rho) acc += sa[rho - sha] * sT[rho];
                const long long e = (long long)col * VM + a;
                gH[e] = acc + p.W3[e];
            }
            lds_barrier();
        }
    }
}

}  // namespace

// H and / or fvec for a dense W4; Phi and Gamma are produced by the build kernels.
int launch_qp_dense(const QpArgs &a, hipStream_t s) {
    if (a.W != 1) return MPCX_ERR_LAYOUT;
    const DwLds L = dw_lds_layout(a.n, a.m, a.r, a.f, a.v);
    const size_t lds_bytes = (size_t)L.total * sizeof(double);
    if (lds_bytes > 160 * 1024 || a.batch > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)qp_dense_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(qp_dense_kernel, dim3((unsigned)a.batch), dim3(kThreads), lds_bytes, s, a);
    return (int)hipGetLastError();
}

}  // namespace mpcx
