// Condensed-QP build for plants too large for one lane: ONE WORKGROUP PER INSTANCE, instance-major layout.
//
// A, B, C of the instance are staged in LDS.  The Markov blocks G_k = C A^k B and the Phi blocks C A^(k+1) come
// from two O(f) recurrences that advance together, one barrier per step.  Gamma is then a set of shifted copies
// of the Markov sequence (kept in LDS as one vector per input column), written column by column with lanes
// along the rows, so every store instruction covers a contiguous run of a column.  H uses the block-Toeplitz
// structure of Gamma: with W4 = blockdiag(Q),
//     H[a][b] = H[a+1][b+1] + G_(f-1-a)^T Q G_(f-1-b)        for block indices a, b <= v-3,
// so the workgroup sweeps the column blocks from right to left, each thread owning rows of H; every step
// writes one whole column block of H with lanes along the rows.  The blocks that touch the last two block
// rows / columns are summed directly (they involve the held-input column S_j = G_0 + ... + G_j when f > v).
//
// Reference math: src/control/controller.rs:32-44 (Phi), :46-90 (Gamma), :92-93 (H), :96 and :125-127 (fvec).
#include "mpcx_internal.h"

namespace mpcx {
namespace {

constexpr int kWgBlock = 256;

struct WgLds {
    // offsets in doubles into the dynamic LDS buffer
    int A, B, C, x0, Q, Wst, Pst, G, S, QG, QS, Sr, Z, Hp, total;
};

__host__ __device__ inline WgLds wg_lds_layout(int n, int m, int r, int f, int v) {
    WgLds L;
    int o = 0;
    const int fr = f * r, nl = f - v + 1;
    L.A = o; o += n * n;
    L.B = o; o += n * m;
    L.C = o; o += r * n;
    L.x0 = o; o += n;
    L.Q = o; o += r * r;
    L.Wst = o; o += 2 * n * m;   // Markov state, ping-pong
    L.Pst = o; o += 2 * r * n;   // Phi state, ping-pong
    o = (o + 1) & ~1;
    L.G = o; o += m * fr;        // [c][k*r + a] = G_k[a][c]
    L.S = o; o += m * nl * r;    // [c][j*r + a] = S_j[a][c]
    L.QG = o; o += m * fr;       // Q G_k, same indexing as G
    L.QS = o; o += m * nl * r;   // Q S_j
    L.Sr = o; o += fr;           // Phi x0 - ref
    L.Z = o; o += fr;            // Q (Phi x0 - ref)
    L.Hp = o; o += 2 * v * m * m;  // previous column block of H, ping-pong: [row][j]
    L.total = o;
    return L;
}

__global__ __launch_bounds__(kWgBlock) void qp_wg_kernel(QpArgs p) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const long long b = blockIdx.x;
    const int n = p.n, m = p.m, r = p.r, f = p.f, v = p.v;
    const int FR = f * r, VM = v * m, nl = f - v + 1;
    const WgLds L = wg_lds_layout(n, m, r, f, v);
    double *sA = lds + L.A, *sB = lds + L.B, *sC = lds + L.C, *sx0 = lds + L.x0, *sQ = lds + L.Q;
    double *sW = lds + L.Wst, *sP = lds + L.Pst, *sG = lds + L.G, *sS = lds + L.S, *sQG = lds + L.QG;
    double *sQS = lds + L.QS, *sSr = lds + L.Sr, *sZ = lds + L.Z, *sHp = lds + L.Hp;

    const double *gA = p.A + b * (long long)(n * n);
    const double *gB = p.B + b * (long long)(n * m);
    const double *gC = p.C + b * (long long)(r * n);
    const bool need_q = p.H || p.fvec;

    for (int e = tid; e < n * n; e += kWgBlock) sA[e] = gA[e];
    for (int e = tid; e < n * m; e += kWgBlock) { sB[e] = gB[e]; sW[e] = gB[e]; }
    for (int e = tid; e < r * n; e += kWgBlock) { sC[e] = gC[e]; sP[e] = gC[e]; }
    if (p.fvec) for (int e = tid; e < n; e += kWgBlock) sx0[e] = p.x0[b * n + e];
    if (need_q) for (int e = tid; e < r * r; e += kWgBlock) sQ[e] = p.W4[e];
    __syncthreads();

    // ---- recurrences: step k produces G_k (from W_k), W_(k+1) = A W_k, P_(k+1) = P_k A (P_0 = C) ----
    const int nG = r * m, nW = n * m, nP = r * n;
    double *gPhi = p.Phi ? p.Phi + b * (long long)FR * n : nullptr;
    const double *gref = p.fvec ? p.ref + b * (long long)FR : nullptr;
    for (int k = 0; k < f; ++k) {
        const double *Wc = sW + (k & 1) * nW;
        double *Wn = sW + ((k + 1) & 1) * nW;
        const double *Pc = sP + (k & 1) * nP;
        double *Pn = sP + ((k + 1) & 1) * nP;
        for (int t = tid; t < nG + nW + nP; t += kWgBlock) {
            if (t < nG) {  // G_k[a][c] = sum_j C[a][j] W_k[j][c]
                const int a = t % r, c = t / r;
                double s = 0.0;
                for (int j = 0; j < n; ++j) s += sC[j * r + a] * Wc[c * n + j];
                sG[c * FR + k * r + a] = s;
            } else if (t < nG + nW) {  // W_(k+1)[i][c] = sum_j A[i][j] W_k[j][c]
                const int u = t - nG, i = u % n, c = u / n;
                double s = 0.0;
                for (int j = 0; j < n; ++j) s += sA[j * n + i] * Wc[c * n + j];
                Wn[u] = s;
            } else {  // P_(k+1)[a][j] = sum_i P_k[a][i] A[i][j]   (= Phi block k)
                const int u = t - nG - nW, a = u % r, j = u / r;
                double s = 0.0;
                for (int i = 0; i < n; ++i) s += Pc[i * r + a] * sA[j * n + i];
                Pn[u] = s;
                if (gPhi) gPhi[(long long)j * FR + k * r + a] = s;
            }
        }
        __syncthreads();
        if (p.fvec) {  // s_k = Phi_k x0 - ref_k (multiplied by Q after the loop)
            for (int a = tid; a < r; a += kWgBlock) {
                double s = 0.0;
                for (int j = 0; j < n; ++j) s += Pn[j * r + a] * sx0[j];
                sSr[k * r + a] = s - gref[k * r + a];
            }
        }
    }
    __syncthreads();

    // ---- S_j = G_0 + ... + G_j (j < nl), Q G_k, Q S_j, Z = Q s ----
    if (need_q || p.Gamma) {
        for (int t = tid; t < nG; t += kWgBlock) {
            const int a = t % r, c = t / r;
            double run = 0.0;
            for (int j = 0; j < nl; ++j) {
                run += sG[c * FR + j * r + a];
                sS[c * nl * r + j * r + a] = run;
            }
        }
    }
    __syncthreads();
    if (need_q) {
        for (int t = tid; t < m * FR; t += kWgBlock) {  // QG[c][k*r+a] = sum_q Q[a][q] G[c][k*r+q]
            const int c = t / FR, ka = t % FR, k = ka / r, a = ka % r;
            double s = 0.0;
            for (int q = 0; q < r; ++q) s += sQ[q * r + a] * sG[c * FR + k * r + q];
            sQG[t] = s;
        }
        for (int t = tid; t < m * nl * r; t += kWgBlock) {
            const int c = t / (nl * r), ja = t % (nl * r), j = ja / r, a = ja % r;
            double s = 0.0;
            for (int q = 0; q < r; ++q) s += sQ[q * r + a] * sS[c * nl * r + j * r + q];
            sQS[t] = s;
        }
        if (p.fvec) {
            for (int t = tid; t < FR; t += kWgBlock) {  // Z[i*r+a] = sum_q Q[a][q] s[i*r+q]
                const int i = t / r, a = t % r;
                double z = 0.0;
                for (int q = 0; q < r; ++q) z += sQ[q * r + a] * sSr[i * r + q];
                sZ[t] = z;
            }
        }
    }
    __syncthreads();

    // ---- Gamma: column (k, c) = [zeros(k r) ; Markov vector c] (or the cumulated vector for k = v-1) ----
    if (p.Gamma) {
        double *gG = p.Gamma + b * (long long)FR * VM;
        const int wave = tid >> 6, lane = tid & 63, nwaves = kWgBlock >> 6;
        for (int col = wave; col < VM; col += nwaves) {
            const int k = col / m, c = col % m;
            const int shift = k * r;
            const double *src = (k == v - 1) ? (sS + c * nl * r) : (sG + c * FR);
            double *dst = gG + (long long)col * FR;
            for (int row = lane; row < FR; row += 64) dst[row] = (row < shift) ? 0.0 : src[row - shift];
        }
    }

    // ---- H, swept by column block from v-1 down to 0 ----
    if (p.H) {
        double *gH = p.H + b * (long long)VM * VM;
        for (int bb = v - 1; bb >= 0; --bb) {
            double *cur = sHp + (bb & 1) * VM * m;
            const double *prev = sHp + ((bb + 1) & 1) * VM * m;
            for (int row = tid; row < VM; row += kWgBlock) {
                const int aa = row / m, i = row % m;
                for (int j = 0; j < m; ++j) {
                    double acc;
                    if (aa <= v - 3 && bb <= v - 3) {
                        acc = prev[(row + m) * m + j];
                        const double *ga = sG + i * FR + (f - 1 - aa) * r;
                        const double *qb = sQG + j * FR + (f - 1 - bb) * r;
                        for (int q = 0; q < r; ++q) acc += ga[q] * qb[q];
                    } else {
                        // direct sum over the rows i2 >= max(aa, bb) (and >= v-1 when a last block is involved)
                        acc = 0.0;
                        const bool la = (aa == v - 1), lb = (bb == v - 1);
                        const int i0 = (la || lb) ? (v - 1) : (aa > bb ? aa : bb);
                        for (int i2 = i0; i2 < f; ++i2) {
                            const double *ga = la ? (sS + i * nl * r + (i2 - (v - 1)) * r) : (sG + i * FR + (i2 - aa) * r);
                            const double *qb = lb ? (sQS + j * nl * r + (i2 - (v - 1)) * r) : (sQG + j * FR + (i2 - bb) * r);
                            for (int q = 0; q < r; ++q) acc += ga[q] * qb[q];
                        }
                    }
                    cur[row * m + j] = acc;
                    const long long e = (long long)(bb * m + j) * VM + row;
                    gH[e] = acc + p.W3[e];
                }
            }
            __syncthreads();
        }
    }

    // ---- fvec[col] = sum_rows Gamma[row][col] Z[row] ----
    if (p.fvec) {
        double *gf = p.fvec + b * (long long)VM;
        for (int col = tid; col < VM; col += kWgBlock) {
            const int k = col / m, c = col % m;
            const int shift = k * r;
            const double *src = (k == v - 1) ? (sS + c * nl * r) : (sG + c * FR);
            double a0 = 0.0, a1 = 0.0;
            int t = 0;
            const int len = FR - shift;
            for (; t + 1 < len; t += 2) {
                a0 += src[t] * sZ[shift + t];
                a1 += src[t + 1] * sZ[shift + t + 1];
            }
            if (t < len) a0 += src[t] * sZ[shift + t];
            gf[col] = a0 + a1;
        }
    }
}

}  // namespace

int launch_qp_wg(const QpArgs &a, int hessian_algo, hipStream_t s) {
    (void)hessian_algo;
    if (a.W != 1) return MPCX_ERR_LAYOUT;
    if (a.weights_mode != MPCX_WEIGHTS_BLOCK && (a.H || a.fvec)) return MPCX_ERR_UNSUPPORTED;
    const WgLds L = wg_lds_layout(a.n, a.m, a.r, a.f, a.v);
    const size_t lds_bytes = (size_t)L.total * sizeof(double);
    if (lds_bytes > 160 * 1024) return MPCX_ERR_UNSUPPORTED;
    if ((long long)a.f * a.r * a.v * a.m > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    if (a.batch > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)qp_wg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(qp_wg_kernel, dim3((unsigned)a.batch), dim3(kWgBlock), lds_bytes, s, a);
    return (int)hipGetLastError();
}

}  // namespace mpcx
