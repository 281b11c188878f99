// Condensed Hessian as a dense contraction on the FP64 matrix cores:  H = Gamma^T (W4 Gamma) + W3.
//
// One workgroup per instance.  Gamma is never read from memory: its columns are shifted copies of the Markov
// vectors, so the workgroup rebuilds the Markov sequence G_k = C A^k B in LDS (the O(f) recurrence of the build
// kernel) and every MFMA operand is one LDS read at a shifted index with a zero mask.  With W4 = blockdiag(Q) the
// other operand, T = W4 Gamma, has the same structure over the vectors Q G_k.
//
// Tiling: a wavefront owns a 48 x 48 tile of H as 3 x 3 accumulators of v_mfma_f64_16x16x4_f64 and walks the
// contraction index rho in steps of 4, starting at the first row where both operand tiles can be non-zero (Gamma
// is block lower-triangular: column block k is zero above row k r).  The product is formed transposed
// (D[i][j] = H[a0 + j][b0 + i]) so that the 16 lanes sharing an accumulator register hold 16 consecutive rows of
// one column of H and the epilogue stores contiguous 128-byte runs.
//
// Reference math: src/control/controller.rs:92-93 (tmp1 = M^T W4 M; H = tmp1 + W3), :46-90 for the structure of M.
#include "mpcx_internal.h"

namespace mpcx {
namespace {

constexpr int kThreads = 256;
constexpr int kTrLd = 17;  // leading dimension of the transpose scratch (odd: conflict-free both ways)
using d4 = __attribute__((ext_vector_type(4))) double;

struct MfLds {
    int A, B, C, Q, W, G, S, QG, QS, Tr, total;  // offsets in doubles
};

__host__ __device__ inline MfLds mf_lds_layout(int n, int m, int r, int f, int v) {
    MfLds L;
    int o = 0;
    const int fr = f * r, nl = f - v + 1;
    L.A = o; o += n * n;
    L.B = o; o += n * m;
    L.C = o; o += r * n;
    L.Q = o; o += r * r;
    L.W = o; o += 2 * n * m;
    L.G = o; o += m * fr;       // [c][k*r + a] = G_k[a][c]
    L.S = o; o += m * nl * r;   // cumulated blocks of the held-input column
    L.QG = o; o += m * fr;      // Q G_k
    L.QS = o; o += m * nl * r;  // Q S_j
    L.Tr = o; o += (kThreads / 64) * kTrLd * 16;  // one 16 x 16 block per wavefront (mirrored stores of a symmetric H)
    L.total = o;
    return L;
}

// Column `col` of Gamma (or of T = blockdiag(Q) Gamma when vec/cum are the Q-multiplied vectors) as an LDS vector
// plus the row where it starts: element rho is base[rho - shift] for rho >= shift, zero above.
struct ColRef {
    const double *base;
    int shift;  // first non-zero row; a column beyond the matrix gets shift = INT_MAX / 2 (all zero)
};

__device__ inline ColRef col_ref(int col, int VM, int m, int r, int v, int FR, int nl, const double *vec, const double *cum) {
    ColRef c;
    if (col >= VM) {
        c.base = vec;
        c.shift = 0x3fffffff;
        return c;
    }
    const int blk = col / m, cc = col - blk * m;
    c.shift = blk * r;
    c.base = (blk == v - 1) ? (cum + cc * nl * r) : (vec + cc * FR);
    return c;
}

__device__ inline double col_at(const ColRef &c, int rho, int FR) {
    const int idx = rho - c.shift;
    const bool ok = (idx >= 0) && (rho < FR);
    const double x = c.base[ok ? idx : 0];
    return ok ? x : 0.0;
}

// *flag (set to 1 by the caller) is cleared when Q (r x r) or W3 (vm x vm) is not exactly symmetric.
__global__ __launch_bounds__(256) void symmetry_kernel(const double *__restrict__ Q, int r, const double *__restrict__ W3, int vm,
                                                       int *flag) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, nq = (long long)r * r, nw = (long long)vm * vm;
    bool bad = false;
    if (idx < nq) {
        const int i = (int)(idx % r), j = (int)(idx / r);
        bad = i < j && !(Q[idx] == Q[(long long)i * r + j]);
    } else if (idx < nq + nw) {
        const long long e = idx - nq;
        const int i = (int)(e % vm), j = (int)(e / vm);
        bad = i < j && !(W3[e] == W3[(long long)i * vm + j]);
    }
    if (bad) *flag = 0;
}

__global__ __launch_bounds__(kThreads) void hessian_mfma_kernel(QpArgs p) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const long long b = blockIdx.x;
    const int n = p.n, m = p.m, r = p.r, f = p.f, v = p.v;
    const int FR = f * r, VM = v * m, nl = f - v + 1;
    const MfLds L = mf_lds_layout(n, m, r, f, v);
    double *sA = lds + L.A, *sB = lds + L.B, *sC = lds + L.C, *sQ = lds + L.Q, *sW = lds + L.W;
    double *sG = lds + L.G, *sS = lds + L.S, *sQG = lds + L.QG, *sQS = lds + L.QS;
    // Q and W3 symmetric (measured on the device by the launcher): H is symmetric, only the tiles on and below the
    // diagonal are computed and the others are their mirror images
    const bool sym = p.w3win != nullptr && ((const int *)p.w3win)[2] != 0;

    const double *gA = p.A + b * (long long)(n * n);
    const double *gB = p.B + b * (long long)(n * m);
    const double *gC = p.C + b * (long long)(r * n);
    for (int e = tid; e < n * n; e += kThreads) sA[e] = gA[e];
    for (int e = tid; e < n * m; e += kThreads) { const double x = gB[e]; sB[e] = x; sW[e] = x; }
    for (int e = tid; e < r * n; e += kThreads) sC[e] = gC[e];
    for (int e = tid; e < r * r; e += kThreads) sQ[e] = p.W4[e];
    lds_barrier();

    // Markov recurrence: G_k = C W_k, W_(k+1) = A W_k, W_0 = B
    const int nG = r * m, nW = n * m;
    for (int k = 0; k < f; ++k) {
        const double *Wc = sW + (k & 1) * nW;
        double *Wn = sW + ((k + 1) & 1) * nW;
        for (int t = tid; t < nG + nW; t += kThreads) {
            if (t < nG) {
                const int a = t % r, c = t / r;
                double s = 0.0;
                for (int j = 0; j < n; ++j) s += sC[j * r + a] * Wc[c * n + j];
                sG[c * FR + k * r + a] = s;
            } else {
                const int u = t - nG, i = u % n, c = u / n;
                double s = 0.0;
                for (int j = 0; j < n; ++j) s += sA[j * n + i] * Wc[c * n + j];
                Wn[u] = s;
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
    for (int t = tid; t < m * FR; t += kThreads) {
        const int c = t / FR, ka = t % FR, k = ka / r, a = ka % r;
        double s = 0.0;
        for (int q = 0; q < r; ++q) s += sQ[q * r + a] * sG[c * FR + k * r + q];
        sQG[t] = s;
    }
    for (int t = tid; t < m * nl * r; t += kThreads) {
        const int c = t / (nl * r), ja = t % (nl * r), j = ja / r, a = ja % r;
        double s = 0.0;
        for (int q = 0; q < r; ++q) s += sQ[q * r + a] * sS[c * nl * r + j * r + q];
        sQS[t] = s;
    }
    lds_barrier();

    // ---- kT x kT tiles of H (kT = 16 kU), one per wavefront at a time: kU + kU operand reads feed kU * kU MFMAs ----
    constexpr int kU = 3, kT = 16 * kU;
    const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int nt = (VM + kT - 1) / kT;
    double *gH = p.H + b * (long long)VM * VM;
    double *sTr = lds + L.Tr + wave * (kTrLd * 16);
    const int ntiles = sym ? nt * (nt + 1) / 2 : nt * nt;
    for (int tile = wave; tile < ntiles; tile += kThreads / 64) {
        int ta, tb;
        if (sym) {  // tile = ta (ta + 1) / 2 + tb with tb <= ta
            ta = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
            while (ta * (ta + 1) / 2 > tile) --ta;
            while ((ta + 1) * (ta + 2) / 2 <= tile) ++ta;
            tb = tile - ta * (ta + 1) / 2;
        } else {
            tb = tile / nt;  // consecutive waves walk down a column of tiles
            ta = tile - tb * nt;
        }
        const int a0 = ta * kT, b0 = tb * kT;
        // A operand rows = columns b0.. of T, B operand columns = columns a0.. of Gamma (transposed product)
        ColRef tcol[kU], gcol[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            tcol[u] = col_ref(b0 + 16 * u + l15, VM, m, r, v, FR, nl, sQG, sQS);
            gcol[u] = col_ref(a0 + 16 * u + l15, VM, m, r, v, FR, nl, sG, sS);
        }
        // first row where both operand tiles can be non-zero: the larger of the two tiles' smallest shifts
        const int sa = (a0 / m) * r, sb = (b0 / m) * r;
        const int kstart = (sa > sb ? sa : sb) & ~3;
        // The accumulators start from this tile's share of W3 (H = Gamma^T W4 Gamma + W3): its loads are in flight during
        // the first contraction steps instead of standing between the last one and the stores.
        d4 acc[kU][kU];
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
            for (int w = 0; w < kU; ++w) {
                const int row = a0 + 16 * w + l15;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int col = b0 + 16 * u + kq + 4 * reg;
                    acc[u][w][reg] = (row < VM && col < VM) ? p.W3[(long long)col * VM + row] : 0.0;
                }
            }
        // Three stretches of the contraction index.  Up to the row where the last column of either operand tile has
        // started (kfull) an operand may still be a structural zero: masked reads.  From there to the last whole step
        // every operand is a plain element of its vector: one LDS read at pointer + rho, no compare, no select (the
        // index arithmetic of the masked form costs more issue slots than the matrix instructions it feeds).  A tile
        // that hangs over the edge of H (columns >= v m) stays masked throughout.
        auto masked_steps = [&](int from, int to) {
            for (int rho0 = from; rho0 < to; rho0 += 4) {
                const int rho = rho0 + kq;
                double av[kU], bv[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    av[u] = col_at(tcol[u], rho, FR);
                    bv[u] = col_at(gcol[u], rho, FR);
                }
#pragma unroll
                for (int u = 0; u < kU; ++u)
#pragma unroll
                    for (int w = 0; w < kU; ++w) acc[u][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[w], acc[u][w], 0, 0, 0);
            }
        };
        const bool inside = (a0 + kT <= VM) && (b0 + kT <= VM);
        const int lasta = ((a0 + kT - 1) / m) * r, lastb = ((b0 + kT - 1) / m) * r;  // shifts of the tiles' last columns
        int kfull = ((lasta > lastb ? lasta : lastb) + 3) & ~3;
        const int kwhole = FR & ~3;
        if (!inside || kfull > kwhole) kfull = FR;  // all masked
        masked_steps(kstart, kfull < FR ? kfull : FR);
        if (kfull < kwhole) {
            const double *pa[kU], *pb[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                pa[u] = tcol[u].base - tcol[u].shift + kq;
                pb[u] = gcol[u].base - gcol[u].shift + kq;
            }
#pragma unroll 2
            for (int rho0 = kfull; rho0 < kwhole; rho0 += 4) {
                double av[kU], bv[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    av[u] = pa[u][rho0];
                    bv[u] = pb[u][rho0];
                }
#pragma unroll
                for (int u = 0; u < kU; ++u)
#pragma unroll
                    for (int w = 0; w < kU; ++w) acc[u][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[w], acc[u][w], 0, 0, 0);
            }
            masked_steps(kwhole, FR);
        }
        // D[i][j] = H[a0 + 16 w + j][b0 + 16 u + i]; lane holds j = l15 and i = kq + 4 reg
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
            for (int w = 0; w < kU; ++w) {
                const int row = a0 + 16 * w + l15;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int col = b0 + 16 * u + kq + 4 * reg;
                    if (row < VM && col < VM) gH[(long long)col * VM + row] = acc[u][w][reg];
                }
                if (sym && ta != tb) {
                    // the mirror image H[col][row]: the block goes through LDS so that the 16 lanes that share a register
                    // hold 16 consecutive rows of the mirrored column again (128-byte runs)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) sTr[l15 * kTrLd + kq + 4 * reg] = acc[u][w][reg];  // [j][i]
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int mrow = b0 + 16 * u + l15;  // = col index of the original
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int jj = kq + 4 * reg, mcol = a0 + 16 * w + jj;
                        const double x = sTr[jj * kTrLd + l15];
                        if (mrow < VM && mcol < VM) gH[(long long)mcol * VM + mrow] = x;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
    }
}

}  // namespace

// H only (Phi, Gamma, fvec come from the build kernels).  Needs instance-major data and a block weight W4.
int launch_hessian_mfma(const QpArgs &a, hipStream_t s) {
    if (a.W != 1) return MPCX_ERR_LAYOUT;
    if (a.weights_mode != MPCX_WEIGHTS_BLOCK) return MPCX_ERR_UNSUPPORTED;
    const MfLds L = mf_lds_layout(a.n, a.m, a.r, a.f, a.v);
    const size_t lds_bytes = (size_t)L.total * sizeof(double);
    if (lds_bytes > 160 * 1024) return MPCX_ERR_UNSUPPORTED;
    if (a.batch > 0x7fffffffLL || (long long)a.v * a.m > (1 << 20)) return MPCX_ERR_UNSUPPORTED;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)hessian_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    QpArgs b = a;
    if (a.w3win && a.w3win_bytes >= 3 * sizeof(int)) {
        int *flag = (int *)a.w3win + 2;
        hipError_t e = hipMemsetD32Async((hipDeviceptr_t)flag, 1, 1, s);
        if (e != hipSuccess) return (int)e;
        const long long cells = (long long)a.r * a.r + (long long)a.v * a.m * a.v * a.m;
        hipLaunchKernelGGL(symmetry_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, a.W4, a.r, a.W3, a.v * a.m, flag);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    } else {
        b.w3win = nullptr;
    }
    hipLaunchKernelGGL(hessian_mfma_kernel, dim3((unsigned)a.batch), dim3(kThreads), lds_bytes, s, b);
    return (int)hipGetLastError();
}

}  // namespace mpcx
