// Condensed-QP build for tiny plants: ONE LANE PER INSTANCE.
//
// For n <= 6 and one or two inputs/outputs an instance's whole working set fits a lane's registers plus a
// few dozen LDS words, so 64 instances ride one wavefront with no idle lanes and no cross-lane traffic.  With
// a batch-interleaved layout (W >= 64) lane l of a wave touches address base + l for every element, so every
// global access of the wave is one contiguous 512-byte segment.  The kernel is a pure HBM-write stream:
// per instance it reads n^2 + n m + r n + n + f r values and writes f r n + f r v m + (v m)^2 + v m.
//
// Math (reference: src/control/controller.rs; G_k = C A^k B are the Markov blocks):
//   Phi block i      = C A^(i+1)                                          (:32-44)
//   Gamma block i,k  = G_(i-k) for k <= i, k < v-1; the last block column (k = v-1) holds
//                      S_(i-v+1) = sum_{s <= i-v+1} G_s for rows i >= v-1      (:46-90)
//   H                = Gamma^T W4 Gamma + W3, W4 = blockdiag(Q)                (:92-93), evaluated per block
//                      diagonal as a running sum: H[a][a+d] = sum_{t <= f-1-a-d} G_(t+d)^T Q G_t
//   fvec             = Gamma^T W4 (Phi x0 - ref)                               (:96, :125-127)
#include "mpcx_internal.h"

namespace mpcx {
namespace {

constexpr int kLaneBlock = 64;  // one wavefront per workgroup: LDS is carved per lane, no barriers needed

struct LaneOut {
    double *Phi, *Gamma, *H, *fvec;
};

// UNI: every lane of a wavefront lies in the same interleave tile (W is a multiple of 64, or W >= batch), so the
// tile offset is wave-uniform and each access is "scalar base + 32-bit lane offset".
template <int N, int M, int R, bool UNI>
__global__ __launch_bounds__(kLaneBlock) void qp_lane_kernel(const double *__restrict__ gA, const double *__restrict__ gB,
                                                             const double *__restrict__ gC,
                                                             const double *__restrict__ gW3,
                                                             const double *__restrict__ gW4,
                                                             const double *__restrict__ gx0,
                                                             const double *__restrict__ gref, LaneOut p, long long batch,
                                                             long long W, int f, int v) {
    extern __shared__ double lds[];  // [slot][kLaneBlock]; lane t owns lds[slot * kLaneBlock + t]
    const int tid = threadIdx.x;
    const long long b = (long long)blockIdx.x * kLaneBlock + tid;
    if (b >= batch) return;
    long long tile = b / W;
    const long long lane_ll = b - tile * W;
    if (UNI) tile = (long long)__builtin_amdgcn_readfirstlane((int)tile);
    const unsigned lane = (unsigned)lane_ll;
    const int FR = f * R, VM = v * M;

    // element e of this lane's instance in an array of E elements per instance
#define IN(ptr, E, e) (UNI ? ((ptr) + (tile * (long long)(E) + (long long)(e)) * W)[lane] \
                           : (ptr)[tile * (long long)(E) * W + (long long)(e) * W + lane_ll])
#define LG(k, a, c) lds[(((k) * R + (a)) * M + (c)) * kLaneBlock + tid]
#define LZ(i, a) lds[((f * R * M) + (i) * R + (a)) * kLaneBlock + tid]

    double A[N][N], Bm[N][M], Cm[R][N];
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) A[i][j] = IN(gA, N * N, j * N + i);
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) Bm[i][j] = IN(gB, N * M, j * N + i);
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < R; ++i) Cm[i][j] = IN(gC, R * N, j * R + i);

    const bool need_g = p.Gamma || p.H || p.fvec;
    const bool need_q = p.H || p.fvec;
    double Q[R][R];
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
        for (int i = 0; i < R; ++i) Q[i][j] = need_q ? gW4[j * R + i] : 0.0;

    // Every global load is issued before the first global store: loads and stores retire through one in-order
    // counter, so a load placed after stores would wait for all of them.
    double x0[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x0[i] = p.fvec ? IN(gx0, N, i) : 0.0;
    if (p.fvec) {
        for (int i = 0; i < FR; ++i) lds[(f * R * M + i) * kLaneBlock + tid] = IN(gref, FR, i);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no load is left in flight once the stores start

    // ---- Markov blocks G_k = C (A^k B), k = 0..f-1, by the recurrence w <- A w ----
    if (need_g) {
        double w[N][M];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int c = 0; c < M; ++c) w[i][c] = Bm[i][c];
        for (int k = 0; k < f; ++k) {
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int c = 0; c < M; ++c) {
                    double g = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) g += Cm[a][j] * w[j][c];
                    LG(k, a, c) = g;
                }
            double wn[N][M];
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int c = 0; c < M; ++c) {
                    double s = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) s += A[i][j] * w[j][c];
                    wn[i][c] = s;
                }
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int c = 0; c < M; ++c) w[i][c] = wn[i][c];
        }
    }

    // ---- Phi blocks P_i = C A^(i+1) by P <- P A, and z_i = Q (P_i x0 - ref_i) ----
    if (p.Phi || p.fvec) {
        double P[R][N];
#pragma unroll
        for (int a = 0; a < R; ++a)
#pragma unroll
            for (int j = 0; j < N; ++j) P[a][j] = Cm[a][j];
        for (int i = 0; i < f; ++i) {
            double Pn[R][N];
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < N; ++k) s += P[a][k] * A[k][j];
                    Pn[a][j] = s;
                }
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int j = 0; j < N; ++j) P[a][j] = Pn[a][j];
            if (p.Phi) {
#pragma unroll
                for (int j = 0; j < N; ++j)
#pragma unroll
                    for (int a = 0; a < R; ++a) IN(p.Phi, FR * N, j * FR + i * R + a) = P[a][j];
            }
            if (p.fvec) {
                double s[R];
#pragma unroll
                for (int a = 0; a < R; ++a) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) acc += P[a][j] * x0[j];
                    s[a] = acc - LZ(i, a);  // the slot holds ref_i until z_i replaces it
                }
#pragma unroll
                for (int a = 0; a < R; ++a) {
                    double z = 0.0;
#pragma unroll
                    for (int q = 0; q < R; ++q) z += Q[a][q] * s[q];
                    LZ(i, a) = z;
                }
            }
        }
    }

    // ---- Gamma: column block k is the Markov sequence shifted down by k blocks ----
    if (p.Gamma) {
        for (int k = 0; k < v; ++k) {
            const bool last = (k == v - 1);
#pragma unroll
            for (int c = 0; c < M; ++c) {
                double run[R];  // running sum S for the held-input column
#pragma unroll
                for (int a = 0; a < R; ++a) run[a] = 0.0;
                const long long col = (long long)(k * M + c) * FR;
                for (int i = 0; i < f; ++i) {
#pragma unroll
                    for (int a = 0; a < R; ++a) {
                        double val = 0.0;
                        if (i >= k) {
                            const double g = LG(i - k, a, c);
                            run[a] += g;
                            val = last ? run[a] : g;
                        }
                        IN(p.Gamma, (long long)FR * VM, col + i * R + a) = val;
                    }
                }
            }
        }
    }

    // ---- H = Gamma^T blockdiag(Q) Gamma + W3 ----
    if (p.H) {
        const int nl = f - v + 1;  // rows i >= v-1: terms of every sum that involves the last block column
        // interior block diagonals: running sums over t of G_(t+e)^T Q G_t (upper, d = +e) and
        // G_t^T Q G_(t+e) (lower, d = -e); the block at (p, p+e) / (p+e, p) completes when t = f-1-e-p.
        for (int e = 0; e <= v - 2; ++e) {
            double up[M][M], lo[M][M];
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int j = 0; j < M; ++j) up[i][j] = lo[i][j] = 0.0;
            for (int t = 0; t <= f - 1 - e; ++t) {
                double g0[R][M], g1[R][M];
#pragma unroll
                for (int a = 0; a < R; ++a)
#pragma unroll
                    for (int c = 0; c < M; ++c) {
                        g0[a][c] = LG(t, a, c);
                        g1[a][c] = LG(t + e, a, c);
                    }
                double q0[R][M], q1[R][M];  // Q g0, Q g1
#pragma unroll
                for (int a = 0; a < R; ++a)
#pragma unroll
                    for (int c = 0; c < M; ++c) {
                        double s0 = 0.0, s1 = 0.0;
#pragma unroll
                        for (int q = 0; q < R; ++q) {
                            s0 += Q[a][q] * g0[q][c];
                            s1 += Q[a][q] * g1[q][c];
                        }
                        q0[a][c] = s0;
                        q1[a][c] = s1;
                    }
#pragma unroll
                for (int i = 0; i < M; ++i)
#pragma unroll
                    for (int j = 0; j < M; ++j) {
#pragma unroll
                        for (int a = 0; a < R; ++a) {
                            up[i][j] += g1[a][i] * q0[a][j];
                            lo[i][j] += g0[a][i] * q1[a][j];
                        }
                    }
                const int pb = f - 1 - e - t;  // smaller block index of the pair completed by this term
                if (pb + e <= v - 2) {
#pragma unroll
                    for (int i = 0; i < M; ++i)
#pragma unroll
                        for (int j = 0; j < M; ++j) {
                            const int r_up = pb * M + i, c_up = (pb + e) * M + j;
                            IN(p.H, (long long)VM * VM, (long long)c_up * VM + r_up) = up[i][j] + gW3[c_up * VM + r_up];
                            if (e != 0) {
                                const int r_lo = (pb + e) * M + i, c_lo = pb * M + j;
                                IN(p.H, (long long)VM * VM, (long long)c_lo * VM + r_lo) =
                                    lo[i][j] + gW3[c_lo * VM + r_lo];
                            }
                        }
                }
            }
        }
        // last block row and column: sums over the nl rows i >= v-1 against S_j = G_0 + ... + G_j
        for (int e = 0; e <= v - 1; ++e) {  // other block index is v-1-e
            double up[M][M], lo[M][M];  // up: H[v-1-e][v-1], lo: H[v-1][v-1-e]
            double S[R][M], S2[R][M];   // S_j, and S_j again as left factor when e == 0
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int j = 0; j < M; ++j) up[i][j] = lo[i][j] = 0.0;
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int c = 0; c < M; ++c) S[a][c] = 0.0;
            for (int j2 = 0; j2 < nl; ++j2) {
                double g1[R][M];
#pragma unroll
                for (int a = 0; a < R; ++a)
#pragma unroll
                    for (int c = 0; c < M; ++c) {
                        S[a][c] += LG(j2, a, c);
                        g1[a][c] = LG(j2 + e, a, c);
                        S2[a][c] = S[a][c];
                    }
                if (e == 0) {
#pragma unroll
                    for (int a = 0; a < R; ++a)
#pragma unroll
                        for (int c = 0; c < M; ++c) g1[a][c] = S2[a][c];
                }
                double qs[R][M], q1[R][M];
#pragma unroll
                for (int a = 0; a < R; ++a)
#pragma unroll
                    for (int c = 0; c < M; ++c) {
                        double s0 = 0.0, s1 = 0.0;
#pragma unroll
                        for (int q = 0; q < R; ++q) {
                            s0 += Q[a][q] * S[q][c];
                            s1 += Q[a][q] * g1[q][c];
                        }
                        qs[a][c] = s0;
                        q1[a][c] = s1;
                    }
#pragma unroll
                for (int i = 0; i < M; ++i)
#pragma unroll
                    for (int j = 0; j < M; ++j)
#pragma unroll
                        for (int a = 0; a < R; ++a) {
                            up[i][j] += g1[a][i] * qs[a][j];
                            lo[i][j] += S[a][i] * q1[a][j];
                        }
            }
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const int r_up = (v - 1 - e) * M + i, c_up = (v - 1) * M + j;
                    IN(p.H, (long long)VM * VM, (long long)c_up * VM + r_up) = up[i][j] + gW3[c_up * VM + r_up];
                    if (e != 0) {
                        const int r_lo = (v - 1) * M + i, c_lo = (v - 1 - e) * M + j;
                        IN(p.H, (long long)VM * VM, (long long)c_lo * VM + r_lo) = lo[i][j] + gW3[c_lo * VM + r_lo];
                    }
                }
        }
    }

    // ---- fvec = Gamma^T z ----
    if (p.fvec) {
        for (int k = 0; k < v; ++k) {
            const bool last = (k == v - 1);
            double acc[M], run[R][M];
#pragma unroll
            for (int c = 0; c < M; ++c) acc[c] = 0.0;
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int c = 0; c < M; ++c) run[a][c] = 0.0;
            for (int i = k; i < f; ++i) {
#pragma unroll
                for (int a = 0; a < R; ++a) {
                    const double z = LZ(i, a);
#pragma unroll
                    for (int c = 0; c < M; ++c) {
                        const double g = LG(i - k, a, c);
                        run[a][c] += g;
                        acc[c] += (last ? run[a][c] : g) * z;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < M; ++c) IN(p.fvec, VM, k * M + c) = acc[c];
        }
    }
#undef IN
#undef LG
#undef LZ
}

template <int N, int M, int R, bool UNI>
int launch_k(const QpArgs &a, size_t lds_bytes, unsigned blocks, hipStream_t s) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)qp_lane_kernel<N, M, R, UNI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    LaneOut o{a.Phi, a.Gamma, a.H, a.fvec};
    hipLaunchKernelGGL((qp_lane_kernel<N, M, R, UNI>), dim3(blocks), dim3(kLaneBlock), lds_bytes, s, a.A, a.B, a.C, a.W3,
                       a.W4, a.x0, a.ref, o, a.batch, a.W, a.f, a.v);
    return (int)hipGetLastError();
}

template <int N, int M, int R>
int launch_one(const QpArgs &a, hipStream_t s) {
    const size_t slots = (size_t)a.f * R * M + (size_t)a.f * R;
    const size_t lds_bytes = slots * kLaneBlock * sizeof(double);
    if (lds_bytes > 160 * 1024) return MPCX_ERR_UNSUPPORTED;
    const long long blocks = (a.batch + kLaneBlock - 1) / kLaneBlock;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    // wave-uniform tile and a lane offset whose byte offset fits 32 bits
    const bool uni = ((a.W % kLaneBlock) == 0 || a.W >= a.batch) && a.W <= (1LL << 28);
    return uni ? launch_k<N, M, R, true>(a, lds_bytes, (unsigned)blocks, s)
               : launch_k<N, M, R, false>(a, lds_bytes, (unsigned)blocks, s);
}

}  // namespace

bool qp_lane_supports(int n, int m, int r) {
    if (m == 1 && r == 1) return n >= 1 && n <= 6;
    if (m == 2 && r == 2) return n == 2 || n == 4 || n == 6;
    if (m == 2 && r == 1) return n == 3 || n == 4;
    if (m == 1 && r == 2) return n == 3 || n == 4;
    return false;
}

int launch_qp_lane(const QpArgs &a, hipStream_t s) {
    if (a.weights_mode != MPCX_WEIGHTS_BLOCK && (a.H || a.fvec)) return MPCX_ERR_UNSUPPORTED;
    if (!qp_lane_supports(a.n, a.m, a.r)) return MPCX_ERR_UNSUPPORTED;
#define CASE(N_, M_, R_) \
    if (a.n == N_ && a.m == M_ && a.r == R_) return launch_one<N_, M_, R_>(a, s);
    CASE(1, 1, 1) CASE(2, 1, 1) CASE(3, 1, 1) CASE(4, 1, 1) CASE(5, 1, 1) CASE(6, 1, 1)
    CASE(2, 2, 2) CASE(4, 2, 2) CASE(6, 2, 2)
    CASE(3, 2, 1) CASE(4, 2, 1) CASE(3, 1, 2) CASE(4, 1, 2)
#undef CASE
    return MPCX_ERR_UNSUPPORTED;
}

}  // namespace mpcx
