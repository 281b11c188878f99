// Condensed-QP build for plants too large for one lane: ONE WORKGROUP PER INSTANCE, instance-major layout.
//
// A, B, C of the instance are staged in LDS.  The Markov blocks G_k = C A^k B and the Phi blocks C A^(k+1) come
// from two O(f) recurrences that advance together, one barrier per step; every thread owns one output element of
// a step and keeps its row / column of A or C in registers for the whole loop.  Gamma is then a set of shifted
// copies of the Markov sequence (kept in LDS as one vector per input column), written column by column with
// lanes along the rows, so every store instruction covers a contiguous run of a column.  H uses the
// block-Toeplitz structure of Gamma: with W4 = blockdiag(Q),
//     H[a][b] = H[a+1][b+1] + G_(f-1-a)^T Q G_(f-1-b)        for block indices a, b <= v-3,
// so the workgroup sweeps the column blocks from right to left, one barrier per block.  A thread owns one column
// (a, j) of one block row: it holds G_(f-1-a) in registers for the whole sweep, reads Q G_(f-1-b)[:, j] from LDS
// (the same address across most of the wave) and produces m consecutive rows of column b m + j of H, stored with
// lanes along the rows.  The blocks that touch the last two block rows / columns are summed directly (they
// involve the held-input column S_j = G_0 + ... + G_j when f > v).
//
// Loads and stores of one wave retire through a single in-order counter, so a wave that has stores in flight
// waits for all of them before a later load returns.  The four compute waves therefore never load from global
// memory once their first store is out: everything they need is in LDS, and the shared weight W3 (dense,
// (v m)^2, far larger than LDS) is streamed column block by column block into an LDS ring by three more waves
// that never store and take the blocks in turn, so that every fetch has two sweep steps to arrive.
//
// Reference math: src/control/controller.rs:32-44 (Phi), :46-90 (Gamma), :92-93 (H), :96 and :125-127 (fvec).
#include "mpcx_internal.h"

namespace mpcx {
namespace {

constexpr int kCompute = 256;            // threads that compute and store
constexpr int kLoaders = 3;               // loader waves: each fetches every kLoaders-th column block of W3
constexpr int kWgBlock = kCompute + 64 * kLoaders;
constexpr int kCoef = 16;                // recurrence coefficients a thread keeps in registers (n <= kCoef)

struct WgLds {
    // offsets in doubles into the dynamic LDS buffer
    int G, S, QG, QS, A, B, C, x0, Q, Wst, Pst, Sr, Z, Hp, W3r, Tab, total;
};

// G, S, QG, QS live for the whole kernel.  The inputs, the recurrence states and z are dead once fvec is out, and
// the H sweep's buffers (two column blocks of H, two of W3) reuse their space.
__host__ __device__ inline WgLds wg_lds_layout(int n, int m, int r, int f, int v) {
    WgLds L;
    int o = 0;
    const int fr = f * r, nl = f - v + 1;
    L.G = o; o += m * fr;        // [c][k*r + a] = G_k[a][c]
    L.S = o; o += m * nl * r;    // [c][j*r + a] = S_j[a][c]
    L.QG = o; o += m * fr;       // Q G_k, same indexing as G
    L.QS = o; o += m * nl * r;   // Q S_j
    const int u = o;
    L.A = o; o += n * n;
    L.B = o; o += n * m;
    L.C = o; o += r * n;
    L.x0 = o; o += 2 * n;      // x0, then A x0
    L.Q = o; o += r * r;
    L.Wst = o; o += 2 * n * m;   // Markov state, ping-pong
    L.Pst = o; o += 2 * r * n;   // Phi state, ping-pong
    L.Sr = o; o += fr;           // ref, then Phi x0 - ref
    L.Z = o; o += fr;            // Q (Phi x0 - ref)
    const int end1 = o;
    o = u;
    L.Hp = o; o += 2 * v * m * m;   // column block of H (without W3), ping-pong: [slot][i*m + j][a]
    L.W3r = o; o += 2 * v * m * m;  // ring of two column blocks of W3: [slot][j][row]
    L.Tab = o; o += (f > v) ? 3 * v * m * m : 0;  // held-input case only: last column, last row, Toeplitz seed row
    L.total = o > end1 ? o : end1;
    return L;
}

// One element of block (aa, bb), column j, row i, summed directly over the rows i2 >= max(aa, bb):
// sum_i2 X_aa(i2)[:, i] . Q X_bb(i2)[:, j] with X_k(i2) = G_(i2-k).  With use_s, a last block (index v-1) is the
// held-input column, X_(v-1)(i2) = S_(i2-v+1) (the cumulated Markov blocks); without it every column is a plain shift.
__device__ inline double h_direct(const double *sG, const double *sS, const double *sQG, const double *sQS, int aa,
                                  int bb, int i, int j, int f, int v, int r, int FR, int nl, bool use_s) {
    double acc = 0.0;
    const bool la = use_s && (aa == v - 1), lb = use_s && (bb == v - 1);
    const int i0 = aa > bb ? aa : bb;
    for (int i2 = i0; i2 < f; ++i2) {
        const double *ga = la ? (sS + i * nl * r + (i2 - (v - 1)) * r) : (sG + i * FR + (i2 - aa) * r);
        const double *qb = lb ? (sQS + j * nl * r + (i2 - (v - 1)) * r) : (sQG + j * FR + (i2 - bb) * r);
        for (int q = 0; q < r; ++q) acc += ga[q] * qb[q];
    }
    return acc;
}

// The loader wave's W3 pipeline.  A column block of W3 (cnt doubles, contiguous) is fetched into registers during
// one sweep step and written to its LDS ring slot during the next, so the fetch latency hides behind a whole step.
// Up to kPreDoubles doubles per lane are held; larger blocks are fetched in place (latency exposed).
constexpr int kPre = 13;  // register slots per lane, one or two doubles each
struct W3Pipe {
    double2 r[kPre];
};
// Loads are unconditional (an index past the block is clamped to its last element and the value dropped at commit):
// a load guarded by a per-lane branch makes the compiler wait for it before issuing the next one, which serialises
// the fetch and keeps the wave from reaching the step's barrier.
__device__ __forceinline__ void w3_fetch(W3Pipe &pp, const double *src, int cnt, int lane, bool wide) {
    if (wide) {
        const double2 *s2 = (const double2 *)src;
        const int n2 = cnt >> 1;
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int e = lane + u * 64;
            if (u * 64 < n2) pp.r[u] = s2[e < n2 ? e : n2 - 1];  // wave-uniform guard
        }
    } else {
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int e = lane + u * 64;
            if (u * 64 < cnt) pp.r[u].x = src[e < cnt ? e : cnt - 1];
        }
    }
}
__device__ __forceinline__ void w3_commit(const W3Pipe &pp, double *dst, int cnt, int lane, bool wide) {
    if (wide) {
        double2 *d2 = (double2 *)dst;
        const int n2 = cnt >> 1;
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int e = lane + u * 64;
            if (e < n2) d2[e] = pp.r[u];
        }
    } else {
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int e = lane + u * 64;
            if (e < cnt) dst[e] = pp.r[u].x;
        }
    }
}

// HM, HR, ITEMS > 0: the H sweep is specialised for m = HM, r = HR with up to ITEMS (block row, column) pairs per
// thread held in registers (needs v m <= ITEMS * kCompute).  HM = 0: generic sweep, operands read from LDS.
template <int HM, int HR, int ITEMS>
__global__ __launch_bounds__(kWgBlock, 4) void qp_wg_kernel(QpArgs p) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const bool loader = tid >= kCompute;  // wave-uniform: the waves after the four compute waves
    const int ltid = (tid - kCompute) & 63, lw = (tid - kCompute) >> 6;  // lane and index of the loader wave
    const long long b = blockIdx.x;
    const int n = p.n, m = HM ? HM : p.m, r = HR ? HR : p.r, f = p.f, v = p.v;
    const int FR = f * r, VM = v * m, nl = f - v + 1;
    const WgLds L = wg_lds_layout(n, m, r, f, v);
    double *sA = lds + L.A, *sB = lds + L.B, *sC = lds + L.C, *sx0 = lds + L.x0, *sQ = lds + L.Q;
    double *sW = lds + L.Wst, *sP = lds + L.Pst, *sG = lds + L.G, *sS = lds + L.S, *sQG = lds + L.QG;
    double *sQS = lds + L.QS, *sSr = lds + L.Sr, *sZ = lds + L.Z, *sHp = lds + L.Hp, *sW3 = lds + L.W3r;
    double *sTab = lds + L.Tab, *sAx = sx0 + n;

    const double *gA = p.A + b * (long long)(n * n);
    const double *gB = p.B + b * (long long)(n * m);
    const double *gC = p.C + b * (long long)(r * n);
    const bool need_q = p.H || p.fvec;

    // ---- every per-instance input goes to LDS before the first store of the workgroup ----
    for (int e = tid; e < n * n; e += kWgBlock) sA[e] = gA[e];
    for (int e = tid; e < n * m; e += kWgBlock) { const double x = gB[e]; sB[e] = x; sW[e] = x; }
    for (int e = tid; e < r * n; e += kWgBlock) { const double x = gC[e]; sC[e] = x; sP[e] = x; }
    if (p.fvec) {
        for (int e = tid; e < n; e += kWgBlock) sx0[e] = p.x0[b * n + e];
        for (int e = tid; e < FR; e += kWgBlock) sSr[e] = p.ref[b * (long long)FR + e];
    }
    if (need_q) for (int e = tid; e < r * r; e += kWgBlock) sQ[e] = p.W4[e];
    lds_barrier();

    // ---- recurrences: step k produces G_k (from W_k), W_(k+1) = A W_k, P_(k+1) = P_k A (P_0 = C) ----
    // Task t of a step is one dot product of length n: coefficient vector (a row of C, a row of A or a column of A)
    // times an operand vector read from the current state with a fixed stride.
    // With fvec requested, r more tasks per step give s_k = Phi_k x0 - ref_k as P_k (A x0) - ref_k.
    const int nG = r * m, nW = n * m, nP = r * n, nS = p.fvec ? r : 0, nT = nG + nW + nP + nS;
    double *gPhi = p.Phi ? p.Phi + b * (long long)FR * n : nullptr;
    if (p.fvec) {
        for (int i = tid; i < n; i += kWgBlock) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += sA[j * n + i] * sx0[j];
            sAx[i] = s;
        }
        lds_barrier();
    }
    const bool one_task = (nT <= kCompute) && (n <= kCoef);
    if (one_task) {
        // decode this thread's task once; its coefficients stay in registers for all f steps
        int kind = -1, op_off = 0, op_stride = 1, out_off = 0;
        long long phi_off = 0;
        double coef[kCoef];
#pragma unroll
        for (int j = 0; j < kCoef; ++j) coef[j] = 0.0;
        if (!loader && tid < nT) {
            if (tid < nG) {
                const int a = tid % r, c = tid / r;
                kind = 0; op_off = c * n; op_stride = 1; out_off = c * FR + a;
#pragma unroll
                for (int j = 0; j < kCoef; ++j) if (j < n) coef[j] = sC[j * r + a];
            } else if (tid < nG + nW) {
                const int u = tid - nG, i = u % n, c = u / n;
                kind = 1; op_off = c * n; op_stride = 1; out_off = u;
#pragma unroll
                for (int j = 0; j < kCoef; ++j) if (j < n) coef[j] = sA[j * n + i];
            } else if (tid < nG + nW + nP) {
                const int u = tid - nG - nW, a = u % r, j2 = u / r;
                kind = 2; op_off = a; op_stride = r; out_off = u; phi_off = (long long)j2 * FR + a;
#pragma unroll
                for (int j = 0; j < kCoef; ++j) if (j < n) coef[j] = sA[j2 * n + j];
            } else {
                const int a = tid - nG - nW - nP;
                kind = 3; op_off = a; op_stride = r; out_off = a;
#pragma unroll
                for (int j = 0; j < kCoef; ++j) if (j < n) coef[j] = sAx[j];
            }
        }
        for (int k = 0; k < f; ++k) {
            const double *Wc = sW + (k & 1) * nW;
            double *Wn = sW + ((k + 1) & 1) * nW;
            const double *Pc = sP + (k & 1) * nP;
            double *Pn = sP + ((k + 1) & 1) * nP;
            if (kind >= 0) {
                const double *op = (kind >= 2 ? Pc : Wc) + op_off;
                double s = 0.0;
                // unconditional loads (index clamped, coefficient zero past n): all issue together, one wait
#pragma unroll
                for (int j = 0; j < kCoef; ++j) s += coef[j] * op[(j < n ? j : n - 1) * op_stride];
                if (kind == 0) {
                    sG[out_off + k * r] = s;
                } else if (kind == 1) {
                    Wn[out_off] = s;
                } else if (kind == 2) {
                    Pn[out_off] = s;
                    if (gPhi) gPhi[phi_off + k * r] = s;
                } else {
                    sSr[k * r + out_off] = s - sSr[k * r + out_off];
                }
            }
            lds_barrier();
        }
    } else {
        for (int k = 0; k < f; ++k) {
            const double *Wc = sW + (k & 1) * nW;
            double *Wn = sW + ((k + 1) & 1) * nW;
            const double *Pc = sP + (k & 1) * nP;
            double *Pn = sP + ((k + 1) & 1) * nP;
            if (!loader) {
                for (int t = tid; t < nT; t += kCompute) {
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
                    } else if (t < nG + nW + nP) {  // P_(k+1)[a][j] = sum_i P_k[a][i] A[i][j]   (= Phi block k)
                        const int u = t - nG - nW, a = u % r, j = u / r;
                        double s = 0.0;
                        for (int i = 0; i < n; ++i) s += Pc[i * r + a] * sA[j * n + i];
                        Pn[u] = s;
                        if (gPhi) gPhi[(long long)j * FR + k * r + a] = s;
                    } else {  // s_k[a] = sum_i P_k[a][i] (A x0)[i] - ref_k[a]
                        const int a = t - nG - nW - nP;
                        double s = 0.0;
                        for (int i = 0; i < n; ++i) s += Pc[i * r + a] * sAx[i];
                        sSr[k * r + a] = s - sSr[k * r + a];
                    }
                }
            }
            lds_barrier();
        }
    }
    lds_barrier();

    // ---- S_j = G_0 + ... + G_j (j < nl), Q G_k, Q S_j, Z = Q s ----
    for (int t = tid; t < nG; t += kWgBlock) {
        const int a = t % r, c = t / r;
        double run = 0.0;
        for (int j = 0; j < nl; ++j) {
            run += sG[c * FR + j * r + a];
            sS[c * nl * r + j * r + a] = run;
        }
    }
    lds_barrier();
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
    lds_barrier();

    // ---- Gamma: column (k, c) = [zeros(k r) ; Markov vector c] (or the cumulated vector for k = v-1).
    //      A pure store stream from LDS (G and S stay valid to the end of the kernel), so it can run before or after
    //      the H sweep.  Workgroups that share a CU run the same phases for the same time and would stay in lockstep,
    //      all storing Gamma at once and all sweeping at once; every other resident workgroup therefore writes Gamma
    //      last, so that one workgroup's store stream overlaps the other's sweep.
    //      Only where Gamma is the bulk of the bytes (f r >= 2 v m): with H as large as Gamma the sweep is a store
    //      stream itself, and workgroups sweeping in step share their W3 blocks in L2. ----
    const bool gamma_last = p.Gamma && p.H && FR >= 2 * VM && (blockIdx.x & 1);
    auto write_gamma = [&]() {
        double *gG = p.Gamma + b * (long long)FR * VM;
        const int wave = tid >> 6, lane = tid & 63, nwaves = kCompute >> 6;
        for (int col = wave; col < VM; col += nwaves) {
            const int k = col / m, c = col % m;
            const int shift = k * r;
            const double *src = (k == v - 1) ? (sS + c * nl * r) : (sG + c * FR);
            double *dst = gG + (long long)col * FR;
            for (int row = lane; row < FR; row += 64) dst[row] = (row < shift) ? 0.0 : src[row - shift];
        }
    };
    if (p.Gamma && !loader && !gamma_last) write_gamma();

    // ---- fvec[col] = sum_rows Gamma[row][col] Z[row] ----
    if (p.fvec && !loader) {
        double *gf = p.fvec + b * (long long)VM;
        for (int col = tid; col < VM; col += kCompute) {
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

    // ---- H, swept by column block from v-1 down to 0.  The previous column block sits in one half of sHp, the
    //      new one goes to the other half; the loader wave streams the next block of W3 into its ring meanwhile.
    //      sHp is indexed [i*m + j][a]: lanes along the block rows, so reads and writes are conflict-free.
    //      What is swept is the Gram matrix Ht of the plain-shift columns, for which the recurrence holds for every
    //      block with Ht[v][.] = 0 when f = v.  When f > v the row Ht[v-1][.] is seeded from a table, and the last
    //      block row and column of H (which involve the held-input column) are taken from two more tables. ----
    if (p.H) {
        lds_barrier();  // z and the inputs are dead: their LDS becomes the sweep's buffers
        // W3 pipeline of the loader wave: block v-1 goes straight to its slot, block v-2 is fetched ahead
        const int wcnt = VM * m;  // doubles in one column block of W3
        const bool wide = ((wcnt & 1) == 0) && (((uintptr_t)p.W3 & 15) == 0);
        const bool piped = wide ? (wcnt <= 2 * 64 * kPre) : (wcnt <= 64 * kPre);
        if (loader) {
            const double *src = p.W3 + (long long)(v - 1) * wcnt;
            double *dst = sW3 + ((v - 1) & 1) * wcnt;
            for (int e = tid - kCompute; e < wcnt; e += 64 * kLoaders) dst[e] = src[e];
        }
        const int bsz = m * m, tsz = v * bsz;
        const bool held = nl > 1;
        if (held) {
            for (int t = tid; t < 3 * tsz; t += kWgBlock) {
                const int which = t / tsz, rem = t - which * tsz, blk = rem / bsz, ij = rem - blk * bsz;
                const int i = ij / m, j = ij - i * m;
                sTab[t] = which == 0   ? h_direct(sG, sS, sQG, sQS, blk, v - 1, i, j, f, v, r, FR, nl, true)
                          : which == 1 ? h_direct(sG, sS, sQG, sQS, v - 1, blk, i, j, f, v, r, FR, nl, true)
                                       : h_direct(sG, sS, sQG, sQS, v - 1, blk, i, j, f, v, r, FR, nl, false);
            }
        }
        const double *tabCol = sTab, *tabRow = sTab + tsz, *tabSeed = sTab + 2 * tsz;
        double *gH = p.H + b * (long long)VM * VM;

        if constexpr (HM > 0) {
            // item it of this thread: column j of block row aa, t = tid + it * kCompute = j * v + aa
            double ga[ITEMS][HR][HM];  // G_(f-1-aa)[q][i]
            int item_aa[ITEMS], item_j[ITEMS];
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int t = tid + it * kCompute;
                const bool live = !loader && t < VM;
                item_j[it] = live ? t / v : -1;
                item_aa[it] = live ? t - (t / v) * v : 0;
#pragma unroll
                for (int q = 0; q < HR; ++q)
#pragma unroll
                    for (int i = 0; i < HM; ++i)
                        ga[it][q][i] = live ? sG[i * FR + (f - 1 - item_aa[it]) * HR + q] : 0.0;
            }
            lds_barrier();
            // Two loops with the same number of barriers: the loader wave's and the compute waves'.  Keeping them
            // apart keeps the loader's fetch registers and the compute waves' G blocks out of each other's live ranges.
            if (loader) {
                // Block k belongs to loader wave k % kLoaders, which fetches it into registers kLoaders steps before
                // the step that needs it and commits it to the ring one step before: a fetch has kLoaders - 1 whole
                // steps to complete, and no wave waits on a load it has just issued.
                W3Pipe pipe;
                {
                    int k0 = v - 2;  // first block this wave commits: the largest k <= v-2 with k % kLoaders == lw
                    while (k0 >= 0 && (k0 % kLoaders) != lw) --k0;
                    if (piped && k0 >= 0) w3_fetch(pipe, p.W3 + (long long)k0 * wcnt, wcnt, ltid, wide);
                }
                for (int bb = v - 1; bb >= 0; --bb) {
                    if (bb > 0 && ((bb - 1) % kLoaders) == lw) {  // block bb-1 into the slot the next step reads
                        double *dst = sW3 + ((bb - 1) & 1) * wcnt;
                        if (piped) {
                            w3_commit(pipe, dst, wcnt, ltid, wide);
                            if (bb - 1 - kLoaders >= 0)
                                w3_fetch(pipe, p.W3 + (long long)(bb - 1 - kLoaders) * wcnt, wcnt, ltid, wide);
                        } else {
                            const double *src = p.W3 + (long long)(bb - 1) * wcnt;
                            for (int e = ltid; e < wcnt; e += 64) dst[e] = src[e];
                        }
                    }
                    lds_barrier();
                }
            } else {
                for (int bb = v - 1; bb >= 0; --bb) {
                    double *cur = sHp + (bb & 1) * VM * m;
                    const double *prev = sHp + ((bb + 1) & 1) * VM * m;
                    const double *w3 = sW3 + (bb & 1) * VM * m;
                    const bool first = (bb == v - 1);
#pragma unroll
                    for (int it = 0; it < ITEMS; ++it) {
                        const int j = item_j[it], aa = item_aa[it];
                        if (j < 0) continue;
                        const bool lastrow = (aa == v - 1);
                        double acc[HM];
                        if (first && held) {
#pragma unroll
                            for (int i = 0; i < HM; ++i)
                                acc[i] = h_direct(sG, sS, sQG, sQS, aa, bb, i, j, f, v, HR, FR, nl, false);
                        } else {
                            const double *qb = sQG + j * FR + (f - 1 - bb) * HR;
                            const int an = lastrow ? aa : aa + 1;
#pragma unroll
                            for (int i = 0; i < HM; ++i) {
                                const double pv = prev[(i * HM + j) * v + an];
                                acc[i] = (first || lastrow) ? 0.0 : pv;
                            }
#pragma unroll
                            for (int q = 0; q < HR; ++q) {
                                const double x = qb[q];
#pragma unroll
                                for (int i = 0; i < HM; ++i) acc[i] += ga[it][q][i] * x;
                            }
                            if (held && lastrow) {
#pragma unroll
                                for (int i = 0; i < HM; ++i) acc[i] = tabSeed[(bb * HM + i) * HM + j];
                            }
                        }
                        double *dst = gH + (long long)(bb * HM + j) * VM + aa * HM;
                        const double *wsrc = w3 + j * VM + aa * HM;
#pragma unroll
                        for (int i = 0; i < HM; ++i) {
                            cur[(i * HM + j) * v + aa] = acc[i];
                            if (held && first) acc[i] = tabCol[(aa * HM + i) * HM + j];
                            else if (held && lastrow) acc[i] = tabRow[(bb * HM + i) * HM + j];
                            acc[i] += wsrc[i];
                        }
                        if constexpr (HM % 2 == 0) {
#pragma unroll
                            for (int i = 0; i < HM; i += 2) *(double2 *)(dst + i) = make_double2(acc[i], acc[i + 1]);
                        } else {
#pragma unroll
                            for (int i = 0; i < HM; ++i) dst[i] = acc[i];
                        }
                    }
                    lds_barrier();
                }
            }
        } else {
            lds_barrier();
            // Two loops with the same number of barriers: the loader wave's and the compute waves'.  Keeping them
            // apart keeps the loader's fetch registers and the compute waves' G blocks out of each other's live ranges.
            if (loader) {
                // Block k belongs to loader wave k % kLoaders, which fetches it into registers kLoaders steps before
                // the step that needs it and commits it to the ring one step before: a fetch has kLoaders - 1 whole
                // steps to complete, and no wave waits on a load it has just issued.
                W3Pipe pipe;
                {
                    int k0 = v - 2;  // first block this wave commits: the largest k <= v-2 with k % kLoaders == lw
                    while (k0 >= 0 && (k0 % kLoaders) != lw) --k0;
                    if (piped && k0 >= 0) w3_fetch(pipe, p.W3 + (long long)k0 * wcnt, wcnt, ltid, wide);
                }
                for (int bb = v - 1; bb >= 0; --bb) {
                    if (bb > 0 && ((bb - 1) % kLoaders) == lw) {  // block bb-1 into the slot the next step reads
                        double *dst = sW3 + ((bb - 1) & 1) * wcnt;
                        if (piped) {
                            w3_commit(pipe, dst, wcnt, ltid, wide);
                            if (bb - 1 - kLoaders >= 0)
                                w3_fetch(pipe, p.W3 + (long long)(bb - 1 - kLoaders) * wcnt, wcnt, ltid, wide);
                        } else {
                            const double *src = p.W3 + (long long)(bb - 1) * wcnt;
                            for (int e = ltid; e < wcnt; e += 64) dst[e] = src[e];
                        }
                    }
                    lds_barrier();
                }
            } else {
                for (int bb = v - 1; bb >= 0; --bb) {
                    double *cur = sHp + (bb & 1) * VM * m;
                    const double *prev = sHp + ((bb + 1) & 1) * VM * m;
                    const double *w3 = sW3 + (bb & 1) * VM * m;
                    const bool first = (bb == v - 1);
                    for (int t = tid; t < VM; t += kCompute) {  // t = j * v + aa, as in the specialised sweep
                        const int j = t / v, aa = t - j * v;
                        const bool lastrow = (aa == v - 1);
                        const int an = lastrow ? aa : aa + 1;
                        const double *qb = sQG + j * FR + (f - 1 - bb) * r;
                        for (int i = 0; i < m; ++i) {
                            double acc;
                            if (first && held) {
                                acc = h_direct(sG, sS, sQG, sQS, aa, bb, i, j, f, v, r, FR, nl, false);
                            } else {
                                const double pv = prev[(i * m + j) * v + an];
                                acc = (first || lastrow) ? 0.0 : pv;
                                const double *ga = sG + i * FR + (f - 1 - aa) * r;
                                for (int q = 0; q < r; ++q) acc += ga[q] * qb[q];
                                if (held && lastrow) acc = tabSeed[(bb * m + i) * m + j];
                            }
                            cur[(i * m + j) * v + aa] = acc;
                            if (held && first) acc = tabCol[(aa * m + i) * m + j];
                            else if (held && lastrow) acc = tabRow[(bb * m + i) * m + j];
                            gH[(long long)(bb * m + j) * VM + aa * m + i] = acc + w3[j * VM + aa * m + i];
                        }
                    }
                    lds_barrier();
                }
            }
        }
    }
    if (gamma_last && !loader) write_gamma();
}

template <int HM, int HR, int ITEMS>
int launch_wg_k(const QpArgs &a, size_t lds_bytes, hipStream_t s) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)qp_wg_kernel<HM, HR, ITEMS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((qp_wg_kernel<HM, HR, ITEMS>), dim3((unsigned)a.batch), dim3(kWgBlock), lds_bytes, s, a);
    return (int)hipGetLastError();
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
    const long long items = ((long long)a.v * a.m + kCompute - 1) / kCompute;
    // 16-byte stores of the specialised sweep: the leading dimension v m is even whenever m is; the base must be too
    const bool even_rows = (a.m % 2 != 0) || (((uintptr_t)a.H & 15) == 0);
#define FAST(M_, R_, I_) \
    if (a.m == M_ && a.r == R_ && items <= I_ && even_rows) return launch_wg_k<M_, R_, I_>(a, lds_bytes, s);
    FAST(4, 12, 1) FAST(2, 2, 4) FAST(1, 1, 4) FAST(4, 4, 2) FAST(3, 3, 2) FAST(2, 4, 2)
#undef FAST
    return launch_wg_k<0, 0, 1>(a, lds_bytes, s);
}

}  // namespace mpcx
