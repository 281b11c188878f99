// Condensed-QP build for plants too large for one lane: ONE WORKGROUP PER INSTANCE, instance-major layout.
//
// A, B, C of the instance are staged in LDS.  For plants of up to 16 states the Markov blocks G_k = C A^k B, the Phi
// blocks C A^(k+1) and s_k = Phi_k x0 - ref_k come from power tables built by doubling (about log2 f rounds of 16x16
// tile products on the matrix cores, one barrier each) and one tile product per 16 output rows (see WgTab); larger
// plants run the two O(f) recurrences step by step, one barrier per step.  Gamma is then a set of shifted
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
// (v m)^2, far larger than LDS) is streamed column block by column block into an LDS ring by four more waves
// that never store to it: each holds its share of up to eight blocks in flight in registers, so that a fetch has
// several sweep steps to arrive (a CU's vector-memory pipeline is in order: loads queue behind the store stream).
// The same four waves form fvec (a Hankel product on the matrix cores) and, with held inputs, the short sums that
// border the swept matrix, while the compute waves write Gamma (16 bytes per lane where the counts are even: a
// wavefront sustains 40 GB/s of 16-byte stores against 17 GB/s of 8-byte ones, tools/cu_write_rate.hip).
//
// Reference math: src/control/controller.rs:32-44 (Phi), :46-90 (Gamma), :92-93 (H), :96 and :125-127 (fvec).
#include <limits.h>

#include <type_traits>
#include <utility>

#include "mpcx_internal.h"

namespace mpcx {
namespace {

constexpr int kCompute = 256;            // threads that compute and store
#ifndef MPCX_WG_LOADERS
#define MPCX_WG_LOADERS 4
#endif
constexpr int kLoaders = MPCX_WG_LOADERS; // loader waves: each fetches every kLoaders-th column block of W3
constexpr int kWgBlock = kCompute + 64 * kLoaders;

struct WgLds {
    // offsets in doubles into the dynamic LDS buffer
    int G, S, QG, QS, Cnt, A, B, C, x0, Q, Wst, Pst, Sr, Z, Pow, Hp, W3r, Tab, Vcol, total;
};

// Power tables of the Markov / Phi construction (plants with n <= kTabN states).  With k = Lp hi + lo (Lp a power of
// two, Lp^2 >= f) every block of the three sequences is one short product of two table entries:
//     Phi block k = (C A^lo) (A^(Lp hi + 1)),   G_k = (C A^lo) (A^(Lp hi) B),   s_k = (C A^lo) (A^(Lp hi + 1) x0) - ref_k,
// so the serial chain of f steps becomes log2(Lp) + log2(nh) doubling rounds that build the tables (nh = ceil(f / Lp)
// values of hi), and all blocks then come from independent 16x16 products on the matrix cores.
struct WgTab {
    int Lp, nh, RP;          // lo range, hi range, rows of the (C A^lo) table padded to whole tiles of 16
    int PP, TC, X, Y, total; // offsets (doubles) from the tables' base: ping-pong powers, C A^lo, A^(Lp hi + 1), [A^(Lp hi) B | A^(Lp hi + 1) x0]
};
constexpr int kTabN = 16;
__host__ __device__ inline WgTab wg_tab_layout(int n, int m, int r, int f) {
    WgTab T;
    int Lp = 1;
    while (Lp * Lp < f) Lp *= 2;
    T.Lp = Lp;
    T.nh = (f + Lp - 1) / Lp;
    T.RP = (Lp * r + 15) & ~15;
    int o = 0;
    T.PP = o; o += 2 * n * n;
    T.TC = o; o += n * T.RP;
    T.X = o; o += T.nh * n * n;
    T.Y = o; o += T.nh * n * (m + 1);
    T.total = o;
    return T;
}

// G, S, QG, QS live for the whole kernel.  The inputs, the recurrence states and z are dead once fvec is out, and
// the H sweep's buffers (two column blocks of H, two of W3) reuse their space.
__host__ __device__ inline WgLds wg_lds_layout(int n, int m, int r, int f, int v) {
    WgLds L;
    int o = 0;
    const int fr = f * r, nl = f - v + 1;
    const bool tables = n <= kTabN;
    L.G = o; o += m * fr;        // [c][k*r + a] = G_k[a][c]
    L.S = o; o += m * nl * r;    // [c][j*r + a] = S_j[a][c]
    L.QG = o; o += m * fr;       // Q G_k, same indexing as G
    L.QS = o; o += m * nl * r;   // Q S_j
    L.Cnt = o; o += 2;           // work counter of the held-input sums (an int)
    const int u = o;
    L.A = o; o += n * n;
    L.B = o; o += n * m;
    L.C = o; o += r * n;
    L.x0 = o; o += 2 * n;      // x0, then A x0
    L.Q = o; o += r * r;
    L.Wst = o; o += tables ? 0 : 2 * n * m;   // Markov state, ping-pong (plants beyond kTabN states)
    L.Pst = o; o += tables ? 0 : 2 * r * n;   // Phi state, ping-pong
    L.Sr = o; o += fr;           // ref, then Phi x0 - ref
    L.Z = o; o += fr;            // Q (Phi x0 - ref)
    // The power tables are dead before S, Q G and Q S are formed: they borrow that space when it is large enough.
    const int tab_sz = tables ? wg_tab_layout(n, m, r, f).total : 0;
    if (tab_sz <= u - L.S) { L.Pow = L.S; } else { L.Pow = o; o += tab_sz; }
    const int end1 = o;
    o = u;
    L.Hp = o; o += 2 * (v + 1) * m * m;  // column block of the swept Gram matrix, ping-pong: [slot][i*m + j][a], a = 0..v
    L.W3r = o; o += 2 * v * m * m;       // ring of two column blocks of W3: [slot][j][row]
    // Held inputs (f > v): the virtual block row, [i*m + j][b'].  Where the border is summed on the matrix cores
    // (2 m <= 16) the sums run beside fvec and the Gamma stores, i.e. while the inputs and z are still live: the
    // virtual row and the virtual column (Vcol, the "previous column" of the first sweep step) then lie past both uses
    // of the shared space.
    L.Vcol = -1;
    if (f > v && 2 * m <= 16) {
        o = o > end1 ? o : end1;
        L.Tab = o; o += (v + 1) * m * m;
        L.Vcol = o; o += (v + 1) * m * m;
    } else {
        L.Tab = o; o += (f > v) ? (v + 1) * m * m : 0;
    }
    L.total = o > end1 ? o : end1;
    return L;
}

// One element of block (aa, bb), column j, row i, summed directly over the rows i2 >= max(aa, bb):
// sum_i2 X_aa(i2)[:, i] . Q X_bb(i2)[:, j] with X_k(i2) = G_(i2-k).  With use_s, a last block (index v-1) is the
// held-input column, X_(v-1)(i2) = S_(i2-v+1) (the cumulated Markov blocks); without it every column is a plain shift.
__device__ inline double h_direct(const double *sG, const double *sS, const double *sQG, const double *sQS, int aa,
                                  int bb, int i, int j, int f, int v, int r, int FR, int nl, bool use_s) {
    const bool la = use_s && (aa == v - 1), lb = use_s && (bb == v - 1);
    const int i0 = aa > bb ? aa : bb;
    // both operands are contiguous over (i2, q): one dot product of length (f - i0) r
    const double *ga = la ? (sS + i * nl * r + (i0 - (v - 1)) * r) : (sG + i * FR + (i0 - aa) * r);
    const double *qb = lb ? (sQS + j * nl * r + (i0 - (v - 1)) * r) : (sQG + j * FR + (i0 - bb) * r);
    const int len = (f - i0) * r;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    int t = 0;
    for (; t + 7 < len; t += 8) {  // eight independent LDS read pairs per round: one latency, not eight
        double x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = ga[t + u]; y[u] = qb[t + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u & 3] += x[u] * y[u];
    }
    for (; t < len; ++t) a[t & 3] += ga[t] * qb[t];
    return (a[0] + a[1]) + (a[2] + a[3]);
}

// Held inputs (f > v): what the sweep's recurrence cannot produce are families of short sums over the block rows
// i2 >= v-1 (L = f - v + 1 of them): the virtual block column and row that border the swept Gram matrix Ht,
//     Ht[a'][v] = sum_(i2 >= v) G_(i2-a')^T Q G_(i2-v),     Ht[v][b'] = sum_(i2 >= v) G_(i2-v)^T Q G_(i2-b')      (L - 1 terms)
// and the last block column and row of H itself, which involve the held-input column S (L terms).  With d = v - a'
// (resp. v - b', v-1-a, v-1-b) all of them are two products with one shifted (Hankel) operand, contraction index
// k = (jj, rho):
//     Z1[(d,c)][j] = sum_k G_(jj+d)[rho][c]  . [Q G_jj | Q S_jj][rho][j]   -> Ht[a'][v] | H[a][v-1]   (row c, column j)
//     Z2[(d,c)][i] = sum_k QG_(jj+d)[rho][c] . [G_jj   | S_jj  ][rho][i]   -> Ht[v][b'] | H[v-1][b]   (row i, column c)
// (k < (L-1) r for the plain columns, k < K = L r for the cumulated ones), plus the corner H[v-1][v-1] = sum S^T Q S,
// which is Z1 with operand rows S_jj.  Both operands of a row or column are contiguous in k in LDS (G, S, QG, QS are
// stored one vector per input column), so every operand of v_mfma_f64_16x16x4_f64 is one LDS read.
// The right operand has only 2 m columns.  A shift of d by one is a shift of k by r in the Hankel operand, so ND =
// 16 / (2 m) consecutive values of d share one operand row: operand row (dq, c) is G[c][dq ND r + k'], k' < K + (ND-1) r,
// and column (dd, j) of the right operand is column j delayed by dd r (zero outside its range); the product then holds
// Z[(dq ND + dd, c)][j] in column (dd, j), and all 16 columns of the matrix core are used.  A wavefront takes 16 operand
// rows at a time, all wavefronts of the workgroup take part.  Needs 2 m <= 16.
// Results: the border goes to LDS (vcol: the column block the first sweep step reads as its predecessor; vrow: the
// table the last block row reads; both [i*m + j][block index], v + 1 entries each); H's last block column and row go
// straight to memory with their share of W3 (the sweep skips those positions).
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void held_sums_mfma(const double *sG, const double *sS, const double *sQG, const double *sQS,
                                               double *vcol, double *vrow, double *gH, const double *gW3, int *next_item,
                                               int m, int r, int f, int v, int tid) {
    const int lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int FR = f * r, nl = f - v + 1, K = nl * r, VM = v * m, ldh = v + 1;
    const int ND = 16 / (2 * m), NQ = (v + ND - 1) / ND;        // values of d per operand row, operand row groups
    const int rows = (NQ + 1) * m, nt = (rows + 15) >> 4;       // the last m rows are the S rows of the corner
    const int Kx = K + (ND - 1) * r;                            // contraction length with the delayed columns
    // right-operand column of this lane: (dd, jt); jt < m pairs with the plain Markov vectors, else with the cumulated ones
    const int dd = l15 / (2 * m), jt = l15 - dd * 2 * m;
    const bool bok = dd < ND, bg = jt < m;
    const int jb = bg ? jt : jt - m;
    const int koff = dd * r, kend = koff + (bg ? K - r : K);    // this column is non-zero for koff <= k' < kend
    const double *pb1 = (bg ? sQG + jb * FR : sQS + jb * K) - koff;  // Z1: right operand [Q G | Q S]
    const double *pb2 = (bg ? sG + jb * FR : sS + jb * K) - koff;    // Z2: right operand [G | S]
    // Work item = a tile of 16 operand rows, BOTH products: the two are independent chains, so one wavefront has
    // twice the LDS reads in flight and four products to issue per round trip (a lone wavefront per SIMD is bound by
    // that round trip, not by the matrix core).  Whichever wavefront is free takes the next tile.
    for (;;) {
        const int tile = __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(next_item, 1) : 0);
        if (tile >= nt) break;
        const int R = tile * 16 + l15;
        const bool rowok = R < rows;
        const int dq = rowok ? R / m : 0, c = rowok ? R - dq * m : 0;
        const bool srow = (dq == NQ);
        const double *pa1 = srow ? sS + c * K : sG + c * FR + dq * ND * r;    // Z1: Hankel operand from G (S for the corner)
        const double *pa2 = srow ? sQS + c * K : sQG + c * FR + dq * ND * r;  // Z2: Hankel operand from Q G
        const int klim = srow ? K : Kx;  // (rows whose d runs past v-1 read on into the next vector: finite, discarded)
        // where this lane's results go: memory (with W3, read now and used after the contraction) or the border
        double w3a[4], w3b[4];
        int ega[4], egb[4];  // element of H (v m <= 1 232: the index fits)
        int ta[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int Ri = tile * 16 + kq + 4 * reg;
            const int dqi = Ri / m, ci = Ri - dqi * m, di = dqi * ND + dd;
            ega[reg] = -1;
            egb[reg] = -1;
            ta[reg] = -1;
            if (Ri < rows && bok) {
                if (dqi == NQ) {
                    if (dd == 0 && !bg) ega[reg] = ((v - 1) * m + jb) * VM + (v - 1) * m + ci;  // corner
                } else if (di < v) {
                    if (bg) ta[reg] = v - di;  // a' resp. b'
                    else if (di > 0) {
                        ega[reg] = ((v - 1) * m + jb) * VM + (v - 1 - di) * m + ci;  // H[a][v-1][ci][jb]
                        egb[reg] = ((v - 1 - di) * m + ci) * VM + (v - 1) * m + jb;  // H[v-1][b][jb][ci]
                    }
                }
            }
            w3a[reg] = ega[reg] >= 0 ? gW3[ega[reg]] : 0.0;
            w3b[reg] = egb[reg] >= 0 ? gW3[egb[reg]] : 0.0;
        }
        d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = a0, b0 = a0, b1 = a0;  // two chains per product
        // Four steps of four per round.  A round is masked where a live row or column can be outside its range (the
        // delayed columns' first (ND-1) r terms; everything from K - r on); the stretch between takes plain reads at
        // constant offsets.  (Dead rows and columns read valid, finite data there: a row or column of the product
        // depends on its own operand row or column only.)
        auto masked_half = [&](int k0, const double *pa, const double *pb, d4 &c0, d4 &c1) {
            double x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k0 + 4 * u + kq;
                const bool rk = rowok && kk < klim, ck = bok && kk >= koff && kk < kend;
                const double xv = pa[rk ? kk : 0], yv = pb[ck ? kk : koff];
                x[u] = rk ? xv : 0.0;
                y[u] = ck ? yv : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], y[u], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u + 1], y[u + 1], c1, 0, 0, 0);
            }
        };
        auto masked_round = [&](int k0) {
            masked_half(k0, pa1, pb1, a0, a1);
            masked_half(k0, pa2, pb2, b0, b1);
        };
        const int khead = ((ND - 1) * r + 15) & ~15;
        int k0 = 0;
        for (; k0 < khead && k0 < Kx; k0 += 16) masked_round(k0);
        {
            const double *pa1l = pa1 + kq, *pb1l = pb1 + kq, *pa2l = pa2 + kq, *pb2l = pb2 + kq;
            for (; k0 + 8 <= K - r; k0 += 8) {  // half rounds: eight reads, four products
                const double x10 = pa1l[k0], x11 = pa1l[k0 + 4], y10 = pb1l[k0], y11 = pb1l[k0 + 4];
                const double x20 = pa2l[k0], x21 = pa2l[k0 + 4], y20 = pb2l[k0], y21 = pb2l[k0 + 4];
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x10, y10, a0, 0, 0, 0);
                b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x20, y20, b0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x11, y11, a1, 0, 0, 0);
                b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x21, y21, b1, 0, 0, 0);
            }
        }
        for (; k0 < Kx; k0 += 16) masked_round(k0);
        a0 += a1;
        b0 += b1;
        // D[i][j]: this lane holds column j = l15 and rows i = kq + 4 reg of the tile
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int Ri = tile * 16 + kq + 4 * reg;
            const int ci = Ri % m;
            if (ta[reg] >= 0) {
                vcol[(ci * m + jb) * ldh + ta[reg]] = a0[reg];  // Ht[a'][v][ci][jb]
                vrow[(jb * m + ci) * ldh + ta[reg]] = b0[reg];  // Ht[v][b'][jb][ci]
            }
            if (ega[reg] >= 0) gH[ega[reg]] = a0[reg] + w3a[reg];
            if (egb[reg] >= 0) gH[egb[reg]] = b0[reg] + w3b[reg];
        }
    }
}

// fvec = Gamma^T z on the matrix cores.  Column (k, c) of Gamma, k < v-1, is the Markov vector c shifted down by k r, so
//     fvec[k m + c] = sum_t G[c][t] . z[t + k r]          (t < f r - k r)
// is a product with a shifted (Hankel) operand in z: operand row kg holds z[kg ND r + t'], and column (dd, c) of the other
// operand is G[c] delayed by dd r (ND = 16 / m consecutive shifts share an operand row, so all 16 columns of
// v_mfma_f64_16x16x4_f64 are used); D[kg][(dd, c)] = fvec[(kg ND + dd) m + c].  Two LDS reads feed 1 024 multiply-adds
// where the per-column dot products read two values per multiply-add.  The output is tiny (v m values) and the
// contraction long, so the wavefronts split the contraction index: `chunks` partial tiles per 16 operand rows go to
// LDS and are added in a fixed order afterwards.  The held-input column (k = v-1, cumulated vector S in place of G)
// is cut into 64 / m slices summed by one wavefront, likewise added afterwards.
// fvec_chunks: partial tiles per operand-row tile that fit `cap` doubles of LDS; 0: use the dot products instead.
__host__ __device__ inline int fvec_chunks(int m, int v, int cap) {
    if (m > 16 || v < 2) return 0;
    const int ND = 16 / m, NQ = (v - 1 + ND - 1) / ND, tiles = (NQ + 15) >> 4;
    int ch = kLoaders / tiles;
    if (ch < 1) ch = 1;
    while (ch > 1 && ch * NQ * 16 + 64 > cap) --ch;
    return (ch * NQ * 16 + 64 <= cap) ? ch : 0;
}
__device__ __forceinline__ void fvec_partials(const double *sG, const double *sS, const double *sZ, double *part,
                                              int chunks, int m, int r, int f, int v, int lw, int lane) {
    const int l15 = lane & 15, kq = lane >> 4, FR = f * r, nl = f - v + 1;
    const int ND = 16 / m, NQ = (v - 1 + ND - 1) / ND, tiles = (NQ + 15) >> 4;
    const int ddr = l15 / m;
    const bool bok = ddr < ND;
    const int dd = bok ? ddr : 0, c = bok ? l15 - dd * m : 0;
    const double *pb = sG + c * FR - dd * r;  // pb[t'] = G[c][t' - dd r]
    for (int item = lw; item < tiles * chunks; item += kLoaders) {
        const int T = item / chunks, ch = item - T * chunks;
        const int kg = T * 16 + l15;
        const bool rowok = kg < NQ;
        const int zoff = (rowok ? kg : 0) * ND * r;  // operand row: z[zoff + t'], zero from f r on
        const int Kx = FR - T * 16 * ND * r + (ND - 1) * r;
        const int rounds = (Kx + 15) >> 4;
        const int r0 = rounds * ch / chunks, r1 = rounds * (ch + 1) / chunks;
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = (d4){0.0, 0.0, 0.0, 0.0};
        for (int rd = r0; rd < r1; ++rd) {
            double x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = rd * 16 + 4 * u + kq;
                const bool ak = rowok && zoff + t < FR, bk = bok && t >= dd * r && t - dd * r < FR;
                const double xv = sZ[ak ? zoff + t : 0], yv = pb[bk ? t : dd * r];
                x[u] = ak ? xv : 0.0;
                y[u] = bk ? yv : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], y[u], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u + 1], y[u + 1], acc1, 0, 0, 0);
            }
        }
        acc += acc1;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {  // D[i][j]: this lane holds column j = l15 and rows i = kq + 4 reg
            const int kgi = T * 16 + kq + 4 * reg;
            if (kgi < NQ) part[(ch * NQ + kgi) * 16 + l15] = acc[reg];
        }
    }
    if (lw == kLoaders - 1) {  // held-input column: slice sl of S[c] . z[(v-1) r ...]
        const int nsl = 64 / m, sl = lane / m, cc = lane - sl * m, len = nl * r, seg = (len + nsl - 1) / nsl;
        double sum = 0.0;
        if (sl < nsl) {
            const double *src = sS + cc * len, *zz = sZ + (v - 1) * r;
            const int t1 = (sl + 1) * seg < len ? (sl + 1) * seg : len;
            for (int t = sl * seg; t < t1; ++t) sum += src[t] * zz[t];
        }
        part[chunks * NQ * 16 + lane] = sum;
    }
}
__device__ __forceinline__ void fvec_finish(const double *part, double *gf, int chunks, int m, int v, int tid) {
    const int ND = 16 / m, NQ = (v - 1 + ND - 1) / ND, nsl = 64 / m;
    for (int idx = tid; idx < v * m; idx += kWgBlock) {
        const int k = idx / m, c = idx - k * m;
        double sum = 0.0;
        if (k < v - 1) {
            const int kg = k / ND, dd = k - kg * ND;
            for (int ch = 0; ch < chunks; ++ch) sum += part[(ch * NQ + kg) * 16 + dd * m + c];
        } else {
            for (int sl = 0; sl < nsl; ++sl) sum += part[chunks * NQ * 16 + sl * m + c];
        }
        gf[idx] = sum;
    }
}

// The loader waves' W3 pipeline.  Step s of the sweep (s = 0 .. v-1) uses column block v-1-s of W3 (wcnt doubles,
// contiguous).  A block moves as 16-byte pairs in slots of 64 pairs (1 KiB, one load instruction of a wave); slot g
// of every block belongs to loader wave g % kLoaders.  A wave fetches its slots of a block into registers NB steps
// before the step that reads it and writes them to the block's LDS ring slot one step before, so a fetch has NB - 1
// whole steps to arrive.  The depth matters because a CU's vector-memory pipeline is in order: while the compute
// waves (and the other workgroup on the CU) keep it full of stores, a load queues behind them and returns
// microseconds later.  A wave has kPreTotal register slots, i.e. NB = kPreTotal / (its slots per block) blocks in
// flight.  Every loader wave runs the same straight-line sequence (commit, fetch, barrier), which lets the compiler
// wait for exactly the oldest block.  Blocks of odd length or unaligned base, or too large for the registers, are
// copied in place (latency exposed).
constexpr int kPreTotal = 26;
typedef double v2d __attribute__((ext_vector_type(2)));  // a first-class 16-byte value (plain loads / stores, no struct copies)

__device__ __forceinline__ unsigned opaque(unsigned x) {  // x unchanged, but not hoistable / not foldable past this point
    asm volatile("" : "+v"(x));
    return x;
}
// The part of one column block of W3 that the sweep has to fetch: rows [row0, row0 + 2 wp) of each of its m columns
// (n2 = m wp 16-byte pairs).  W3 as the reference builds it (W1^T W2 W1, controller.rs:149-177) is block
// tridiagonal, so most of a column block is exact zeros.  w3_band_kernel measures how far the non-zeros of any column
// block bb reach above and below row bb m; the loader then moves, for block bb, only a window of fixed height that
// covers that reach for block bb and for block bb + 2 (the previous occupant of the ring slot, whose non-zeros the
// window must overwrite): the rest of the slot already holds zeros.  A dense W3 gives the full block as window.
// magic = floor(2^32 / wp) + 1 turns the division of a pair index (< 2^16) by wp (< 2^16) into a multiply-high.
struct W3Win {
    int row0, wp, n2;
    unsigned magic;
};
struct W3Band {  // wave-uniform, set up once per loader wave
    int up, wp, m, VM;  // reach above row bb m (may be negative), window height in pairs
    unsigned magic;
    bool full;          // the window is the whole column block
    __device__ __forceinline__ W3Win window(int bb) const {
        int r0 = (bb * m - up) & ~1;
        const int top = VM - 2 * wp;
        r0 = r0 < 0 ? 0 : (r0 > top ? top : r0);
        W3Win w;
        w.row0 = full ? 0 : r0; w.wp = wp; w.n2 = m * wp; w.magic = magic;
        return w;
    }
};
// reach[0], reach[1]: largest (bb m - first non-zero row) and (one past the last non-zero row - bb m) over the column
// blocks of W3, INT_MIN where no block has a non-zero; null: no measurement, the window is the whole block.
__device__ __forceinline__ W3Band w3_band_setup(const int *reach, int v, int m) {
    W3Band b;
    b.m = m; b.VM = v * m;
    int up = 0, down = 0;
    bool full = true;
    if (reach) {
        up = reach[0]; down = reach[1];
        if (up == INT_MIN) { up = 0; down = 0; }
        full = false;
    }
    int span = (up + down + 2 * m + 2) & ~1;  // both blocks' reach, and one row lost to the even start
    if (span < 2) span = 2;
    if (full || span >= b.VM) { full = true; span = b.VM; }
    b.up = up; b.wp = span >> 1; b.full = full;
    b.magic = b.wp > 1 ? (unsigned)(0x100000000ULL / (unsigned)b.wp) + 1u : 0u;
    return b;
}
// byte offset, within a column block, of pair e of the window
__device__ __forceinline__ unsigned w3_pair_offset(unsigned e, const W3Win &w, int VM) {
    const unsigned c = w.wp == 1 ? e : __umulhi(e, w.magic);
    const unsigned q = e - c * (unsigned)w.wp;
    return (c * (unsigned)VM + (unsigned)w.row0) * 8u + q * 16u;
}
// Every fetch issues exactly SPB loads and every step runs the same straight-line code (no branch decides whether a
// load is issued): the compiler can then count the loads in flight and wait for the oldest block only.  A slot past
// the window's end re-reads the window's last pair, a block index past the last step re-reads the last block; what
// they deliver is never used.
template <int SPB>
__device__ __forceinline__ void w3_fetch(v2d (&r)[SPB], const double *src, const W3Win &w, int VM, int lw, unsigned lane) {
    const unsigned l = opaque(lane);  // keeps the per-slot offsets from being precomputed and held in registers
    const unsigned last = (unsigned)w.n2 - 1u;
#pragma unroll
    for (int u = 0; u < SPB; ++u) {
        const unsigned e = 64u * (unsigned)(u * kLoaders + lw) + l;  // pair index: wave-uniform slot base + lane
        r[u] = *(const v2d *)((const char *)src + w3_pair_offset(e < last ? e : last, w, VM));
    }
}
template <int SPB>
__device__ __forceinline__ void w3_commit(const v2d (&r)[SPB], double *dst, const W3Win &w, int VM, int lw, unsigned lane) {
    const unsigned l = opaque(lane);
#pragma unroll
    for (int u = 0; u < SPB; ++u) {
        const unsigned e = 64u * (unsigned)(u * kLoaders + lw) + l;
        if ((int)e < w.n2) *(v2d *)((char *)dst + w3_pair_offset(e, w, VM)) = r[u];
    }
}
// The loader waves' side of the sweep: v steps, one lds_barrier() each, matching the compute waves' loop.
// The steps of one round (NB of them) are instantiated with compile-time positions, so that every register set is
// addressed statically and stays in registers.
template <int NB>
struct W3Loader {
    static constexpr int NBR = NB > 0 ? NB : 1;
    static constexpr int SPB = NB > 0 ? kPreTotal / NBR : 1;
    using Regs = v2d[NBR][SPB];
    const double *gW3;
    double *sW3;
    W3Band band;
    int v, wcnt, VM, m, lw;
    unsigned ltid;

    __device__ __forceinline__ int blk_of(int step) const { return v - 1 - (step < v ? step : v - 1); }
    __device__ __forceinline__ const double *block_of(int step) const { return gW3 + (long long)blk_of(step) * wcnt; }
    __device__ __forceinline__ W3Win win_of(int step) const { return band.window(blk_of(step)); }
    template <int U>
    __device__ __forceinline__ void prefetch(Regs &reg) const {  // U = 1 .. NB: the block of step U, into set U % NB
        if (NB > 0) w3_fetch<SPB>(reg[U % NBR], block_of(U), win_of(U), VM, lw, ltid);
    }
    // Step s0 + U: the block of the next step goes to the ring slot that step reads (past the last step this writes
    // the slot nobody reads any more), and its register set is refilled with the block NB steps further on.
    template <int U>
    __device__ __forceinline__ void step(Regs &reg, int s0) const {
        const int s = s0 + U;
        constexpr int SET = (U + 1) % NBR;
        double *dst = sW3 + ((v - 2 - s) & 1) * wcnt;
        if (NB > 0) {
            w3_commit<SPB>(reg[SET], dst, win_of(s + 1), VM, lw, ltid);
            w3_fetch<SPB>(reg[SET], block_of(s + 1 + NBR), win_of(s + 1 + NBR), VM, lw, ltid);
        } else if (s + 1 < v) {
            const double *src = block_of(s + 1);
            for (int e = ltid + 64 * lw; e < wcnt; e += 64 * kLoaders) dst[e] = src[e];
        }
        lds_barrier();
    }
    template <int U>
    __device__ __forceinline__ void tail_step(Regs &reg, int s0, int rem) const {
        if (U < rem) step<U>(reg, s0);
    }
    template <int... U>
    __device__ __forceinline__ void run(std::integer_sequence<int, U...>) const {
        Regs reg;
        (prefetch<U + 1>(reg), ...);
        int s0 = 0;
        for (; s0 + NBR <= v; s0 += NBR) (step<U>(reg, s0), ...);
        (tail_step<U>(reg, s0, v - s0), ...);
    }
};
template <int NB>
__device__ __forceinline__ void w3_loader_sweep(const double *gW3, double *sW3, const W3Band &band, int v, int m, int lw,
                                                unsigned ltid) {
    W3Loader<NB> L;
    L.gW3 = gW3; L.sW3 = sW3; L.band = band; L.v = v; L.m = m; L.VM = v * m; L.wcnt = v * m * m; L.lw = lw; L.ltid = ltid;
    L.run(std::make_integer_sequence<int, W3Loader<NB>::NBR>{});
}
__device__ __forceinline__ void w3_loader_run(const double *gW3, double *sW3, const int *reach, int v, int m, int lw, int ltid,
                                              bool wide) {
    lw = __builtin_amdgcn_readfirstlane(lw);  // wave-uniform by construction; tell the compiler
    const W3Band band = w3_band_setup(wide ? reach : nullptr, v, m);
    // slots of a window that one wave moves
    const int spw = ((m * band.wp + 63) / 64 + kLoaders - 1) / kLoaders;
    if (!wide || spw > kPreTotal) w3_loader_sweep<0>(gW3, sW3, band, v, m, lw, (unsigned)ltid);
    else if (spw <= kPreTotal / 8) w3_loader_sweep<8>(gW3, sW3, band, v, m, lw, (unsigned)ltid);
    else if (spw <= kPreTotal / 5) w3_loader_sweep<5>(gW3, sW3, band, v, m, lw, (unsigned)ltid);
    else if (spw <= kPreTotal / 3) w3_loader_sweep<3>(gW3, sW3, band, v, m, lw, (unsigned)ltid);
    else if (spw <= kPreTotal / 2) w3_loader_sweep<2>(gW3, sW3, band, v, m, lw, (unsigned)ltid);
    else w3_loader_sweep<1>(gW3, sW3, band, v, m, lw, (unsigned)ltid);
}

// One workgroup per column of W3 ((v m) x (v m), column-major; column block bb = column / m): how far the column's
// non-zeros (NaNs included) reach above and below row bb m, folded into reach[0] / reach[1] by atomic maximum (the
// caller sets both to INT_MIN first).
__global__ __launch_bounds__(256) void w3_band_kernel(const double *__restrict__ W3, int v, int m, int *reach) {
    __shared__ int slo, shi;
    const int col = blockIdx.x, VM = v * m, bb = col / m;
    if (threadIdx.x == 0) { slo = VM; shi = 0; }
    __syncthreads();
    int lo = VM, hi = 0;
    const double *src = W3 + (long long)col * VM;
    for (int row = threadIdx.x; row < VM; row += 256) {
        if (!(src[row] == 0.0)) {
            lo = row < lo ? row : lo;
            hi = row + 1 > hi ? row + 1 : hi;
        }
    }
    if (hi > lo) {
        atomicMin(&slo, lo);
        atomicMax(&shi, hi);
    }
    __syncthreads();
    if (threadIdx.x == 0 && shi > slo) {
        // atomics on one address serialise in the L2: look first, most columns of a banded W3 have nothing to add
        const int up = bb * m - slo, down = shi - bb * m;
        if (__atomic_load_n(&reach[0], __ATOMIC_RELAXED) < up) atomicMax(&reach[0], up);
        if (__atomic_load_n(&reach[1], __ATOMIC_RELAXED) < down) atomicMax(&reach[1], down);
    }
}

// HM, HR, ITEMS > 0: the H sweep is specialised for m = HM, r = HR with up to ITEMS (block row, column) pairs per
// thread held in registers (needs v m <= ITEMS * kCompute).  HM = 0: generic sweep, operands read from LDS.
// HELD: f > v (the last block column is the held-input column); false removes the table look-ups
// from the specialised sweep, and with them a dozen live registers.
template <int HM, int HR, int ITEMS, bool HELD>
__global__ __launch_bounds__(kWgBlock, 4) void qp_wg_kernel(QpArgs p) {
    extern __shared__ __align__(16) double lds[];
    // Diagnostic build (-DMPCX_WG_STAMPS, see tools/probe_phases.py): thread 0 records the 100 MHz wall clock at the
    // phase boundaries into the caller's fvec buffer (v m >= 10 doubles per instance), and fvec itself is not computed.
    // -DMPCX_WG_STAMPS2 keeps fvec and records into the caller's Phi buffer instead (Phi is then not computed), with three
    // more stamps taken by the first loader wave around fvec and the held-input sums (STAMPS2=1 in tools/probe_phases.py).
#ifdef MPCX_WG_STAMPS2
    double *stampbuf = p.Phi + (long long)blockIdx.x * p.f * (HR ? HR : p.r) * p.n;
    p.Phi = nullptr;
#define STAMP(k) do { if (threadIdx.x == 0) stampbuf[k] = (double)wall_clock64(); } while (0)
#define LSTAMP(k) do { if (threadIdx.x == kCompute) stampbuf[k] = (double)wall_clock64(); } while (0)
#elif defined(MPCX_WG_STAMPS)
    double *stampbuf = p.fvec + (long long)blockIdx.x * p.v * (HM ? HM : p.m);
    p.fvec = nullptr;
#define STAMP(k) do { if (threadIdx.x == 0) stampbuf[k] = (double)wall_clock64(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
#ifndef LSTAMP
#define LSTAMP(k) do {} while (0)
#endif
    STAMP(0);
    const long long b = blockIdx.x;
    const int tid = threadIdx.x;
    const bool loader = tid >= kCompute;  // wave-uniform: the waves after the four compute waves
    const int ltid = (tid - kCompute) & 63, lw = (tid - kCompute) >> 6;  // lane and index of the loader wave
    const int n = p.n, m = HM ? HM : p.m, r = HR ? HR : p.r, f = p.f, v = p.v;
    const int FR = f * r, VM = v * m, nl = f - v + 1;
    const WgLds L = wg_lds_layout(n, m, r, f, v);
    double *sA = lds + L.A, *sB = lds + L.B, *sC = lds + L.C, *sx0 = lds + L.x0, *sQ = lds + L.Q;
    double *sW = lds + L.Wst, *sP = lds + L.Pst, *sG = lds + L.G, *sS = lds + L.S, *sQG = lds + L.QG;
    double *sQS = lds + L.QS, *sSr = lds + L.Sr, *sZ = lds + L.Z, *sHp = lds + L.Hp, *sW3 = lds + L.W3r;
    double *sTab = lds + L.Tab, *sAx = sx0 + n;
    int *sCnt = (int *)(lds + L.Cnt);

    const double *gA = p.A + b * (long long)(n * n);
    const double *gB = p.B + b * (long long)(n * m);
    const double *gC = p.C + b * (long long)(r * n);
    const bool need_q = p.H || p.fvec;

    // ---- every per-instance input goes to LDS before the first store of the workgroup ----
    if (tid == 0) *sCnt = 0;
    const bool tables = n <= kTabN;  // wave-uniform: power tables instead of the step-by-step recurrences
    if (n * n <= kWgBlock && r * n <= kWgBlock && n * m <= kWgBlock && r * r <= kWgBlock && FR <= 2 * kWgBlock) {
        // every array fits one element per thread (ref: two): all loads issue before the first LDS write, so the
        // workgroup waits for memory once (a load queues behind the stores of the other workgroup on the CU:
        // microseconds each time)
        const bool wf = p.fvec != nullptr;
        const double *gref = wf ? p.ref + b * (long long)FR : nullptr;
        const double va = tid < n * n ? gA[tid] : 0.0;
        const double vb = tid < n * m ? gB[tid] : 0.0;
        const double vc = tid < r * n ? gC[tid] : 0.0;
        const double vq = (need_q && tid < r * r) ? p.W4[tid] : 0.0;
        const double vx = (wf && tid < n) ? p.x0[b * n + tid] : 0.0;
        const double vr0 = (wf && tid < FR) ? gref[tid] : 0.0;
        const double vr1 = (wf && tid + kWgBlock < FR) ? gref[tid + kWgBlock] : 0.0;
        if (tid < n * n) sA[tid] = va;
        if (tid < n * m) { sB[tid] = vb; if (!tables) sW[tid] = vb; }
        if (tid < r * n) { sC[tid] = vc; if (!tables) sP[tid] = vc; }
        if (need_q && tid < r * r) sQ[tid] = vq;
        if (wf && tid < n) sx0[tid] = vx;
        if (wf && tid < FR) sSr[tid] = vr0;
        if (wf && tid + kWgBlock < FR) sSr[tid + kWgBlock] = vr1;
    } else {
        for (int e = tid; e < n * n; e += kWgBlock) sA[e] = gA[e];
        for (int e = tid; e < n * m; e += kWgBlock) { const double x = gB[e]; sB[e] = x; if (!tables) sW[e] = x; }
        for (int e = tid; e < r * n; e += kWgBlock) { const double x = gC[e]; sC[e] = x; if (!tables) sP[e] = x; }
        if (p.fvec) {
            for (int e = tid; e < n; e += kWgBlock) sx0[e] = p.x0[b * n + e];
            for (int e = tid; e < FR; e += kWgBlock) sSr[e] = p.ref[b * (long long)FR + e];
        }
        if (need_q) for (int e = tid; e < r * r; e += kWgBlock) sQ[e] = p.W4[e];
    }
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
    if (tables) {
        // ---- power tables by doubling, then every block from one product on the matrix cores ----
        const WgTab T = wg_tab_layout(n, m, r, f);
        double *tb = lds + L.Pow;
        double *sPP = tb + T.PP, *sTC = tb + T.TC, *sX = tb + T.X, *sY = tb + T.Y;
        const int Lp = T.Lp, nh = T.nh, RP = T.RP, nn = n * n, m1 = m + 1;
        // sTC[q * RP + rho] = (C A^lo)[a][q], rho = lo r + a;  sX[h] = A^(Lp h + 1) (n x n, column-major);
        // sY[h] = [A^(Lp h) B | A^(Lp h + 1) x0] (n x (m + 1), column-major; the last column only with fvec)
        for (int e = tid; e < n * RP; e += kWgBlock) {
            const int q = e / RP, rho = e - q * RP;
            sTC[e] = (rho < r) ? sC[q * r + rho] : 0.0;
        }
        for (int e = tid; e < nn; e += kWgBlock) { const double x = sA[e]; sPP[e] = x; sX[e] = x; }
        for (int e = tid; e < n * m1; e += kWgBlock) sY[e] = (e < n * m) ? sB[e] : (p.fvec ? sAx[e - n * m] : 0.0);
        lds_barrier();
        // All products of the rounds and of the block phase are 16-row tiles on the matrix cores, transposed so that
        // lanes run along the rows (conflict-free LDS accesses, contiguous global stores):
        //   D[j2][row] = sum_q Yt[j2][q] . X[row0 + row][q],  Yt[j2][q] = Y[j2 * n + q] (n x ncol, column-major),
        //   X[row][q] = X[q * ldx + row].  A wavefront takes one tile; operand reads all issue before the first product.
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
        const int KS = (n + 3) / 4;
        auto tile = [&](const double *X, int ldx, int row0, int nrow, const double *Y, int ncol) -> d4 {
            double av[kTabN / 4], bv[kTabN / 4];
#pragma unroll
            for (int ks = 0; ks < kTabN / 4; ++ks) {
                const int q = 4 * ks + kq, qq = q < n ? q : 0;
                const double a0 = Y[(l15 < ncol ? l15 : 0) * n + qq];
                const double b0 = X[qq * ldx + row0 + (l15 < nrow ? l15 : 0)];
                av[ks] = (q < n && l15 < ncol) ? a0 : 0.0;
                bv[ks] = (q < n && l15 < nrow) ? b0 : 0.0;
            }
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < kTabN / 4; ++ks)
                if (ks < KS) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks], acc, 0, 0, 0);
            return acc;  // acc[reg] = D[j2 = kq + 4 reg][row = l15]
        };
        auto put = [&](double *dst, int ldd, int row0, int nrow, int ncol, d4 acc) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (kq + 4 * reg < ncol && l15 < nrow) dst[(kq + 4 * reg) * ldd + row0 + l15] = acc[reg];
        };
        constexpr int NWV = kWgBlock / 64;
        int cur = 0;
        for (int g = 1; g < Lp; g *= 2) {  // rows [g r, 2 g r) of the C A^lo table from rows [0, g r); A^g squared
            const double *Ap = sPP + cur * nn;
            const int tiles_tc = (g * r + 15) / 16;
            for (int t = wave; t < 1 + tiles_tc; t += NWV) {
                if (t == 0) {
                    put(sPP + (cur ^ 1) * nn, n, 0, n, n, tile(Ap, n, 0, n, Ap, n));
                } else {
                    const int row0 = (t - 1) * 16, nrow = (g * r - row0 < 16) ? g * r - row0 : 16;
                    put(sTC + g * r, RP, row0, nrow, n, tile(sTC, RP, row0, nrow, Ap, n));
                }
            }
            cur ^= 1;
            lds_barrier();
        }
        for (int g = 1; g < nh; g *= 2) {  // entries [g, 2 g) of the hi tables from entries [0, g); (A^Lp)^g squared
            const double *Mp = sPP + cur * nn;
            const int cnt = (2 * g <= nh ? g : nh - g), sq = (2 * g < nh) ? 1 : 0;
            for (int t = wave; t < sq + 2 * cnt; t += NWV) {
                if (t < sq) {
                    put(sPP + (cur ^ 1) * nn, n, 0, n, n, tile(Mp, n, 0, n, Mp, n));
                } else {
                    const int u = t - sq, hh = u >> 1;
                    if (u & 1) put(sY + (g + hh) * n * m1, n, 0, n, m1, tile(Mp, n, 0, n, sY + hh * n * m1, m1));
                    else put(sX + (g + hh) * nn, n, 0, n, n, tile(Mp, n, 0, n, sX + hh * nn, n));
                }
            }
            cur ^= 1;
            lds_barrier();
        }
        // Blocks: D[c][rho] = sum_q R_h[q][c] . (C A^lo)[rho][q],  R_h = A^(Lp h + 1) (Phi) or sY[h] (G and s).  An item is
        // one tile of 16 rows rho for one h: its (C A^lo) operand is read once and serves the Phi product and the G / s one.
        {
            const int RT = (Lp * r + 15) / 16;
            const int cols_gs = m + (p.fvec ? 1 : 0), CT = (cols_gs + 15) / 16;  // column tiles of the G / s product
            auto prod = [&](const double *R, int ncol, const double (&bv)[kTabN / 4]) -> d4 {
                double av[kTabN / 4];
#pragma unroll
                for (int ks = 0; ks < kTabN / 4; ++ks) {
                    const int q = 4 * ks + kq;
                    const double a0 = R[(l15 < ncol ? l15 : 0) * n + (q < n ? q : 0)];
                    av[ks] = (q < n && l15 < ncol) ? a0 : 0.0;
                }
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < kTabN / 4; ++ks)
                    if (ks < KS) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks], acc, 0, 0, 0);
                return acc;
            };
            for (int it = wave; it < nh * RT; it += NWV) {
                const int h = it / RT, rt = it - h * RT;  // wave-uniform
                double bv[kTabN / 4];
#pragma unroll
                for (int ks = 0; ks < kTabN / 4; ++ks) {
                    const int q = 4 * ks + kq;
                    const double b0 = sTC[(q < n ? q : 0) * RP + rt * 16 + l15];
                    bv[ks] = (q < n) ? b0 : 0.0;
                }
                const int rho = rt * 16 + l15, row = Lp * h * r + rho;
                const bool live = rho < Lp * r && row < FR;
                if (gPhi) {
                    const d4 acc = prod(sX + h * nn, n, bv);
                    if (live) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            if (kq + 4 * reg < n) gPhi[(long long)(kq + 4 * reg) * FR + row] = acc[reg];
                    }
                }
                for (int ct = 0; ct < CT; ++ct) {
                    const int ncol = cols_gs - ct * 16;
                    const d4 acc = prod(sY + h * n * m1 + ct * 16 * n, ncol < 16 ? ncol : 16, bv);
                    if (live) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int c = ct * 16 + kq + 4 * reg;
                            if (c < m) sG[c * FR + row] = acc[reg];
                            else if (c == m && p.fvec) sSr[row] = acc[reg] - sSr[row];
                        }
                    }
                }
            }
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
    STAMP(1);

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
    //      Only where Gamma is at least as large as H (f r >= v m), and not with held inputs (f > v): there the
    //      other wavefronts form fvec and the held-input sums while Gamma is stored, which hides both. ----
    //      With held inputs the alternation is on only for a batch of at most two rounds of resident workgroups: those
    //      run in lockstep (all store Gamma at once, all sweep at once), and what alternates is then the CU slot
    //      (workgroups i and i + #CU share a CU in the first round; i and i + 1 sit on different XCDs).
    const bool lockstep = p.cu_count > 0 && gridDim.x <= 4u * (unsigned)p.cu_count;
    const bool gamma_last = p.Gamma && p.H && FR >= VM &&
                            ((f > v) ? (lockstep && ((blockIdx.x / (unsigned)p.cu_count) & 1)) : (blockIdx.x & 1));
    auto write_gamma = [&]() {
        // A wavefront takes a column at a time, lanes along its rows.  The LDS reads of a stretch of rows all issue
        // before the first store (one LDS round trip per stretch, not per store instruction); rows above the block
        // diagonal read a valid address and store zero.
        double *gG = p.Gamma + b * (long long)FR * VM;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nwaves = kCompute >> 6;
        constexpr int U = 8;
        const bool wide16 = (((FR | r) & 1) == 0) && (((uintptr_t)p.Gamma & 15) == 0);
        for (int col = wave; col < VM; col += nwaves) {
            const int k = col / m, c = col - k * m;
            const int shift = k * r;
            const double *src = ((k == v - 1) ? (sS + c * nl * r) : (sG + c * FR)) - shift;  // src[row], row >= shift
            double *dst = gG + (long long)col * FR;
            if (wide16) {
                // two rows per lane: a wavefront stores 16 bytes per lane at more than twice the rate of 8 (measured:
                // 40 against 17 GB/s per wavefront, tools/cu_write_rate.hip).  Every count here is even, so a pair
                // lies wholly above the diagonal block or wholly in the data, and never across the end of the column.
                const v2d *src2 = (const v2d *)src;
                v2d *dst2 = (v2d *)dst;
                const int FP = FR >> 1, sp = shift >> 1;
                const v2d zero = {0.0, 0.0};
                constexpr int U2 = 4;
                int p0 = 0;
                for (; p0 + 64 * U2 <= FP; p0 += 64 * U2) {
                    v2d x[U2];
#pragma unroll
                    for (int u = 0; u < U2; ++u) {
                        const int pr = p0 + 64 * u + lane;
                        x[u] = src2[pr < sp ? sp : pr];
                    }
#pragma unroll
                    for (int u = 0; u < U2; ++u) {
                        const int pr = p0 + 64 * u + lane;
                        dst2[pr] = pr < sp ? zero : x[u];
                    }
                }
                for (; p0 < FP; p0 += 64 * U2) {  // the rest of the column, again as one stretch (reads past the end clamped)
                    v2d x[U2];
#pragma unroll
                    for (int u = 0; u < U2; ++u) {
                        const int pr = p0 + 64 * u + lane;
                        x[u] = src2[pr < sp ? sp : (pr < FP ? pr : FP - 1)];
                    }
#pragma unroll
                    for (int u = 0; u < U2; ++u) {
                        const int pr = p0 + 64 * u + lane;
                        if (pr < FP) dst2[pr] = pr < sp ? zero : x[u];
                    }
                }
                continue;
            }
            int row0 = 0;
            for (; row0 + 64 * U <= FR; row0 += 64 * U) {
                double x[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int row = row0 + 64 * u + lane;
                    x[u] = src[row < shift ? shift : row];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int row = row0 + 64 * u + lane;
                    dst[row] = row < shift ? 0.0 : x[u];
                }
            }
            for (; row0 < FR; row0 += 64 * 2) {
                double x[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int row = row0 + 64 * u + lane;
                    x[u] = src[row < shift ? shift : (row < FR ? row : FR - 1)];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int row = row0 + 64 * u + lane;
                    if (row < FR) dst[row] = row < shift ? 0.0 : x[u];
                }
            }
        }
    };
    STAMP(2);
    if (p.Gamma && !loader && !gamma_last) write_gamma();
    STAMP(3);

    // ---- fvec[col] = sum_rows Gamma[row][col] Z[row], by the loader waves: they are idle until the sweep, so with
    //      Gamma requested the dot products run beside the compute waves' Gamma store stream. ----
    LSTAMP(10);
    // The first column block of W3 the sweep needs (block v-1) is requested ahead of time by the loader waves and
    // written to its ring slot when the sweep's buffers become free: its latency passes behind fvec (without held
    // inputs) or behind the wait for the other wavefronts (with them: requested after the sums, which need the registers).
    constexpr int kW3Pre = HM > 0 ? 8 : 2;  // (the generic kernel has no registers to spare for more)
    const int wcnt0 = VM * m;
    const bool w3early = p.H && loader && wcnt0 <= kW3Pre * 64 * kLoaders;
    double w3pre[kW3Pre];
    auto request_w3 = [&]() {
        if (w3early) {
            const double *src = p.W3 + (long long)(v - 1) * wcnt0;
#pragma unroll
            for (int u = 0; u < kW3Pre; ++u) {
                const int e = tid - kCompute + u * 64 * kLoaders;
                w3pre[u] = src[e < wcnt0 ? e : 0];
            }
        }
    };
    const bool held = HM > 0 ? HELD : nl > 1;
    if (!held) request_w3();
    // (the inputs are dead by now: their LDS holds the partial tiles of the matrix-core form)
    const int fchunks = p.fvec ? fvec_chunks(m, v, L.Z - L.A) : 0;
    double *sPart = lds + L.A;
    if (fchunks > 0) {
        if (loader) fvec_partials(sG, sS, sZ, sPart, fchunks, m, r, f, v, __builtin_amdgcn_readfirstlane(lw), ltid);
    } else if (p.fvec && loader) {
        double *gf = p.fvec + b * (long long)VM;
        for (int col = tid - kCompute; col < VM; col += 64 * kLoaders) {
            const int k = col / m, c = col % m;
            const int shift = k * r;
            const double *src = (k == v - 1) ? (sS + c * nl * r) : (sG + c * FR);
            const double *zz = sZ + shift;
            const int len = FR - shift;
            double a[4] = {0.0, 0.0, 0.0, 0.0};
            int t = 0;
            for (; t + 7 < len; t += 8) {  // eight independent LDS read pairs per round: one latency, not eight
                double x[8], y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { x[u] = src[t + u]; y[u] = zz[t + u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u & 3] += x[u] * y[u];
            }
            for (; t < len; ++t) a[t & 3] += src[t] * zz[t];
            gf[col] = (a[0] + a[1]) + (a[2] + a[3]);
        }
    }
    LSTAMP(11);
    const bool mfma_sums = p.H && held && 2 * m <= 16;
    // (a wavefront coming from the Gamma stores would take its item late and wait for the item's loads behind its
    // own stores: with Gamma requested the sums are the other wavefronts' alone)
    if (mfma_sums && (loader || !p.Gamma || gamma_last))
        held_sums_mfma(sG, sS, sQG, sQS, lds + L.Vcol, sTab, p.H + b * (long long)VM * VM, p.W3, sCnt, m, r, f, v, tid);
    if (held) request_w3();
    LSTAMP(12);
    if (fchunks > 0) {
        lds_barrier();
        fvec_finish(sPart, p.fvec + b * (long long)VM, fchunks, m, v, tid);
    }
    STAMP(4);
    if (p.H) {
        lds_barrier();  // z and the inputs are dead: their LDS becomes the sweep's buffers
        // W3 pipeline of the loader wave: block v-1 goes straight to its slot, block v-2 is fetched ahead
        const int wcnt = VM * m;  // doubles in one column block of W3
        const bool wide = ((wcnt & 1) == 0) && (((uintptr_t)p.W3 & 15) == 0);
        if (loader) {  // block v-1 whole; the other slot starts as zeros (the loader then moves windows only)
            const double *src = p.W3 + (long long)(v - 1) * wcnt;
            double *dst = sW3 + ((v - 1) & 1) * wcnt, *oth = sW3 + (v & 1) * wcnt;
            if (w3early) {
#pragma unroll
                for (int u = 0; u < kW3Pre; ++u) {
                    const int e = tid - kCompute + u * 64 * kLoaders;
                    if (e < wcnt) { dst[e] = w3pre[u]; oth[e] = 0.0; }
                }
            } else {
                for (int e = tid - kCompute; e < wcnt; e += 64 * kLoaders) { dst[e] = src[e]; oth[e] = 0.0; }
            }
        }
        // The sweep runs on the Gram matrix Ht of plain-shift columns, Ht[a][b] = Ht[a+1][b+1] + G_(f-1-a)^T Q G_(f-1-b),
        // bordered by a virtual block row and column of index v: zero when f = v, and for f > v the short sums
        //     Ht[a'][v] = sum_(i2 >= v) G_(i2-a')^T Q G_(i2-v),   Ht[v][b'] = sum_(i2 >= v) G_(i2-v)^T Q G_(i2-b')   (a', b' = 1..v)
        // With the border in place every block of every step is the same recurrence: no first-column or last-row case.
        // A column block of Ht is kept as [i*m + j][a], a = 0..v (ldh = v + 1 entries, the last one the virtual row).
        // The virtual column is the "previous column" of the first step; the virtual row sits in a table of the same
        // layout (sVrow, entry b' for column block b'), from which the last block row reads its predecessor.
        // For f > v the last block row and column of H itself involve the held-input column and are not swept: they are
        // summed (with the border, on the matrix cores where 2 m <= 16) and stored before or after the sweep.
        const int bsz = m * m, ldh = v + 1, slot = bsz * ldh;
        double *sVrow = sTab;
        double *gH = p.H + b * (long long)VM * VM;
        // what step bb = v-1 reads as its previous column
        double *first_prev = mfma_sums ? lds + L.Vcol : sHp + (v & 1) * slot;
        if (mfma_sums) {
            // summed above
        } else if (held) {
            for (int t = tid; t < 2 * v * bsz; t += kWgBlock) {
                const int which = t / (v * bsz), rem = t - which * (v * bsz), blk = 1 + rem / bsz, ij = rem % bsz;
                const int i = ij / m, j = ij - i * m;
                if (which == 0) first_prev[ij * ldh + blk] = h_direct(sG, sS, sQG, sQS, blk, v, i, j, f, v, r, FR, nl, false);
                else sVrow[ij * ldh + blk] = h_direct(sG, sS, sQG, sQS, v, blk, i, j, f, v, r, FR, nl, false);
            }
        } else {
            for (int e = tid; e < slot; e += kWgBlock) first_prev[e] = 0.0;
            for (int e = tid; e < bsz; e += kWgBlock) sHp[((v - 1) & 1) * slot + e * ldh + v] = 0.0;
        }

        if constexpr (HM > 0) {
            // item it of this thread: column j of block row aa, t = tid + it * kCompute = j * v + aa
            double ga[ITEMS][HR][HM];  // G_(f-1-aa)[q][i]
            int item_aa[ITEMS], item_j[ITEMS];
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int t = tid + it * kCompute;
                const bool live = !loader && t < VM;
                if (it == 0) STAMP(5);
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
                w3_loader_run(p.W3, sW3, (const int *)p.w3win, v, m, lw, ltid, wide);
            } else {
                STAMP(6);
                for (int bb = v - 1; bb >= 0; --bb) {
                    double *cur = sHp + (bb & 1) * slot;
                    const double *prev = (bb == v - 1) ? first_prev : sHp + ((bb + 1) & 1) * slot;
                    const double *w3 = sW3 + (bb & 1) * VM * m;
                    if (bb == v / 2) STAMP(7);
                    // Three passes over the thread's items, two items at a time -- every LDS read, then the arithmetic,
                    // then the LDS writes and the stores -- so that the reads of a pair are in flight together (one LDS
                    // latency per pair instead of two per item).  A dead item reads valid addresses and writes nothing.
                    constexpr int CH = (ITEMS >= 2 && 2 * HM + HR <= 8) ? 2 : 1;  // pairs only where the operands are few
#pragma unroll
                    for (int c0 = 0; c0 < ITEMS; c0 += CH) {
                    double acc[CH][HM], w3v[CH][HM], qv[CH][HR];
#pragma unroll
                    for (int ic = 0; ic < CH; ++ic) {
                        const int it = c0 + ic;
                        const int j = item_j[it] < 0 ? 0 : item_j[it], aa = item_aa[it];
                        // predecessor Ht[aa+1][bb+1]: the previous column, or for the last block row (f > v) the virtual row
                        const double *src = (HELD && aa == v - 1) ? sVrow + j * ldh + bb + 1 : prev + j * ldh + aa + 1;
                        const double *qb = sQG + j * FR + (f - 1 - bb) * HR;
                        const double *wsrc = w3 + j * VM + aa * HM;
#pragma unroll
                        for (int i = 0; i < HM; ++i) {
                            acc[ic][i] = src[i * HM * ldh];
                            w3v[ic][i] = wsrc[i];
                        }
#pragma unroll
                        for (int q = 0; q < HR; ++q) qv[ic][q] = qb[q];
                    }
#pragma unroll
                    for (int ic = 0; ic < CH; ++ic) {
                        const int it = c0 + ic;
#pragma unroll
                        for (int q = 0; q < HR; ++q)
#pragma unroll
                            for (int i = 0; i < HM; ++i) acc[ic][i] += ga[it][q][i] * qv[ic][q];
                    }
#pragma unroll
                    for (int ic = 0; ic < CH; ++ic) {
                        const int it = c0 + ic;
                        const int j = item_j[it], aa = item_aa[it];
                        if (j < 0) continue;
                        double *dst = gH + (long long)(bb * HM + j) * VM + aa * HM;
                        double out[HM];
#pragma unroll
                        for (int i = 0; i < HM; ++i) {
                            cur[(i * HM + j) * ldh + aa] = acc[ic][i];
                            out[i] = acc[ic][i] + w3v[ic][i];
                        }
                        if (HELD && (bb == v - 1 || aa == v - 1)) continue;  // H's last block row / column: not swept
                        if constexpr (HM % 2 == 0) {
#pragma unroll
                            for (int i = 0; i < HM; i += 2) *(double2 *)(dst + i) = make_double2(out[i], out[i + 1]);
                        } else {
#pragma unroll
                            for (int i = 0; i < HM; ++i) dst[i] = out[i];
                        }
                    }
                    }
                    lds_barrier();
                }
            }
        } else {
            lds_barrier();
            // Two loops with the same number of barriers: the loader wave's and the compute waves'.
            if (loader) {
                w3_loader_run(p.W3, sW3, (const int *)p.w3win, v, m, lw, ltid, wide);
            } else {
                for (int bb = v - 1; bb >= 0; --bb) {
                    double *cur = sHp + (bb & 1) * slot;
                    const double *prev = (bb == v - 1) ? first_prev : sHp + ((bb + 1) & 1) * slot;
                    const double *w3 = sW3 + (bb & 1) * VM * m;
                    for (int t = tid; t < VM; t += kCompute) {  // t = j * v + aa, as in the specialised sweep
                        const int j = t / v, aa = t - j * v;
                        const bool lastrow = (aa == v - 1);
                        const double *src = (held && lastrow) ? sVrow + j * ldh + bb + 1 : prev + j * ldh + aa + 1;
                        const double *qb = sQG + j * FR + (f - 1 - bb) * r;
                        for (int i = 0; i < m; ++i) {
                            double acc = src[i * m * ldh];
                            const double *ga = sG + i * FR + (f - 1 - aa) * r;
                            for (int q = 0; q < r; ++q) acc += ga[q] * qb[q];
                            cur[(i * m + j) * ldh + aa] = acc;
                            if (held && (bb == v - 1 || lastrow)) continue;  // H's last block row / column: not swept
                            gH[(long long)(bb * m + j) * VM + aa * m + i] = acc + w3[j * VM + aa * m + i];
                        }
                    }
                    lds_barrier();
                }
            }
        }
        // f > v without the matrix-core path (m > 8): last block column (which = 0) and last block row (which = 1) of H,
        // with the held-input column S in place of a plain shift, summed directly (f - v + 1 terms each) and stored with
        // their share of W3.  The two meet in the corner block with the same value.
        if (held && !mfma_sums) {
            const int tsz = v * bsz;
            for (int t = tid; t < 2 * tsz; t += kWgBlock) {
                const int which = t / tsz, rem = t - which * tsz, blk = rem / bsz, ij = rem - blk * bsz;
                const int i = ij / m, j = ij - i * m;
                const int row = (which == 0 ? blk : v - 1) * m + i, col = (which == 0 ? v - 1 : blk) * m + j;
                const double hv = which == 0 ? h_direct(sG, sS, sQG, sQS, blk, v - 1, i, j, f, v, r, FR, nl, true)
                                             : h_direct(sG, sS, sQG, sQS, v - 1, blk, i, j, f, v, r, FR, nl, true);
                gH[(long long)col * VM + row] = hv + p.W3[(long long)col * VM + row];
            }
        }
    }
    STAMP(8);
    if (gamma_last && !loader) write_gamma();
    STAMP(9);
}

template <int HM, int HR, int ITEMS, bool HELD>
int launch_wg_k(const QpArgs &a, size_t lds_bytes, hipStream_t s) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)qp_wg_kernel<HM, HR, ITEMS, HELD>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((qp_wg_kernel<HM, HR, ITEMS, HELD>), dim3((unsigned)a.batch), dim3(kWgBlock), lds_bytes, s, a);
    return (int)hipGetLastError();
}

}  // namespace

// True when an instance's working set (wg_lds_layout) fits the 160 KB of LDS and the index ranges of the kernel.
bool qp_wg_takes(const QpArgs &a) {
    if (a.weights_mode != MPCX_WEIGHTS_BLOCK && (a.H || a.fvec)) return false;
    // (the plan's offsets are ints: sizes far beyond LDS must not reach it)
    const double rough = 2.0 * a.m * a.f * a.r + 2.0 * a.v * a.m * a.m + (double)a.n * a.n;
    if (rough > 1e6) return false;
    const WgLds L = wg_lds_layout(a.n, a.m, a.r, a.f, a.v);
    if ((size_t)L.total * sizeof(double) > 160 * 1024) return false;
    return (long long)a.f * a.r * a.v * a.m <= 0x7fffffffLL && a.batch <= 0x7fffffffLL;
}

int launch_qp_wg(const QpArgs &a, int hessian_algo, hipStream_t s) {
    (void)hessian_algo;
    if (a.W != 1) return MPCX_ERR_LAYOUT;
    if (!qp_wg_takes(a)) return MPCX_ERR_UNSUPPORTED;
    const WgLds L = wg_lds_layout(a.n, a.m, a.r, a.f, a.v);
    const size_t lds_bytes = (size_t)L.total * sizeof(double);
    // reach of W3's non-zeros around the block diagonal, for the sweep's loader waves (only where they move 16-byte pairs)
    QpArgs b = a;
    b.cu_count = 0;
    {
        static int cus[64];  // per device, filled on first use (0: not asked yet)
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            if (cus[dev] == 0) {
                int n = 0;
                if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus[dev] = n;
            }
            b.cu_count = cus[dev];
        }
    }
    const long long wcnt = (long long)a.v * a.m * a.m;
    if (a.H && a.W3 && a.w3win && (wcnt & 1) == 0 && ((uintptr_t)a.W3 & 15) == 0 && a.w3win_bytes >= 2 * sizeof(int)) {
        hipError_t e = hipMemsetD32Async((hipDeviceptr_t)a.w3win, INT_MIN, 2, s);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(w3_band_kernel, dim3((unsigned)(a.v * a.m)), dim3(256), 0, s, a.W3, a.v, a.m, (int *)a.w3win);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    } else {
        b.w3win = nullptr;
    }
    const long long items = ((long long)a.v * a.m + kCompute - 1) / kCompute;
    // 16-byte stores of the specialised sweep: the leading dimension v m is even whenever m is; the base must be too
    const bool even_rows = (a.m % 2 != 0) || (((uintptr_t)a.H & 15) == 0);
#define FAST(M_, R_, I_) \
    if (a.m == M_ && a.r == R_ && items <= I_ && even_rows)                                        \
        return a.f > a.v ? launch_wg_k<M_, R_, I_, true>(b, lds_bytes, s) : launch_wg_k<M_, R_, I_, false>(b, lds_bytes, s);
    FAST(4, 12, 1) FAST(2, 2, 4) FAST(1, 1, 4) FAST(4, 4, 2) FAST(3, 3, 2) FAST(2, 4, 2)
#undef FAST
    return launch_wg_k<0, 0, 1, false>(b, lds_bytes, s);
}

}  // namespace mpcx
