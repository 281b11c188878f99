// Solve with the condensed Hessian for systems too large for LDS (v m from 65 up to 1232 unknowns): blocked
// elimination with partial pivoting, H held in a device workspace (one workgroup per instance) and the right-hand
// sides in LDS (one workgroup per instance and 16 columns), updates on the FP64 matrix cores.
//
// Reference: src/control/controller.rs:93-96 (try_inverse of the Hessian, then H^-1 M^T W4).  Pivot rule and failure
// test are those of an LU with partial pivoting, as nalgebra's: at column k the largest |a[i][k]| among the rows not
// yet used as a pivot; a zero or non-finite pivot marks the instance failed (the reference's "Matrix inversion failed").
//
// Algorithm, per panel of kNB = 16 columns starting at k0:
//   1. the whole panel (all rows, 16 columns) is factored in LDS: for column j the pivot is searched in rows >= k0 + j,
//      exchanged into row k0 + j, and every other row outside the panel's pivot block, the rows above k0 included,
//      gets its multiplier l = a / pivot and is reduced in the remaining panel columns.  Eliminating above as well as
//      below (block Gauss-Jordan) means there is no backward sweep over the matrix.
//   2. a block of 16 other columns (of H, or of right-hand sides) follows the panel: the panel's row exchanges,
//      U12 = L11^-1 T for the 16 pivot rows (the inverse of the panel's unit lower triangle is formed once per panel),
//      and (multipliers) x U12 subtracted from every other 16-row tile on the matrix cores -- evaluated as
//      (multipliers x L11^-1) x T, the first product taken once per panel when it is factored.  Products are computed
//      transposed, so that a lane's four accumulators are four columns of one row.
// What remains is block diagonal with the 16x16 upper triangles U11 of the panels; x_p = U11_p^-1 y_p finishes.
//
// Kernels.  H is factored first, one workgroup per instance, and leaves per panel what a block of columns needs to
// follow it: the multipliers (in place, in the panel's columns), the net row exchanges, L11^-1 and U11^-1.
//   - up to 160 rows: lu_block_kernel, right-looking (after each panel the wavefronts sweep the trailing columns of H
//     in the workspace);
//   - beyond: lu_applyr_kernel<..., FACTOR = true>, left-looking (column block j is loaded, the panels before it are
//     applied with their multipliers streamed from the workspace, then it is factored in LDS and written back): H's
//     columns are read and written once, and the streamed multipliers are half the bytes of the trailing sweeps.
// lu_applyr_kernel<..., false> then gives every block of 16 right-hand-side columns to a workgroup of its own, which
// applies all panels to it, then U11^-1, and writes the result in the caller's layout.  The block is formed from the
// caller's arrays as it is loaded (Gamma^T blockdiag(Q) included), so a right-hand side is read once and its solution
// written once, and never enters the workspace; the workgroups of one instance run on one XCD and read the same
// multipliers at about the same time (L2 hits).
// While panels are applied the block lives in the wavefronts' accumulator registers (only the rows a panel's exchanges
// touch pass through LDS), and the stored multipliers are L L11^-1, so that a panel is applied as
// rows -= (L L11^-1) T straight from the old content T of the pivot rows: one barrier per panel step.
#include <type_traits>
#include <utility>

#include "mpcx_internal.h"

namespace mpcx {
namespace {

using d4 = __attribute__((ext_vector_type(4))) double;

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant in every pass
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int kNB = 16;
constexpr int kSmallRows = 256;           // the most rows lu_block_kernel (right-looking) takes
#ifndef MPCX_LEFT_FROM
#define MPCX_LEFT_FROM 160
#endif
// Systems of more (padded) rows than this are factored left-looking by lu_applyr_kernel<..., true>: measured on 16 384
// instances, solve only, right- against left-looking: 128 rows 7.4 / 7.7 ms, 160 rows 11.8 / 11.9 ms, 208 rows 22.4 / 18.1 ms,
// 240 rows 32.2 / 23.0 ms (the right-looking sweeps move the trailing matrix through memory once per panel).
constexpr int kLeftFrom = MPCX_LEFT_FROM;
constexpr int kRhsMaxRows = 8 * 10 * kNB; // the widest variant of lu_applyr_kernel: 8 wavefronts x 10 row tiles
constexpr int kMaxTouched = 2 * kNB;

struct LuArgs {
    double *ws;      // per instance: H padded to nrp x nrp, column-major
    int *failed;     // per instance
    long long count; // instances in this launch
    int nrp;
    double *meta;    // per instance and panel: L11^-1 and U11^-1 (256 doubles each), then the exchanges as ints (kMetaDoubles in all)
#ifdef MPCX_LU_STAMPS
    double *dbg;     // diagnostic build: per instance the time spent loading panels, factoring them, sweeping the trailing columns
    long long dbg_stride;
#endif
};
constexpr int kMetaInts = 2 * kNB * kNB;                               // where the ints start, in doubles
// a panel's exchanges as ints: tcount, trow[32], tsrc[32], then for a block that keeps the touched rows' old content as a
// list: tsl[16] (where in the list each pivot row's source lies), the same packed four to an int in the order the
// matrix-core operand wants them, and per outside row its table entry (or -1)
constexpr int kTslAt = 1 + 2 * kMaxTouched, kAtpAt = kTslAt + kNB, kEntAt = kAtpAt + 4;
constexpr int kTmetaInts = kEntAt + kNB + 3;                           // (+ 3: what follows stays 8-byte aligned)
constexpr int kMetaDoubles = kMetaInts + kTmetaInts / 2;               // 512 + 52

// ---- pack: workspace <- H padded with an identity block ----
struct PackArgs {
    const double *H;
    double *ws;
    long long first, count, W;  // instances [first, first + count) of the batch
    int nr, nrp;
    int transposed;  // 1: the workspace receives H^T
};

// Lanes run along the rows of a column on both sides.
__global__ __launch_bounds__(256) void lu_pack_kernel(PackArgs p) {
    const long long per = (long long)p.nrp * p.nrp;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.count * per) return;
    const long long li = idx / per, e = idx - li * per;
    const int col = (int)(e / p.nrp), row = (int)(e - (long long)col * p.nrp);
    const long long b = p.first + li;
    const long long src = p.transposed ? (long long)row * p.nr + col : (long long)col * p.nr + row;
    p.ws[idx] = (row < p.nr && col < p.nr) ? p.H[elem_index(b, src, (long long)p.nr * p.nr, p.W)] : (row == col ? 1.0 : 0.0);
}

// The same with H^T for instance-major batches, through 16 x 16 LDS tiles so that both the reads of H (along its
// columns) and the writes of the workspace (along its columns) are 128-byte runs.
__global__ __launch_bounds__(256) void lu_pack_transposed_kernel(PackArgs p) {
    __shared__ double tile[16][17];
    const int tiles = p.nrp / 16;
    const long long li = blockIdx.x / (tiles * tiles);
    const int t = (int)(blockIdx.x - li * tiles * tiles), tr = t / tiles, tc = t - tr * tiles;  // tile of the workspace: rows tr, columns tc
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const double *H = p.H + (p.first + li) * (long long)p.nr * p.nr;
    // workspace (row, col) = H(col, row): read H's column (tr * 16 + ty), rows tc * 16 + tx
    const int hr = tc * 16 + tx, hc = tr * 16 + ty;
    tile[ty][tx] = (hr < p.nr && hc < p.nr) ? H[(long long)hc * p.nr + hr] : (hr == hc ? 1.0 : 0.0);
    __syncthreads();
    const int row = tr * 16 + tx, col = tc * 16 + ty;
    p.ws[(size_t)li * p.nrp * p.nrp + (size_t)col * p.nrp + row] = tile[tx][ty];
}

// ---- factor / eliminate: one workgroup per instance ----
// Panel factorisation of the column block in LDS (blk[c * ld + row], 16 columns x nrp rows, pivot block at row k0) and
// what it leaves behind: the block itself back in the workspace (gcol), L11^-1, U11^-1 and the exchange lists in the
// panel's description (mt).  RPT: rows a thread holds (RPT THREADS >= nrp).  Returns false on a zero or non-finite
// pivot.  Kept out of line: inlined into
// lu_applyr_kernel its registers pushed the multipliers of the panel sweep into scratch memory, whose reloads then
// waited behind the prefetched loads of the next panel.
template <int THREADS, int RPT>
__device__ __noinline__ bool factor_block(double *blk, int ld, int nrp, int k0, double *linv, double *uinv, int *tmeta,
                                          double *red_val, int *red_idx, int *ipiv, double *gcol, double *mt
#ifdef MPCX_LU_STAMPS
                                          , double *dbg
#define FSTAMP(k) do { if (threadIdx.x == 0 && dbg) dbg[k] = (double)wall_clock64(); } while (0)
#ifdef MPCX_LU_STAMPS2  // the inside of column 8 instead of the column times: slots 1..6
#undef FSTAMP
#define FSTAMP(k) do {} while (0)
#define CSTAMP(jc_, k) do { if ((jc_) == 8 && threadIdx.x == 0 && dbg) dbg[1 + (k)] = (double)wall_clock64(); } while (0)
#endif
#else
#define FSTAMP(k) do {} while (0)
#endif
#ifndef CSTAMP
#define CSTAMP(jc_, k) do {} while (0)
#endif
                                          ) {
    constexpr int NW = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int *tcount = tmeta, *trow = tmeta + 1, *tsrc = tmeta + 1 + kMaxTouched, *tsl = tmeta + kTslAt;
    // The panel column by column: pivot searched among the rows at positions >= k0 + jc, every row outside the pivot
    // block reduced.  A thread keeps its rows (RPT of them, 16 entries each) in registers for all 16 columns; a row
    // never moves, its POSITION does (an exchange swaps two positions), and the rows go back to LDS at their final
    // positions.  Per column only the pivot row travels (its owner publishes it, everybody reads it back as a
    // broadcast): two barriers and about 20 LDS instructions per wavefront instead of three barriers and a
    // read-modify-write of every entry.
    double a[RPT][kNB];
    int pos[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int rr = tid + u * THREADS;
        pos[u] = rr < nrp ? rr : -1;  // -1: no row
#pragma unroll
        for (int c = 0; c < kNB; ++c) a[u][c] = blk[c * ld + (rr < nrp ? rr : 0)];
    }
    double *pbuf = linv;  // the pivot row of the current column (L11^-1 is formed after the last column)
    bool failed = false;  // uniform over the workgroup
    FSTAMP(0);
    static_for<kNB>([&](auto JC) __attribute__((always_inline)) {
        constexpr int jc = decltype(JC)::value;  // a constant in every pass: the rows stay in registers
        if (failed) return;
        const int kk = k0 + jc;
        double best = -1.0;
        int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            if (pos[u] >= kk) {
                double x = fabs(a[u][jc]);
                if (!(x == x)) x = INFINITY;  // a NaN must be chosen (and fail), not skipped
                if (x > best || (x == best && pos[u] < bi)) { best = x; bi = pos[u]; }
            }
        }
        CSTAMP(jc, 0);
        // The wavefront's candidate: rows of 16 lanes by lane-to-lane moves (the rule "larger value, then smaller
        // position" is idempotent, so rotations serve as well as exchanges), the four row results through scalar
        // registers.  (Six rounds of cross-lane shuffles took 0.65 us of a column's 2.2.)
        auto take = [&](double ob, int oi) {
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        };
        take(dpp_f64<0xB1>(best), __builtin_amdgcn_mov_dpp(bi, 0xB1, 0xf, 0xf, false));    // lane ^ 1
        take(dpp_f64<0x4E>(best), __builtin_amdgcn_mov_dpp(bi, 0x4E, 0xf, 0xf, false));    // lane ^ 2
        take(dpp_f64<0x124>(best), __builtin_amdgcn_mov_dpp(bi, 0x124, 0xf, 0xf, false));  // rotate by 4 within the row
        take(dpp_f64<0x128>(best), __builtin_amdgcn_mov_dpp(bi, 0x128, 0xf, 0xf, false));  // rotate by 8
        {
            double rb[4];
            int ri[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rb[q] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(best), 16 * q),
                                         __builtin_amdgcn_readlane(__double2loint(best), 16 * q));
                ri[q] = __builtin_amdgcn_readlane(bi, 16 * q);
            }
            best = rb[0];
            bi = ri[0];
#pragma unroll
            for (int q = 1; q < 4; ++q) take(rb[q], ri[q]);
        }
        CSTAMP(jc, 1);
        if (lane == 0) { red_val[wv] = best; red_idx[wv] = bi; }
        __syncthreads();
        CSTAMP(jc, 2);
        {   // every wavefront's candidate read before the first comparison, compared as a tree
            double cb[NW];
            int ci[NW];
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) {
                cb[w2] = red_val[w2];
                ci[w2] = red_idx[w2];
            }
#pragma unroll
            for (int span = 1; span < NW; span *= 2)
#pragma unroll
                for (int w2 = 0; w2 + span < NW; w2 += 2 * span)
                    if (cb[w2 + span] > cb[w2] || (cb[w2 + span] == cb[w2] && ci[w2 + span] < ci[w2])) {
                        cb[w2] = cb[w2 + span];
                        ci[w2] = ci[w2 + span];
                    }
            best = cb[0];
            bi = ci[0];
        }
        if (!(best > 0.0) || !isfinite(best)) { failed = true; return; }
        if (tid == 0) ipiv[jc] = bi;
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            if (pos[u] == bi) {  // the pivot row: published, and moved to position kk
#pragma unroll
                for (int c = jc; c < kNB; ++c) pbuf[c] = a[u][c];
                pos[u] = kk;
            } else if (pos[u] == kk) {
                pos[u] = bi;
            }
        }
        CSTAMP(jc, 3);
        __syncthreads();
        CSTAMP(jc, 4);
        double pv[kNB];
#pragma unroll
        for (int c = jc; c < kNB; ++c) pv[c] = pbuf[c];
        const double inv = 1.0 / pv[jc];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            if (pos[u] < 0 || (pos[u] >= k0 && pos[u] <= kk)) continue;  // the panel's pivot rows keep their entry: it belongs to U11
            const double l = a[u][jc] * inv;
            a[u][jc] = l;
#pragma unroll
            for (int c = jc + 1; c < kNB; ++c) a[u][c] -= l * pv[c];
        }
        CSTAMP(jc, 5);
        FSTAMP(1 + jc);
    });
    if (failed) return false;
    __syncthreads();  // (the last column's readers of pbuf are done: it becomes L11^-1 below)
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        if (pos[u] >= 0) {
#pragma unroll
            for (int c = 0; c < kNB; ++c) blk[c * ld + pos[u]] = a[u][c];
        }
    }
    __syncthreads();
    FSTAMP(17);
    // What the later blocks and the right-hand sides need of this panel, three wavefronts side by side:
    if (tid < kNB) {  // column tid of L11^-1 by forward substitution on a unit vector (every read of the panel issued first)
        double x[kNB];
#pragma unroll
        for (int i = 0; i < kNB; ++i) {
            double sacc = (i == tid) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < kNB; ++q)
                if (q < i) sacc -= blk[q * ld + k0 + i] * x[q];
            x[i] = sacc;
            linv[i * kNB + tid] = sacc;
        }
    }
    if (tid >= 128 && tid < 128 + kNB) {  // column of U11^-1 by backward substitution
        const int jc = tid - 128;
        double x[kNB];
#pragma unroll
        for (int i = kNB - 1; i >= 0; --i) {
            double sacc = (i == jc) ? 1.0 : 0.0;
#pragma unroll
            for (int q = kNB - 1; q >= 0; --q)
                if (q > i) sacc -= blk[q * ld + k0 + i] * x[q];
            x[i] = sacc / blk[i * ld + k0 + i];
            uinv[i * kNB + jc] = x[i];
        }
    }
    if (wv == 1) {
        // Net effect of the 16 exchanges: which rows end up with whose old content.  Lane q < 16 stands for row k0 + q,
        // lanes from 16 on for the rows outside the pivot block in the order the exchanges first meet them; a lane
        // carries the row whose old content its row receives.  One exchange is two lane-to-lane moves, no memory.
        const int myp = ipiv[lane & 15];
        int row = k0 + lane, src = k0 + lane, nt = kNB;
        for (int jc = 0; jc < kNB; ++jc) {
            const int bq = __shfl(myp, jc);  // uniform
            if (bq == k0 + jc) continue;
            int ib;
            if (bq < k0 + kNB) {
                ib = bq - k0;  // (a pivot is never searched above its own column, so bq >= k0 + jc)
            } else {
                const unsigned long long hit = __ballot(lane >= kNB && lane < nt && row == bq);
                if (hit) {
                    ib = __ffsll((long long)hit) - 1;
                } else {
                    ib = nt++;
                    if (lane == ib) { row = bq; src = bq; }
                }
            }
            const int sj = __shfl(src, jc), sb = __shfl(src, ib);
            if (lane == jc) src = sb;
            if (lane == ib) src = sj;
        }
        if (lane < kMaxTouched) { trow[lane] = row; tsrc[lane] = src; }
        if (lane == 0) *tcount = nt;
        // Where a block that keeps the old content of the touched rows as a list (the pivot block's 16 rows, then the rows
        // outside it in the order of this description) finds the source of pivot row `lane`: its index in that list.
        int at = src - k0;
        for (int j = kNB; j < nt; ++j) {
            const int rj = __shfl(row, j);
            if (rj == src) at = j;
        }
        if (lane < kNB) tsl[lane] = at;
        const int packed = __shfl(at, lane & 3) | (__shfl(at, (lane & 3) + 4) << 8) | (__shfl(at, (lane & 3) + 8) << 16) | (__shfl(at, (lane & 3) + 12) << 24);
        if (lane < 4) tmeta[kAtpAt + lane] = packed;  // [k = lane + 4 q] in byte q
        const int osrc = __shfl(src, kNB + (lane & 15));
        if (lane < kNB) tmeta[kEntAt + lane] = kNB + lane < nt ? ((osrc - k0) | (lane << 8)) : -1;  // outside row `lane`: the pivot-block row it receives, its place in the list
    }
    __syncthreads();
    // The multipliers as the later blocks use them.  A row outside the pivot block is reduced by (its multipliers) x U12
    // with U12 = L11^-1 T, T the old content of the pivot rows: the product with L11^-1 is taken here, once per panel, so
    // that a block applies the panel as rows -= (L L11^-1) T with no U12 to wait for.  The pivot block's own place takes
    // -L11^-1: its new rows L11^-1 T are the same product started from zero (U11 lives on in U11^-1 only).
    {
        const int kq = lane >> 4, l15 = lane & 15;
        double la[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) la[q] = linv[(kq + 4 * q) * kNB + l15];  // first operand [c = lane & 15][k] = L11^-1[k][c]
        for (int t = wv; t < nrp / kNB; t += NW) {
            const int r0 = t * kNB;
            if (r0 == k0) {  // uniform over the wavefront
#pragma unroll
                for (int q = 0; q < 4; ++q) blk[(kq + 4 * q) * ld + k0 + l15] = -linv[l15 * kNB + kq + 4 * q];
            } else {
                double lm[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) lm[q] = blk[(kq + 4 * q) * ld + r0 + l15];  // second operand [k][row lane & 15] = L[row][k]
                d4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) d = __builtin_amdgcn_mfma_f64_16x16x4f64(la[q], lm[q], d, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) blk[(kq + 4 * q) * ld + r0 + l15] = d[q];  // [c = kq + 4 q][row lane & 15]
            }
        }
    }
    __syncthreads();
    FSTAMP(18);
    // the block goes back to the workspace (one contiguous run), 16-byte pieces, every LDS read of a round issued first
    {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const int pairs = kNB * nrp / 2;  // nrp is a multiple of 16: a pair never straddles two columns
        for (int e0 = tid; e0 < pairs; e0 += 8 * THREADS) {
            v2d v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * THREADS, ee = e < pairs ? e : 0, c = (2 * ee) / nrp, row2 = 2 * ee - c * nrp;
                v[u] = *(const v2d *)(blk + c * ld + row2);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * THREADS;
                if (e < pairs) *(v2d *)(gcol + 2 * e) = v[u];
            }
        }
    }
    if (tid < kNB * kNB) { mt[tid] = linv[tid]; mt[kNB * kNB + tid] = uinv[tid]; }
    if (tid < kEntAt + kNB) ((int *)(mt + kMetaInts))[tid] = tmeta[tid];
    FSTAMP(19);
    return true;
}

// Right-looking, for systems of up to kSmallRows rows (used up to kLeftFrom): their time goes into the barriers of the
// panel factorization, and a workgroup of 256 threads within 128 registers lets four instances share a CU.  (Larger
// systems are factored left-looking by lu_applyr_kernel<..., true>, which moves half the bytes.)
constexpr int kThreads = 256, kWaves = kThreads / 64;
__global__ __launch_bounds__(kThreads, 4) void lu_block_kernel(LuArgs p) {
    extern __shared__ __align__(16) double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nrp = p.nrp;
    double *panel = sm;                                  // [16][nrp]: panel[j * nrp + row]
    double *linv = panel + (size_t)kNB * nrp;            // [16][16]: linv[i * 16 + k] = (L11^-1)[i][k]
    double *uinv = linv + kNB * kNB;                     // [16][16]: uinv[i * 16 + k] = (U11^-1)[i][k]
    double *tiles = uinv + kNB * kNB;                    // per wavefront [16][16]: T[row * 16 + col]
    double *red_val = tiles + kWaves * kNB * kNB;        // [kWaves]
    int *red_idx = (int *)(red_val + kWaves);            // [kWaves]
    int *ipiv = red_idx + kWaves;                        // [16]
    int *tmeta = ipiv + kNB;                             // tcount, trow[32], tsrc[32] (as in a panel's description)
    int *tcount = tmeta;
    int *trow = tmeta + 1;                               // [32] rows whose content changes with the panel's exchanges
    int *tsrc = trow + kMaxTouched;                      // [32] the row whose old content each of them receives
    double *A = p.ws + (size_t)blockIdx.x * nrp * nrp;
    double *mytile = tiles + wv * kNB * kNB;

#ifdef MPCX_LU_STAMPS
    long long bs_last = wall_clock64();
    double bs_sum[3] = {0.0, 0.0, 0.0};
#define BSTAMP(k) do { const long long now_ = wall_clock64(); bs_sum[k] += (double)(now_ - bs_last); bs_last = now_; } while (0)
#else
#define BSTAMP(k) do {} while (0)
#endif
    for (int k0 = 0; k0 < nrp; k0 += kNB) {
        // 1. panel -> LDS (16 adjacent columns are one contiguous run of the workspace)
        {   // 16-byte pieces, eight loads in flight per thread (one memory latency per round, not one per element)
            typedef double v2d __attribute__((ext_vector_type(2)));
            const v2d *src = (const v2d *)(A + (size_t)k0 * nrp);
            const int pairs = kNB * nrp / 2;
            for (int e0 = tid; e0 < pairs; e0 += 8 * kThreads) {
                v2d v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[e0 + u * kThreads < pairs ? e0 + u * kThreads : 0];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u * kThreads < pairs) ((v2d *)panel)[e0 + u * kThreads] = v[u];
            }
        }
        __syncthreads();
        BSTAMP(0);
        // 2. factored in LDS; the panel goes back to the workspace (U11 is needed at the end) together with L11^-1,
        //    U11^-1 and the net effect of the 16 exchanges (which rows end up with whose old content)
        if (!factor_block<kThreads, kSmallRows / kThreads>(panel, nrp, nrp, k0, linv, uinv, tmeta, red_val, red_idx, ipiv,
                                                           A + (size_t)k0 * nrp,
                                                           p.meta + ((size_t)blockIdx.x * (nrp / kNB) + k0 / kNB) * kMetaDoubles
#ifdef MPCX_LU_STAMPS
                                                           , nullptr
#endif
                                                           )) {
            if (tid == 0) p.failed[blockIdx.x] = 1;  // uniform over the workgroup
            return;
        }
        BSTAMP(1);
        // 3. the columns to the right, one block of 16 per wavefront at a time
        const int ntouched = *tcount;
        d4 la;  // A operand of the triangular solve: (L11^-1)[lane & 15][(lane >> 4) + 4 kk]
#pragma unroll
        for (int q = 0; q < 4; ++q) la[q] = linv[(lane & 15) * kNB + (lane >> 4) + 4 * q];
        for (int c0 = k0 + kNB + wv * kNB; c0 < nrp; c0 += kWaves * kNB) {
            // exchanges: read every touched row's source first, then write (registers in between)
            double hold[kMaxTouched * kNB / 64];
#pragma unroll
            for (int q = 0; q < kMaxTouched * kNB / 64; ++q) {
                const int idx = lane + 64 * q, t = idx >> 4, c = idx & 15;
                hold[q] = A[(size_t)(c0 + c) * nrp + tsrc[t < ntouched ? t : 0]];  // branch-free; surplus reads are dropped
            }
#pragma unroll
            for (int q = 0; q < kMaxTouched * kNB / 64; ++q) {
                const int idx = lane + 64 * q, t = idx >> 4, c = idx & 15;
                if (t < kNB) mytile[t * kNB + c] = hold[q];                       // the 16 pivot rows: to LDS
                else if (t < ntouched) A[(size_t)(c0 + c) * nrp + trow[t]] = hold[q];  // displaced rows: back to the workspace
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // U12 = L11^-1 T: D[row (lane >> 4) + 4 q][col lane & 15]
            d4 u12 = {0.0, 0.0, 0.0, 0.0}, nu;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double tb = mytile[((lane >> 4) + 4 * q) * kNB + (lane & 15)];
                nu[q] = -tb;
                u12 = __builtin_amdgcn_mfma_f64_16x16x4f64(la[q], tb, u12, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) A[(size_t)(c0 + (lane & 15)) * nrp + k0 + (lane >> 4) + 4 * q] = u12[q];
            // every other 16-row tile: C^T -= T^T (L L11^-1)^T.  A operand [i' = column][k] = T[k][column].
            // (the stored multipliers are L L11^-1: the tiles are reduced by (L L11^-1) T, nu = -T)
            const int pidx = k0 / kNB, nt = nrp / kNB - 1;
            double *cbase = A + (size_t)(c0 + (lane >> 4)) * nrp + (lane & 15);  // + r0, + 4 q columns
            auto row0 = [&](int t) { return (t < pidx ? t : t + 1) * kNB; };
            // The next tile is requested before the current one is stored: a wavefront's loads and stores retire through
            // one in-order counter, so a load issued behind a store also waits for that store to complete.
            d4 nxt = {0.0, 0.0, 0.0, 0.0};
            if (nt > 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) nxt[q] = cbase[(size_t)(4 * q) * nrp + row0(0)];
            }
            for (int t = 0; t < nt; ++t) {
                const int r0 = row0(t);
                d4 acc = nxt;
                {   // (unconditional -- after the last tile the same tile again, unused: with a fixed number of loads and
                    // stores per pass the wait for a tile is "all but the four stores issued since", not "everything")
                    const int r1 = row0(t + 1 < nt ? t + 1 : t);
#pragma unroll
                    for (int q = 0; q < 4; ++q) nxt[q] = cbase[(size_t)(4 * q) * nrp + r1];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double lb = panel[(size_t)((lane >> 4) + 4 * q) * nrp + r0 + (lane & 15)];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(nu[q], lb, acc, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) cbase[(size_t)(4 * q) * nrp + r0] = acc[q];
            }
        }
        __syncthreads();  // the next panel's columns were written by other wavefronts of this workgroup
        BSTAMP(2);
    }
#ifdef MPCX_LU_STAMPS
    if (tid == 0 && p.dbg)
        for (int k = 0; k < 3; ++k) p.dbg[(size_t)blockIdx.x * p.dbg_stride + k] = bs_sum[k];
#endif
}

// ---- right-hand sides: one workgroup per instance and block of 16 columns, the block resident in LDS ----
struct RhsArgs {
    double *ws;          // as LuArgs::ws; for the right-hand sides: after the panels, multipliers and U11 in the diagonal blocks
    int *failed;
    double *meta;
    const double *R, *Q; // Q != nullptr: R is Gamma (fr x nr per instance) and the right-hand side is Gamma^T blockdiag(Q)
    double *X;
    int *info;
    long long first, count, W, total;  // total = count x blocks per instance
    int nr, nrhs, r, nrp, negate, ld;
};
constexpr int kMaxWeightBlock = 64;  // widest output weight block r served

// Leading dimension of the LDS block: the four column groups of a tile access (lane >> 4) land 16 doubles apart
// modulo 32, so that the two groups served together use disjoint banks.
int rhs_ld(int nrp) { return (nrp % 32 == 0) ? nrp + 16 : nrp; }

// ---- a block of 16 columns follows the panels: right-hand sides, and H's own column blocks (left-looking) ----
// THREADS: workgroup size (>= 256); MT: row tiles per wavefront (THREADS / 64 * MT * 16 >= nrp).
//
// FACTOR: the same machinery run over H's own column blocks, left-looking, one workgroup per instance: block j is
// loaded, the panels 0..j-1 are applied to it exactly as to a block of right-hand sides, then it is factored in LDS as
// panel j (pivot search, exchanges, multipliers for every row outside the pivot block) and written back with what the
// later blocks need.  H's columns are read and written once; what is streamed per block are the multipliers of the
// panels before it, half the bytes of a right-looking sweep over the trailing columns and reads only.
//
// A block of 16 columns is 2 KB per 16-row tile: a wavefront keeps its tiles in its accumulator registers (lane
// (kq, l15) holds columns kq + 4 q of row l15, the layout the products leave them in) through all the panels, and only
// the rows a panel's exchanges touch travel through LDS -- the 16 rows of the pivot block and the rows outside it that
// trade places with them, 32 at most.  (With the tiles in LDS, read and written back at every step, the steps were
// bound by LDS bandwidth: 2 x 103 KB per step and workgroup at 800 rows.)
//
// Per step: the owner of the pivot block's tile and the owners of touched outside rows put those rows' old content
// into the list old[0..15] / old[16..]; a table row -> (pivot-block row it receives, place in the list) tells a lane
// whether its row is touched, the first wavefront keeps it current one panel ahead.  After ONE barrier every wavefront
// reads T -- the old content of the pivot rows, found in the list through the panel's description -- as the first
// operand of its products: a touched outside row first takes the old content of the pivot-block row the table names,
// then every tile is reduced by (its stored multipliers) x T.  The stored multipliers are L L11^-1 (factor_block), so
// there is no U12 = L11^-1 T to wait for; the pivot block's tile, whose place in the workspace holds -L11^-1, starts
// from zero and becomes U12 by the same product.  The list is double-buffered by the step's parity: a fast wavefront
// may already publish rows of the next step while a slow one still reads this step's.
constexpr int kTS = kNB + 1;  // row stride of a 16 x 16 tile staged in LDS (an odd stride: rows and columns both spread over the banks)

#ifndef MPCX_RHS_WGS
#define MPCX_RHS_WGS 3  // workgroups of 256 threads per CU for the right-hand sides: three leave each thread 168 registers (measured 52.9 against 56.7 ms with four)
#endif
template <int THREADS, int MT, bool FACTOR>
__global__ __launch_bounds__(THREADS, THREADS == 256 ? (FACTOR ? 3 : MPCX_RHS_WGS) : 1) void lu_applyr_kernel(RhsArgs p) {
    extern __shared__ __align__(16) double sm[];
    constexpr int NW = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    const int wr = (wvs + (int)(blockIdx.x >> 3)) & (NW - 1);  // first tile of this wavefront (rotated from workgroup to workgroup)
    const int nrp = p.nrp, ld = p.ld, npan = nrp / kNB, nblk = (p.nrhs + kNB - 1) / kNB;
    const long long per_xcd = (p.total + 7) / 8;
    const long long item = FACTOR ? (long long)blockIdx.x : (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (!FACTOR && ((blockIdx.x >> 3) >= per_xcd || item >= p.total)) return;
    const long long li = FACTOR ? item : item / nblk;
    const int bk = FACTOR ? 0 : (int)(item - li * nblk), c0 = bk * kNB;
    const long long b = p.first + li;
    double *A = p.ws + (size_t)li * nrp * nrp;
    // LDS: with FACTOR what the panel factorisation needs beside its block; then the step's buffers -- which the block
    // being factored overlays (it is in registers while the panels are applied and in LDS while it is factored).
    double *linv = sm;                                           // FACTOR: [16][16]
    double *uinv = linv + kNB * kNB;                             // FACTOR: [16][16]
    int *tmeta = (int *)(uinv + kNB * kNB);                      // FACTOR: a panel's exchange lists as factor_block leaves them
    double *red_val = (double *)(tmeta + kTmetaInts);            // FACTOR: [NW]
    int *red_idx = (int *)(red_val + NW);                        // FACTOR: [NW]
    int *ipiv = red_idx + NW;                                    // FACTOR: [16]
    double *old = FACTOR ? (double *)(ipiv + kNB) : sm;          // [2][32][kTS]
    double *blk = old;                                           // FACTOR: [16][ld], over the step's buffers
    double *xp = old + 2 * kMaxTouched * kTS;                    // right-hand sides: [NW][16][kTS], a wavefront's own tile for transposing; the weight block while the block is formed
    int *table = (int *)(xp + (FACTOR ? 0 : NW * kNB * kTS));    // [2][nrp], by the panel's parity
    const bool bad = !FACTOR && p.failed[li] != 0;
    if (!FACTOR && p.info && bk == 0 && tid == 0) p.info[b] = bad ? 1 : 0;
    const long long ER = (long long)p.nr * p.nrhs;
    if (bad) {  // uniform over the workgroup
        for (int e = tid; e < kNB * p.nr; e += THREADS) {
            const int c = e / p.nr, row = e - c * p.nr;
            if (c0 + c < p.nrhs) p.X[elem_index(b, (long long)(c0 + c) * p.nr + row, ER, p.W)] = __longlong_as_double(0x7ff8000000000000LL);
        }
        return;
    }
    const int kq = lane >> 4, l15 = lane & 15;
    const int lofs = kq * nrp + l15;
    double *meta = p.meta + (size_t)li * npan * kMetaDoubles;
    d4 acc[MT];
    double lb[MT][4];
    int tile0[MT];    // first row of the slot's tile (a slot past the last tile: the last tile, never written)
    bool owned[MT];   // the slot holds a tile
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        const int t0 = wr + NW * u;
        owned[u] = t0 < npan;
        tile0[u] = (t0 < npan ? t0 : npan - 1) * kNB;
    }
    auto fetch_one = [&](int pn, int u, int q) __attribute__((always_inline)) {
        return A[(unsigned)(lofs + (pn * kNB + 4 * q) * nrp + tile0[u])];
    };
    auto fetch_tile = [&](int pn, int u, double(&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = fetch_one(pn, u, q);
    };
    // A panel's exchanges as the first wavefront uses them (every wavefront requests them, unconditionally and one panel
    // ahead like the multipliers, so that the waits stay counted): where in the list the 16 pivot rows' sources lie
    // (operand layout [k = kq + 4 q], a byte each) and, one outside row per lane l15, the row and its table entry.
    int l_atp = 0, l_ent = -1, l_row = 0, at_cur = 0;
    auto fetch_lists = [&](int pn) __attribute__((always_inline)) {
        const int *mi = (const int *)(meta + (size_t)pn * kMetaDoubles + kMetaInts);
        l_atp = mi[kAtpAt + kq];
        l_ent = mi[kEntAt + l15];
        l_row = mi[1 + kNB + l15];
    };
    auto panel_step = [&](int pn, int upto) __attribute__((always_inline)) {
        double *ob = old + (pn & 1) * (kMaxTouched * kTS);
        int *tab = table + (pn & 1) * nrp, *tab_next = table + ((pn + 1) & 1) * nrp;
        // Which of this wavefront's rows the panel touches (the entry is cleared by the lane that read it: the table of
        // this parity is clean again when the first wavefront fills it for the panel after the next); their old content
        // into the list.
        int s[MT];
#pragma unroll
        for (int u = 0; u < MT; ++u) s[u] = tab[tile0[u] + l15];
        // the next panel's touched outside rows into the table of the other parity (lanes 0..15 of the first wavefront,
        // one row each): every look-up of that table's previous use lies before the previous step's barrier
        if (wvs == 0 && lane < kNB && pn + 1 < upto && l_ent >= 0) tab_next[l_row] = l_ent;
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            const bool mine = owned[u] && tile0[u] == pn * kNB;  // uniform over the wavefront
            if (!owned[u]) s[u] = -1;
            const bool any = __builtin_amdgcn_ballot_w64(s[u] >= 0) != 0;
            if (mine || any) {
                const int at = mine ? l15 : kNB + (s[u] >> 8);
                if (mine || s[u] >= 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) ob[at * kTS + kq + 4 * q] = acc[u][q];
                }
                if (s[u] >= 0 && kq == 0) tab[tile0[u] + l15] = -1;
            }
        }
        lds_barrier();
        // first operand of every product of the step: [column l15][k = kq + 4 q] = -T[k][column], T[k] the old content of
        // the row that becomes pivot row k
        d4 nt;
#pragma unroll
        for (int q = 0; q < 4; ++q) nt[q] = -ob[((at_cur >> (8 * q)) & 255) * kTS + l15];
        // touched outside rows take the old content of the pivot-block row the table names (loads under a uniform
        // branch, the choice per lane); the pivot block's tile starts from zero
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            const bool mine = owned[u] && tile0[u] == pn * kNB;
            const bool any = __builtin_amdgcn_ballot_w64(s[u] >= 0) != 0;
            if (mine) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[u][q] = 0.0;
            } else if (any) {
                double nv[4];
                const double *from = ob + ((s[u] >= 0 ? s[u] : 0) & 255) * kTS;
#pragma unroll
                for (int q = 0; q < 4; ++q) nv[q] = from[kq + 4 * q];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[u][q] = s[u] >= 0 ? nv[q] : acc[u][q];
            }
        }
        at_cur = l_atp;
        fetch_lists(pn + 2 < upto ? pn + 2 : upto - 1);
        const int pf = pn + 1 < upto ? pn + 1 : pn;
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            if (owned[u]) {  // uniform over the wavefront
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(nt[q], lb[u][q], acc[u], 0, 0, 0);
            }
            fetch_tile(pf, u, lb[u]);
        }
    };
    // the panels [0, upto) applied to the block in the accumulators
    auto apply_panels = [&](int upto) __attribute__((always_inline)) {
        if (upto <= 0) return;
        // (the requests in the order the steps issue them: the waits are counted against the order of issue, and a different
        // order on the way into the loop makes the first wait of every step cover nearly everything in flight)
        fetch_lists(0);
        for (int e = tid; e < 2 * nrp; e += THREADS) table[e] = -1;
        lds_barrier();
        if (wvs == 0 && lane < kNB && l_ent >= 0) table[l_row] = l_ent;
        at_cur = l_atp;
        fetch_lists(upto > 1 ? 1 : 0);
        lds_barrier();  // the tables (cleared by everyone, the first panel's entries set) before the first look-up
#pragma unroll
        for (int u = 0; u < MT; ++u) fetch_tile(0, u, lb[u]);
        for (int pn = 0; pn < upto; ++pn) panel_step(pn, upto);
    };

    if constexpr (FACTOR) {
        for (int j = 0; j < npan; ++j) {
            const int k0 = j * kNB;
            double *gcol = A + (size_t)k0 * nrp;
#pragma unroll
            for (int u = 0; u < MT; ++u) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[u][q] = gcol[(size_t)(kq + 4 * q) * nrp + tile0[u] + l15];
            }
            apply_panels(j);
            __syncthreads();  // the last step's buffers are read no more: the block takes their place
#pragma unroll
            for (int u = 0; u < MT; ++u) {
                if (owned[u]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) blk[(kq + 4 * q) * ld + tile0[u] + l15] = acc[u][q];
                }
            }
            __syncthreads();
            if (!factor_block<THREADS, (NW * MT * kNB + THREADS - 1) / THREADS>(blk, ld, nrp, k0, linv, uinv, tmeta, red_val, red_idx, ipiv, gcol,
                                       meta + (size_t)j * kMetaDoubles
#ifdef MPCX_LU_STAMPS
                                       , nullptr
#endif
                                       )) {
                if (tid == 0) p.failed[li] = 1;
                return;
            }
            __syncthreads();  // (workgroup scope is all the ordering the next block's reads of this panel need)
        }
        return;
    }

    // The block: 16 columns of the right-hand side, rows beyond nr and columns beyond nrhs zero, straight into the
    // accumulators' layout.
    if (p.Q == nullptr) {
#pragma unroll
        for (int u = 0; u < MT; ++u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = tile0[u] + l15, c = c0 + kq + 4 * q;
                acc[u][q] = (owned[u] && row < p.nr && c < p.nrhs) ? p.R[elem_index(b, (long long)c * p.nr + row, ER, p.W)] : 0.0;
            }
        }
    } else {
        // Gamma^T blockdiag(Q) (controller.rs:96, M^T W4): column c = g r + a2, row k is sum_a Gamma[g r + a][k] Q[a2][a]
        // (Q stored column-major: Q[a2 * r + a]).  The block's 16 columns touch the rows [lo, hi) of Gamma, a contiguous
        // run per unknown k of the column-major (f r) x (v m) array.  Per tile of 16 unknowns the block is the product
        // D[column][unknown] = sum_j S[column][j] Gamma[lo + j][unknown], S the selection of Q's entries (zero where row
        // lo + j lies outside the column's weight block): on the matrix cores, Gamma read straight from memory.
        const int r = p.r, lo = (c0 / r) * r;
        int hi = ((c0 + kNB - 1) / r + 1) * r;
        if (hi > p.nrhs) hi = p.nrhs;
        const int w = hi - lo, ns = (w + 3) / 4;
        const bool q_lds = r * r <= kNB * kNB;  // the weight block rides in LDS (the wavefronts' transposing tiles are idle until the end)
        if (q_lds) {
            for (int e = tid; e < r * r; e += THREADS) xp[e] = p.Q[e];
            __syncthreads();
        }
        const double *qw = q_lds ? xp : p.Q;
        const int cc = c0 + l15, g = cc / r, a2 = cc - g * r, joff = lo - g * r;  // A operand: column lane & 15
        if (ns <= 6) {
            // (the usual case: the 16 columns touch at most 24 rows of Gamma)  Four tiles of a wavefront at a time, every
            // load of the four issued before the first product: one memory latency per four tiles.
            constexpr int TB = 4, NS = 6;
            double qa[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int jj = 4 * u + kq, ja = jj + joff;
                qa[u] = (cc < p.nrhs && jj < w && ja >= 0 && ja < r) ? qw[a2 * r + ja] : 0.0;
            }
            static_for<(MT + TB - 1) / TB>([&](auto rb) {
                constexpr int u0 = decltype(rb)::value * TB, UN = MT - u0 < TB ? MT - u0 : TB;
                double gb[UN][NS];
#pragma unroll
                for (int i = 0; i < UN; ++i) {
                    const int k = tile0[u0 + i] + l15;  // B operand: unknown lane & 15
#pragma unroll
                    for (int u = 0; u < NS; ++u) {
                        const int jj = 4 * u + kq;
                        gb[i][u] = (owned[u0 + i] && k < p.nr && jj < w) ? p.R[elem_index(b, (long long)k * p.nrhs + lo + jj, ER, p.W)] : 0.0;
                    }
                }
#pragma unroll
                for (int i = 0; i < UN; ++i) {
                    d4 a = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int u = 0; u < NS; ++u) a = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[u], gb[i][u], a, 0, 0, 0);  // (zeros past w)
                    acc[u0 + i] = a;
                }
            });
        } else {
            constexpr int SC = 12;  // steps of four rows of Gamma in flight per tile
#pragma unroll
            for (int u = 0; u < MT; ++u) {
                const int k = tile0[u] + l15;  // B operand: unknown lane & 15
                d4 a = {0.0, 0.0, 0.0, 0.0};
                for (int s0 = 0; s0 < ns; s0 += SC) {
                    double gb[SC], qa[SC];
#pragma unroll
                    for (int i = 0; i < SC; ++i) {
                        const int jj = 4 * (s0 + i) + kq, ja = jj + joff;
                        gb[i] = (owned[u] && k < p.nr && jj < w) ? p.R[elem_index(b, (long long)k * p.nrhs + lo + jj, ER, p.W)] : 0.0;
                        qa[i] = (cc < p.nrhs && jj < w && ja >= 0 && ja < r) ? qw[a2 * r + ja] : 0.0;
                    }
#pragma unroll
                    for (int i = 0; i < SC; ++i)
                        if (s0 + i < ns) a = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[i], gb[i], a, 0, 0, 0);
                }
                acc[u] = a;
            }
        }
        __syncthreads();  // the weight block leaves the transposing tiles
    }
    apply_panels(npan);
    // x_p = U11_p^-1 y_p per tile, where it lies: X^T = Y^T U11^-T needs the tile transposed as its first operand -- through
    // the wavefront's own LDS tile (no barrier: a wavefront's LDS operations execute in order).  Every inverse is
    // requested before the first product (the multipliers' registers are free now).
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        const double *ui = meta + (size_t)(tile0[u] / kNB) * kMetaDoubles + kNB * kNB;
#pragma unroll
        for (int q = 0; q < 4; ++q) lb[u][q] = ui[l15 * kNB + kq + 4 * q];  // B operand: [k = kq + 4 q][row l15] = U11^-1[row l15][k]
    }
    double *mine = xp + wvs * (kNB * kTS);
    const double sign = p.negate ? -1.0 : 1.0;
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        if (owned[u]) {
            double ya[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) mine[l15 * kTS + kq + 4 * q] = acc[u][q];       // [row][column]
#pragma unroll
            for (int q = 0; q < 4; ++q) ya[q] = mine[(kq + 4 * q) * kTS + l15];         // A operand: [column l15][k = row kq + 4 q]
            d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[q], lb[u][q], x, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // D[column kq + 4 q][row l15]
                const int row = tile0[u] + l15, c = c0 + kq + 4 * q;
                if (row < p.nr && c < p.nrhs) p.X[elem_index(b, (long long)c * p.nr + row, ER, p.W)] = sign * x[q];
            }
        }
    }
}

size_t applyr_lds_bytes(int nrp, int nw, bool factor) {
    const size_t step = (2 * kMaxTouched * kTS + (factor ? 0 : (size_t)nw * kNB * kTS)) * sizeof(double) + (size_t)2 * nrp * sizeof(int);
    if (!factor) return step;
    const size_t fixed = (2 * kNB * kNB + kTmetaInts / 2 + nw) * sizeof(double) + sizeof(int) * (nw + kNB);
    const size_t block = (size_t)kNB * rhs_ld(nrp) * sizeof(double);
    return fixed + (block > step ? block : step);
}

template <int THREADS, int MT, bool FACTOR>
int launch_apply(const RhsArgs &ra, size_t lds, hipStream_t s) {
    auto kernel = lu_applyr_kernel<THREADS, MT, FACTOR>;
    lds = applyr_lds_bytes(ra.nrp, THREADS / 64, FACTOR);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const long long grid = FACTOR ? ra.count : (ra.total + 7) / 8 * 8;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(THREADS), lds, s, ra);
    return 0;
}


template <bool FACTOR>
int launch_apply_any(const RhsArgs &ra, size_t lds, hipStream_t s) {
    const int npan = ra.nrp / kNB;
    if (npan <= 8) return launch_apply<256, 2, FACTOR>(ra, lds, s);
    if (npan <= 16) return launch_apply<256, 4, FACTOR>(ra, lds, s);
    if (npan <= 32) return launch_apply<512, 4, FACTOR>(ra, lds, s);
    if (npan <= 56) return launch_apply<512, 7, FACTOR>(ra, lds, s);
    return launch_apply<512, 10, FACTOR>(ra, lds, s);
}

size_t rhs_lds_bytes(int nrp) {  // block, U12 / U11^-1, L11^-1, exchange lists, then the panel factorization's reduction slots and pivots
    return ((size_t)kNB * rhs_ld(nrp) + 2 * kNB * kNB + kTmetaInts / 2 + 16) * sizeof(double) + sizeof(int) * (16 + kNB);
}

bool left_looking(int nrp) { return nrp > kLeftFrom; }

size_t lu_lds_bytes(int nrp) {  // lu_block_kernel
    size_t d = (size_t)kNB * nrp + 2 * kNB * kNB + kWaves * kNB * kNB + kWaves;
    return d * sizeof(double) + sizeof(int) * (kWaves + kNB + kTmetaInts);
}

}  // namespace

// Doubles of workspace one instance needs, and the ints for the failure flags ride behind them.
size_t solve_large_ws_doubles(int nr, int nrhs) {
    const size_t nrp = (size_t)((nr + 15) & ~15);
    (void)nrhs;  // right-hand sides never enter the workspace
    return nrp * nrp + 1 + (nrp / kNB) * kMetaDoubles;  // + the instance's failure flag + what each panel leaves for the right-hand sides
}

bool solve_large_supports(int nr) {
    const int nrp = (nr + 15) & ~15;
    return (left_looking(nrp) || lu_lds_bytes(nrp) <= 160 * 1024) && rhs_lds_bytes(nrp) <= 160 * 1024 && nrp <= kRhsMaxRows;
}

// R / Q as in the small kernels (Q != nullptr: R is Gamma and the right-hand side Gamma^T blockdiag(Q)).  `ws` holds
// `ws_doubles` doubles; the batch is processed in as many launches as that takes.
int launch_solve_large(const double *H, const double *R, const double *Q, int r, double *X, int *info, long long batch,
                       long long W, int nr, int nrhs, int negate, double *ws, size_t ws_doubles, hipStream_t s, int transposed) {
    const int nrp = (nr + 15) & ~15;
    const size_t lds_bytes = lu_lds_bytes(nrp), rhs_bytes = rhs_lds_bytes(nrp);
    if (!solve_large_supports(nr)) return MPCX_ERR_UNSUPPORTED;
    if (Q != nullptr && (r < 1 || r > kMaxWeightBlock)) return MPCX_ERR_UNSUPPORTED;  // weight blocks wider than 64 outputs are not served
    const size_t per = solve_large_ws_doubles(nr, nrhs);
    const long long chunk = (long long)(ws_doubles / per);
    if (chunk < 1) return MPCX_ERR_UNSUPPORTED;
    for (long long first = 0; first < batch; first += chunk) {
        const long long count = (batch - first < chunk) ? batch - first : chunk;
        int *failed = (int *)(ws + (size_t)count * nrp * nrp);  // behind the matrices of this launch
        double *meta = ws + (size_t)count * nrp * nrp + (size_t)count;
        hipError_t e = hipMemsetAsync(failed, 0, (size_t)count * sizeof(int), s);
        if (e != hipSuccess) return (int)e;
        const long long pblocks = (count * (long long)nrp * nrp + 255) / 256;
        if (pblocks > 0x7fffffffLL || count > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
        PackArgs pa{H, ws, first, count, W, nr, nrp, transposed};
        const long long tblocks = count * (long long)(nrp / 16) * (nrp / 16);
        if (transposed && W == 1 && tblocks <= 0x7fffffffLL)
            hipLaunchKernelGGL(lu_pack_transposed_kernel, dim3((unsigned)tblocks), dim3(256), 0, s, pa);
        else
            hipLaunchKernelGGL(lu_pack_kernel, dim3((unsigned)pblocks), dim3(256), 0, s, pa);
        LuArgs la{ws, failed, count, nrp, meta};
#ifdef MPCX_LU_STAMPS
        la.dbg = X + (size_t)first * nr * nrhs;
        la.dbg_stride = (long long)nr * nrhs;
#endif
        RhsArgs ra{ws, failed, meta, R, Q, X, info, first, count, W, 0, nr, nrhs, r, nrp, negate, rhs_ld(nrp)};
        if (left_looking(nrp)) {
            const int rc = launch_apply_any<true>(ra, rhs_bytes, s);
            if (rc != 0) return rc;
        } else {
            hipLaunchKernelGGL(lu_block_kernel, dim3((unsigned)count), dim3(kThreads), lds_bytes, s, la);
        }
        const long long total = count * (long long)((nrhs + kNB - 1) / kNB), per_xcd = (total + 7) / 8;
        if (per_xcd * 8 > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
        ra.total = total;
#ifdef MPCX_LU_STAMPS
        continue;  // diagnostic build: the factorisation's time stamps stay in X
#endif
        if (total > 0) {
            const int rc = launch_apply_any<false>(ra, rhs_bytes, s);
            if (rc != 0) return rc;
        }
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // namespace mpcx
