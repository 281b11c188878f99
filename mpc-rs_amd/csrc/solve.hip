// Batched solve with the condensed Hessian: X = +-H^-1 R per instance, by LU with partial pivoting on the
// augmented matrix [H | R] held in LDS, one wavefront per instance.
//
// Reference: src/control/controller.rs:93-96 (`(tmp1 + W3).try_inverse()` then `tmp2 * M^T * W4`: the gain matrix) and
// :130-132 (u_seq = gain * s, first move applied).  Above 4 unknowns nalgebra's try_inverse is an LU with partial
// pivoting that fails on an exactly zero pivot; the same test (plus non-finite pivots) sets the per-instance `info` flag
// here, which is how the batched call reports the reference's Err("Matrix inversion failed") without aborting the
// batch.  Up to 4 unknowns try_inverse evaluates closed forms and fails on a zero determinant: solve_tiny_kernel.
//   solve:  R = fvec (one column),        X = -H^-1 fvec = the optimal input sequence
//   gain:   R = Gamma^T blockdiag(Q),     X = H^-1 Gamma^T W4 = the reference's gain_matrix
#include <type_traits>
#include <utility>

#include "mpcx_internal.h"
#include "small_inverse.h"

namespace mpcx {
constexpr int kSmallMaxRows = 64;  // largest system of the one-wavefront kernels that keep it in registers and LDS (solve_small_kernel)
namespace {

constexpr int kWave = 64;

struct SolveArgs {
    const double *H, *R;
    double *X;
    int *info;
    long long batch, W;
    int nrow, nrhs;
    int negate;
    // gain: R is Gamma ((nrhs) x (nrow) per instance, column-major) and the right-hand side Gamma^T blockdiag(Q) is
    // formed while loading; Q is r x r.  Otherwise Q = nullptr and R is the (nrow x nrhs) right-hand side itself.
    const double *Q;
    int r;
    // 1: the system matrix is H^T (H is read transposed): X = +-H^-T R.  What the first-move rows of the gain need.
    int transposed;
};

// Systems of 1..4 unknowns: one lane per instance, the inverse in closed form (what nalgebra's try_inverse does for
// these sizes: singular exactly when the determinant is zero), then X = +-H^-1 R column by column.
template <int K>
__global__ __launch_bounds__(kWave) void solve_tiny_kernel(SolveArgs p) {
    const long long b = (long long)blockIdx.x * kWave + threadIdx.x;
    if (b >= p.batch) return;
    double a[K][K], inv[K][K];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int i = 0; i < K; ++i) a[i][j] = p.H[elem_index(b, j * K + i, K * K, p.W)];
    const bool ok = inverse_closed_form<K>(a, inv);
    const double sign = p.negate ? -1.0 : 1.0, nan = __longlong_as_double(0x7ff8000000000000LL);
    const long long ER = (long long)K * p.nrhs;
    for (int c = 0; c < p.nrhs; ++c) {
        double rr[K];
        if (p.Q) {  // column c = (block i, output a2) of Gamma^T blockdiag(Q)
            const int i = c / p.r, a2 = c - i * p.r;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                double sum = 0.0;
                for (int q = 0; q < p.r; ++q)
                    sum += p.R[elem_index(b, (long long)k * p.nrhs + i * p.r + q, ER, p.W)] * p.Q[a2 * p.r + q];
                rr[k] = sum;
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) rr[k] = p.R[elem_index(b, (long long)c * K + k, ER, p.W)];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double x = 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) x += (p.transposed ? inv[j][k] : inv[k][j]) * rr[j];  // (H^T)^-1 = (H^-1)^T
            p.X[elem_index(b, (long long)c * K + k, ER, p.W)] = ok ? sign * x : nan;
        }
    }
    if (p.info) p.info[b] = ok ? 0 : 1;
}

template <int K>
int launch_tiny(const SolveArgs &a, hipStream_t s) {
    const long long blocks = (a.batch + kWave - 1) / kWave;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(solve_tiny_kernel<K>, dim3((unsigned)blocks), dim3(kWave), 0, s, a);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(kWave) void solve_kernel(SolveArgs p) {
    extern __shared__ double aug[];  // column-major [col][row], leading dimension ld (odd)
    const int lane = threadIdx.x;
    const long long b = blockIdx.x;
    const int nr = p.nrow, nc = p.nrow + p.nrhs, ld = nr | 1;
    const long long EH = (long long)nr * nr, ER = (long long)nr * p.nrhs;
    for (int e = lane; e < nr * nr; e += kWave)
        aug[p.transposed ? (e % nr) * ld + (e / nr) : (e / nr) * ld + (e % nr)] = p.H[elem_index(b, e, EH, p.W)];
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

// Systems of up to 64 unknowns: G = 16, 32 or 64 lanes per instance, lane i owns row i of [H | R], held row-major in LDS
// (so a lane's sweep along its row and the broadcast of the pivot row are 16-byte LDS accesses).  Gauss-Jordan with
// partial pivoting and implicit row exchanges: at step k the pivot is the largest |a[i][k]| among the rows that have not
// been a pivot yet (the same set nalgebra's LU searches; on an exact tie the lowest physical row wins, where nalgebra
// takes the first row in exchanged order), every other row is reduced, and no back substitution is needed.  A group
// never spans wavefronts and LDS operations of a wavefront execute in order, so the steps need no workgroup barrier.
// A workgroup serves `ipb` consecutive instances and moves them between HBM and LDS cooperatively, so that in the
// batch-interleaved layouts one access covers the same element of all its instances.
template <int G>
__global__ __launch_bounds__(1024) void solve_small_kernel(SolveArgs p, int ipb, int ldr) {
    extern __shared__ __align__(16) double rows[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int nr = p.nrow, nrhs = p.nrhs, nc = nr + nrhs;
    const long long b0 = (long long)blockIdx.x * ipb;
    const int live = (p.batch - b0 < ipb) ? (int)(p.batch - b0) : ipb;
    const size_t per_inst = (size_t)nr * ldr;
    const int EH = nr * nr, ER = nr * nrhs;
    // ---- HBM -> LDS (transposing: element e = col * nr + row of a column-major matrix goes to row * ldr + col)
    auto load = [&](const double *src, int E, int col0, bool tr) {  // tr: the square matrix goes in transposed
        if (p.W > 1) {  // instance fastest: one access = the same element of all instances of the block
            const int step = nthreads / ipb, inst = tid % ipb;
            int e = tid / ipb, col = e / nr, row = e - col * nr;
            const int dcol = step / nr, drow = step - dcol * nr;
            const double *sp = src + elem_index(b0 + (inst < live ? inst : 0), 0, E, p.W);  // element e is e * W further
            for (; e < E; e += step) {
                if (inst < live)
                    rows[inst * per_inst + (tr ? (size_t)col * ldr + row : (size_t)row * ldr + col0 + col)] = sp[(long long)e * p.W];
                col += dcol;
                row += drow;
                if (row >= nr) { row -= nr; ++col; }
            }
        } else {  // instance-major: one instance at a time, elements fastest
            const int dcol = nthreads / nr, drow = nthreads - dcol * nr;
            for (int inst = 0; inst < live; ++inst) {
                int e = tid, col = e / nr, row = e - col * nr;
                const double *sp = src + (b0 + inst) * (long long)E;
                for (; e < E; e += nthreads) {
                    rows[inst * per_inst + (tr ? (size_t)col * ldr + row : (size_t)row * ldr + col0 + col)] = sp[e];
                    col += dcol;
                    row += drow;
                    if (row >= nr) { row -= nr; ++col; }
                }
            }
        }
    };
    load(p.H, EH, 0, p.transposed != 0);
    if (p.Q == nullptr) {
        load(p.R, ER, nr, false);
    } else {
        // right-hand side column c = (blk, a2), row k:  sum_a Gamma[blk r + a][k] Q[a][a2]   (controller.rs:96, M^T W4)
        const int r = p.r, fr = nrhs;
        if (p.W > 1) {
            const int step = nthreads / ipb, inst = tid % ipb;
            const double *sp = p.R + elem_index(b0 + (inst < live ? inst : 0), 0, ER, p.W);
            int e = tid / ipb, k = e / fr, c = e - k * fr;  // e = k * fr + c, as Gamma is stored
            const int dk = step / fr, dc = step - dk * fr;
            for (; e < ER; e += step) {
                if (inst < live) {
                    const int blk = c / r, a2 = c - blk * r;
                    double acc = 0.0;
                    for (int a = 0; a < r; ++a) acc += sp[(long long)(k * fr + blk * r + a) * p.W] * p.Q[a2 * r + a];
                    rows[inst * per_inst + (size_t)k * ldr + nr + c] = acc;
                }
                k += dk;
                c += dc;
                if (c >= fr) { c -= fr; ++k; }
            }
        } else {
            const int dk = nthreads / fr, dc = nthreads - dk * fr;
            for (int inst = 0; inst < live; ++inst) {
                const double *sp = p.R + (b0 + inst) * (long long)ER;
                int e = tid, k = e / fr, c = e - k * fr;
                for (; e < ER; e += nthreads) {
                    const int blk = c / r, a2 = c - blk * r;
                    double acc = 0.0;
                    for (int a = 0; a < r; ++a) acc += sp[k * fr + blk * r + a] * p.Q[a2 * r + a];
                    rows[inst * per_inst + (size_t)k * ldr + nr + c] = acc;
                    k += dk;
                    c += dc;
                    if (c >= fr) { c -= fr; ++k; }
                }
            }
        }
    }
    lds_barrier();

    const int g = tid / G, i = tid % G;          // instance within the block, row
    const bool active = g < live && i < nr;
    double *inst_a = rows + (size_t)(g < ipb ? g : 0) * per_inst;
    double *arow = inst_a + (size_t)(i < nr ? i : 0) * ldr;
    const int lane = tid & 63, gbase = lane - i;  // first lane of this group within the wavefront
    bool done = false, failed = false;
    int mycol = 0;
    for (int k = 0; k < nr; ++k) {
        double cand = -1.0;
        if (active && !done) {
            const double a = fabs(arow[k]);
            cand = (a == a) ? a : INFINITY;  // a NaN must be picked (and then fail), not skipped
        }
        double best = cand;
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) best = fmax(best, __shfl_xor(best, off));
        const unsigned long long votes = __ballot(cand == best && active && !done);
        const unsigned long long mine = (G == 64) ? votes : ((votes >> gbase) & ((1ull << G) - 1ull));
        const int prow = mine ? __ffsll((long long)mine) - 1 : 0;
        const double *pr = inst_a + (size_t)prow * ldr;
        const double piv = pr[k];
        if (!(best > 0.0) || !isfinite(best)) failed = true;  // zero or non-finite pivot: the reference's inversion error
        if (i == prow && !failed) { done = true; mycol = k; }
        const double m = (active && i != prow && !failed) ? arow[k] / piv : 0.0;
        // a[i][j] -= m a[prow][j] for j > k, two columns per access; an odd start also rewrites column k (never read again)
        int j = (k + 1) & ~1;
        for (; j + 8 <= nc; j += 8) {
            const double2 p0 = *(const double2 *)(pr + j), p1 = *(const double2 *)(pr + j + 2);
            const double2 p2 = *(const double2 *)(pr + j + 4), p3 = *(const double2 *)(pr + j + 6);
            double2 a0 = *(double2 *)(arow + j), a1 = *(double2 *)(arow + j + 2);
            double2 a2 = *(double2 *)(arow + j + 4), a3 = *(double2 *)(arow + j + 6);
            a0.x -= m * p0.x; a0.y -= m * p0.y; a1.x -= m * p1.x; a1.y -= m * p1.y;
            a2.x -= m * p2.x; a2.y -= m * p2.y; a3.x -= m * p3.x; a3.y -= m * p3.y;
            if (active) {
                *(double2 *)(arow + j) = a0; *(double2 *)(arow + j + 2) = a1;
                *(double2 *)(arow + j + 4) = a2; *(double2 *)(arow + j + 6) = a3;
            }
        }
        for (; j < nc; j += 2) {  // ldr is even and >= nc + 1: the pair never leaves the row
            const double2 p0 = *(const double2 *)(pr + j);
            double2 a0 = *(double2 *)(arow + j);
            a0.x -= m * p0.x; a0.y -= m * p0.y;
            if (active) *(double2 *)(arow + j) = a0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // the row that was the pivot of step k holds unknown k: x[k][c] = a[row][nr + c] / a[row][k]
    if (active) {
        const double sign = p.negate ? -1.0 : 1.0;
        const double inv = 1.0 / arow[mycol];
        const long long b = b0 + g;
        double *xp = p.X + elem_index(b, failed ? i : mycol, ER, p.W);
        const long long xstep = (long long)nr * p.W;
        for (int c = 0; c < nrhs; ++c) {
            const double x = failed ? __longlong_as_double(0x7ff8000000000000LL) : sign * (arow[nr + c] * inv);
            xp[c * xstep] = x;
        }
        if (p.info && i == 0) p.info[b] = failed ? 1 : 0;
    }
}

// The same elimination for the reference's own horizon (v m = 20 unknowns; 20 right-hand sides for the gain of a
// single-output plant, one for the solve) with every size a compile-time constant: lane i of a 32-lane group keeps row
// i of [H | R] in REGISTERS (all loops unrolled, static register indices), and the only LDS traffic of a step is the
// pivot row: the lane that owns it publishes its live columns, every lane of the group reads them back as a
// broadcast.  That is 8 bytes of LDS traffic per updated element where solve_small_kernel moves 24 (it reads the pivot
// element, reads its own and writes its own), and LDS bandwidth is what bounds that kernel.  HBM <-> registers goes
// through an LDS staging area [element][instance] that the whole workgroup fills / drains with instance-fastest
// accesses (one access = the same element of all `ipb` instances of the workgroup in the interleaved layouts).
// Arithmetic, pivot choice and failure handling are those of solve_small_kernel, operation for operation.
// RR > 0: R is Gamma (NRHS x NR per instance, NRHS = f RR) and the right-hand side Gamma^T blockdiag(Q) is formed while
// the rows are pulled from the staging area; RR = 0: R is the NR x NRHS right-hand side itself.
template <class F, int... K>
__device__ __forceinline__ void for_each_index(F &f, std::integer_sequence<int, K...>) {
    (f(std::integral_constant<int, K>{}), ...);
}

constexpr int kRowsIpb = 8;  // instances per workgroup of solve_rows_kernel (256 threads: 32 lanes per instance)

template <int NR, int NRHS, int RR>
__global__ __launch_bounds__(256, 4) void solve_rows_kernel(SolveArgs p, int ipb) {
    constexpr int G = 32, NC = NR + NRHS, NCP = (NC + 1) & ~1;
    constexpr int EH = NR * NR, ER = NR * NRHS, ES = EH > ER ? EH : ER;
    static_assert(NR <= G, "one row per lane of a 32-lane group");
    extern __shared__ __align__(16) double sm[];
    const int tid = threadIdx.x;
    const int ldi = ipb | 1;                 // odd leading dimension of the staging area: [element][instance]
    double *stage = sm;
    double *pbuf = sm + (((size_t)ES * ldi + 1) & ~(size_t)1);  // pivot rows: [group][NCP]
    // Two workgroups of 8 instances share every 128-byte line of an interleaved array, and consecutive workgroups go to
    // different XCDs (different L2s): the two that share lines are made the ones 8 apart in launch order, neighbours on
    // one XCD, so that the second finds the line in that XCD's L2.
#ifndef MPCX_ROWS_NO_PAIRING
    const long long slot = blockIdx.x >> 3, lblk = (((slot >> 1) << 3) + (blockIdx.x & 7)) * 2 + (slot & 1);
#else
    const long long lblk = blockIdx.x;
#endif
    const long long b0 = lblk * ipb;
    if (b0 >= p.batch) return;  // (the grid is rounded up to whole groups of 16 workgroups)
    const int live = (p.batch - b0 < ipb) ? (int)(p.batch - b0) : ipb;

    // tr: the square matrix is staged transposed (element (row, col) lands where (col, row) would), so that the lanes'
    // reads of their rows below are the same conflict-free pattern either way
    // All of a thread's loads go out before the first is used (a loop of load - store to LDS - load waits a memory round
    // trip per element, a dozen in a row here); a slot beyond the array reads a valid address and is dropped.
    auto load = [&](const double *src, auto Ec, bool tr) {
        constexpr int E = decltype(Ec)::value;
        if (p.W > 1) {  // instance fastest
            constexpr int STEP = 256 / kRowsIpb, IT = (E + STEP - 1) / STEP;
            const int inst = tid % kRowsIpb, e0 = tid / kRowsIpb;
            const double *sp = src + elem_index(b0 + (inst < live ? inst : 0), 0, E, p.W);
            double tmp[IT];
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int e = e0 + k * STEP;
                tmp[k] = sp[(long long)(e < E ? e : E - 1) * p.W];
            }
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int e = e0 + k * STEP;
                const int es = tr ? (e % NR) * NR + e / NR : e;
                if (inst < live && e < E) stage[es * ldi + inst] = tmp[k];
            }
        } else {        // instance-major: elements fastest
            constexpr int IT = (kRowsIpb * E + 255) / 256;
            double tmp[IT];
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int t = tid + k * 256, tc = t < live * E ? t : 0;
                tmp[k] = src[b0 * (long long)E + tc];  // (instance inst, element e of it: t = inst E + e)
            }
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int t = tid + k * 256, inst = t / E, e = t - inst * E;
                const int es = tr ? (e % NR) * NR + e / NR : e;
                if (t < live * E) stage[es * ldi + inst] = tmp[k];
            }
        }
    };
    const int g = tid / G, i = tid % G;
    const bool active = g < live && i < NR;
    const int gi = g < ipb ? g : 0, ii = i < NR ? i : 0;
    double a[NC];

    load(p.H, std::integral_constant<int, EH>{}, p.transposed != 0);
    lds_barrier();
#pragma unroll
    for (int j = 0; j < NR; ++j) a[j] = stage[(j * NR + ii) * ldi + gi];
    lds_barrier();
    load(p.R, std::integral_constant<int, ER>{}, false);
    lds_barrier();
    if constexpr (RR == 0) {
#pragma unroll
        for (int c = 0; c < NRHS; ++c) a[NR + c] = stage[(c * NR + ii) * ldi + gi];
    } else {
        // right-hand side column c = (blk, a2), row i:  sum_a Gamma[blk RR + a][i] Q[a][a2]   (controller.rs:96, M^T W4)
        double q[RR > 0 ? RR : 1][RR > 0 ? RR : 1];
#pragma unroll
        for (int a2 = 0; a2 < RR; ++a2)
#pragma unroll
            for (int aa = 0; aa < RR; ++aa) q[aa][a2] = p.Q[a2 * RR + aa];
#pragma unroll
        for (int c = 0; c < NRHS; ++c) {
            const int blk = c / (RR > 0 ? RR : 1), a2 = c % (RR > 0 ? RR : 1);
            double acc = 0.0;
#pragma unroll
            for (int aa = 0; aa < RR; ++aa) acc += stage[(ii * NRHS + blk * RR + aa) * ldi + gi] * q[aa][a2];
            a[NR + c] = acc;
        }
    }
    lds_barrier();  // the staging area is free again (the results go through it)

    const int lane = tid & 63, gbase = lane - i;  // first lane of this group within the wavefront
    double *prow_buf = pbuf + (size_t)gi * NCP;
    bool done = false, failed = false;
    int mycol = 0;
    double mypiv = 1.0;
    // One elimination step with the step index a compile-time constant, so that every inner loop has constant bounds
    // (full unrolling, static register indices: the row never leaves the registers).
    auto step = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        double cand = -1.0;
        if (active && !done) {
            const double x = fabs(a[k]);
            cand = (x == x) ? x : INFINITY;  // a NaN must be picked (and then fail), not skipped
        }
        // maximum over the group: four exchanges inside rows of 16 lanes at ALU speed, one across the two rows
        double best = cand;
        best = fmax(best, dpp_f64<0xB1>(best));   // quad_perm [1,0,3,2]
        best = fmax(best, dpp_f64<0x4E>(best));   // quad_perm [2,3,0,1]
        best = fmax(best, dpp_f64<0x141>(best));  // row_half_mirror: quads pair up
        best = fmax(best, dpp_f64<0x140>(best));  // row_mirror: halves of a row pair up
        best = fmax(best, __shfl_xor(best, 16));
        const unsigned long long votes = __ballot(cand == best && active && !done);
        const unsigned mine = (unsigned)(votes >> gbase);
        const int prow = mine ? __ffs((int)mine) - 1 : 0;
        if (!(best > 0.0) || !isfinite(best)) failed = true;  // zero or non-finite pivot: the reference's inversion error
        const bool is_piv = (i == prow);
        constexpr int J0 = k & ~1;  // 16-byte pairs: an odd k republishes column k - 1 as well (never read)
        if (is_piv) {
#pragma unroll
            for (int j = J0; j < NC; j += 2)
                *(double2 *)(prow_buf + j) = make_double2(a[j], j + 1 < NC ? a[j + 1 < NC ? j + 1 : j] : 0.0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double piv = prow_buf[k];
        if (is_piv && !failed) { done = true; mycol = k; mypiv = a[k]; }
        const double m = (active && !is_piv && !failed) ? a[k] / piv : 0.0;
        // a[j] -= m P[j] for j > k, the pivot row read back in chunks of CH pairs (the whole row at once would need
        // as many registers again as the lane's own row)
        constexpr int CH = 5;
#pragma unroll
        for (int j0 = (k + 1) & ~1; j0 < NC; j0 += 2 * CH) {
            double2 t[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (j0 + 2 * u < NC) t[u] = *(const double2 *)(prow_buf + j0 + 2 * u);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = j0 + 2 * u;
                if (j > k && j < NC) a[j < NC ? j : 0] -= m * t[u].x;
                if (j + 1 > k && j + 1 < NC) a[j + 1 < NC ? j + 1 : 0] -= m * t[u].y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for_each_index(step, std::make_integer_sequence<int, NR>{});
    // the row that was the pivot of step k holds unknown k: x[k][c] = a[row][NR + c] / a[row][k]
    if (active) {
        const double sign = p.negate ? -1.0 : 1.0;
        const double inv = 1.0 / mypiv;
        const int row = failed ? i : mycol;
#pragma unroll
        for (int c = 0; c < NRHS; ++c)
            stage[(c * NR + row) * ldi + g] = failed ? __longlong_as_double(0x7ff8000000000000LL) : sign * (a[NR + c] * inv);
        if (p.info && i == 0) p.info[b0 + g] = failed ? 1 : 0;
    }
    lds_barrier();
    // (the staged results are read first and stored back to back)
    if (p.W > 1) {
        constexpr int STEP = 256 / kRowsIpb, IT = (ER + STEP - 1) / STEP;
        const int inst = tid % kRowsIpb, e0 = tid / kRowsIpb;
        double *xp = p.X + elem_index(b0 + (inst < live ? inst : 0), 0, ER, p.W);
        double tmp[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int e = e0 + k * STEP;
            tmp[k] = stage[(e < ER ? e : ER - 1) * ldi + inst];
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int e = e0 + k * STEP;
            if (inst < live && e < ER) xp[(long long)e * p.W] = tmp[k];
        }
    } else {
        constexpr int IT = (kRowsIpb * ER + 255) / 256;
        double tmp[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int t = tid + k * 256, tc = t < live * ER ? t : 0, inst = tc / ER, e = tc - inst * ER;
            tmp[k] = stage[e * ldi + inst];
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int t = tid + k * 256;
            if (t < live * ER) p.X[b0 * (long long)ER + t] = tmp[k];
        }
    }
}

// The same with TWO rows per lane and 16-lane groups (16 < NR <= 32: lane i keeps rows i and i + 16, the latter only
// where it exists), four instances per wavefront instead of two.  What a step costs beside its multiply-adds -- the
// pivot search, the vote, publishing and reading back the pivot row, the reciprocal -- is paid once per wavefront
// whatever the number of rows a lane holds, the search stays inside one 16-lane row of the wavefront (four lane-to-lane
// exchanges, no cross-row step), and a pivot row read back from LDS serves two rows.  The multipliers are taken as
// a[k] x (1 / pivot), as nalgebra's elimination step scales its column.  A workgroup of 256 threads serves 16
// instances (one access of the cooperative loads = one 128-byte line of an interleaved array) and stages the arrays in
// pieces of CW columns.  RR: 0 (R is the right-hand side) or 1 (R is Gamma of a single-output plant, scaled by Q).
template <int NR, int NRHS, int RR>
__global__ __launch_bounds__(256) void solve_rows2_kernel(SolveArgs p) {
    constexpr int G = 16, IPB = 16, LDI = IPB + 1, NC = NR + NRHS, NCP = (NC + 1) & ~1;
    constexpr int CW = 5, CE = CW * NR;  // columns and elements staged at a time
    static_assert(NR > G && NR <= 2 * G, "rows i and i + 16 in lane i");
    static_assert(RR == 0 || RR == 1, "a right-hand side, or Gamma of a single-output plant");
    extern __shared__ __align__(16) double sm[];
    double *stage = sm;                                    // [CE][LDI]: [element of the piece][instance]
    double *pbuf = sm + (((size_t)CE * LDI + 1) & ~(size_t)1);  // pivot rows: [group][NCP]
    const int tid = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * IPB;
    const int live = (p.batch - b0 < IPB) ? (int)(p.batch - b0) : IPB;
    const int g = tid / G, i = tid % G, i1 = i + G < NR ? i + G : i;
    const bool act0 = g < live, act1 = g < live && i + G < NR;
    constexpr long long EH = (long long)NR * NR, ER = (long long)NR * NRHS;

    // `count` elements of an array (E per instance) into the staging area, element `elem_of(idx)` at slot idx; all of a
    // thread's loads go out before the first is used (a slot beyond the piece reads element 0 and is dropped)
    auto load_piece = [&](const double *src, long long E, auto count_c, auto elem_of) {
        constexpr int count = decltype(count_c)::value;
        if (p.W > 1) {  // instance fastest: one access = the same element of the workgroup's 16 instances
            constexpr int STEP = 256 / IPB, IT = (count + STEP - 1) / STEP;
            const int inst = tid % IPB, i0 = tid / IPB;
            const double *sp = src + elem_index(b0 + (inst < live ? inst : 0), 0, E, p.W);
            double tmp[IT];
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int idx = i0 + k * STEP;
                tmp[k] = sp[(long long)elem_of(idx < count ? idx : 0) * p.W];
            }
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int idx = i0 + k * STEP;
                if (inst < live && idx < count) stage[idx * LDI + inst] = tmp[k];
            }
        } else {
            constexpr int IT = (IPB * count + 255) / 256;
            double tmp[IT];
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int t = tid + k * 256, tc = t < live * count ? t : 0, inst = tc / count, idx = tc - inst * count;
                tmp[k] = src[(b0 + inst) * E + elem_of(idx)];
            }
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int t = tid + k * 256, inst = t / count, idx = t - inst * count;
                if (t < live * count) stage[idx * LDI + inst] = tmp[k];
            }
        }
    };
    double a0[NC], a1[NC];
    // H, CW columns at a time (transposed: the columns of H^T are the rows of H)
    const bool tr = p.transposed != 0;
    auto load_h = [&](auto cc) {
        constexpr int jc = decltype(cc)::value * CW, cw = NR - jc < CW ? NR - jc : CW;
        load_piece(p.H, EH, std::integral_constant<int, cw * NR>{}, [&](int idx) { const int j = jc + idx / NR, row = idx % NR; return tr ? row * NR + j : j * NR + row; });
        lds_barrier();
#pragma unroll
        for (int jj = 0; jj < cw; ++jj) {
            a0[jc + jj] = stage[(jj * NR + i) * LDI + g];
            a1[jc + jj] = stage[(jj * NR + i1) * LDI + g];
        }
        lds_barrier();
    };
    for_each_index(load_h, std::make_integer_sequence<int, (NR + CW - 1) / CW>{});
    const double q = RR ? p.Q[0] : 1.0;
    auto load_r = [&](auto cc) {
        constexpr int c0 = decltype(cc)::value * CW, cw = NRHS - c0 < CW ? NRHS - c0 : CW;
        // right-hand side column c, row `row`: R[c NR + row], or Gamma[c][row] = R[row NRHS + c] (controller.rs:96, M^T W4)
        load_piece(p.R, ER, std::integral_constant<int, cw * NR>{}, [&](int idx) { const int c = c0 + idx / NR, row = idx % NR; return RR ? row * NRHS + c : c * NR + row; });
        lds_barrier();
#pragma unroll
        for (int jj = 0; jj < cw; ++jj) {
            const double r0 = stage[(jj * NR + i) * LDI + g], r1 = stage[(jj * NR + i1) * LDI + g];
            a0[NR + c0 + jj] = RR ? 0.0 + r0 * q : r0;
            a1[NR + c0 + jj] = RR ? 0.0 + r1 * q : r1;
        }
        lds_barrier();
    };
    for_each_index(load_r, std::make_integer_sequence<int, (NRHS + CW - 1) / CW>{});

    const int lane = tid & 63, gbase = lane - i;  // first lane of this group within the wavefront
    double *prow_buf = pbuf + (size_t)g * NCP;
    bool done0 = false, done1 = false, failed = false;
    int col0 = 0, col1 = 0;
    double piv0 = 1.0, piv1 = 1.0;
    auto step = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        double cand0 = -1.0, cand1 = -1.0;
        if (act0 && !done0) {
            const double x = fabs(a0[k]);
            cand0 = (x == x) ? x : INFINITY;  // a NaN must be picked (and then fail), not skipped
        }
        if (act1 && !done1) {
            const double x = fabs(a1[k]);
            cand1 = (x == x) ? x : INFINITY;
        }
        double best = fmax(cand0, cand1);
        best = fmax(best, dpp_f64<0xB1>(best));   // quad_perm [1,0,3,2]
        best = fmax(best, dpp_f64<0x4E>(best));   // quad_perm [2,3,0,1]
        best = fmax(best, dpp_f64<0x141>(best));  // row_half_mirror: quads pair up
        best = fmax(best, dpp_f64<0x140>(best));  // row_mirror: halves of a row pair up
        const unsigned v0 = (unsigned)(__ballot(cand0 == best && act0 && !done0) >> gbase) & 0xffffu;
        const unsigned v1 = (unsigned)(__ballot(cand1 == best && act1 && !done1) >> gbase) & 0xffffu;
        const int prow = v0 ? __ffs((int)v0) - 1 : v1 ? G + __ffs((int)v1) - 1 : 0;  // ties: the lowest row
        if (!(best > 0.0) || !isfinite(best)) failed = true;  // zero or non-finite pivot: the reference's inversion error
        const bool is_piv0 = (i == prow), is_piv1 = (i + G == prow);
        constexpr int J0 = k & ~1;  // 16-byte pairs: an odd k republishes column k - 1 as well (never read)
        if (is_piv0) {
#pragma unroll
            for (int j = J0; j < NC; j += 2)
                *(double2 *)(prow_buf + j) = make_double2(a0[j], j + 1 < NC ? a0[j + 1 < NC ? j + 1 : j] : 0.0);
        } else if (is_piv1) {
#pragma unroll
            for (int j = J0; j < NC; j += 2)
                *(double2 *)(prow_buf + j) = make_double2(a1[j], j + 1 < NC ? a1[j + 1 < NC ? j + 1 : j] : 0.0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double inv = 1.0 / prow_buf[k];
        if (is_piv0 && !failed) { done0 = true; col0 = k; piv0 = a0[k]; }
        if (is_piv1 && !failed) { done1 = true; col1 = k; piv1 = a1[k]; }
        const double m0 = (act0 && !is_piv0 && !failed) ? a0[k] * inv : 0.0;
        const double m1 = (act1 && !is_piv1 && !failed) ? a1[k] * inv : 0.0;
        // a[j] -= m P[j] for j > k, the pivot row read back in chunks of CH pairs
        constexpr int CH = 5;
#pragma unroll
        for (int j0 = (k + 1) & ~1; j0 < NC; j0 += 2 * CH) {
            double2 t[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (j0 + 2 * u < NC) t[u] = *(const double2 *)(prow_buf + j0 + 2 * u);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = j0 + 2 * u;
                if (j > k && j < NC) {
                    a0[j < NC ? j : 0] -= m0 * t[u].x;
                    a1[j < NC ? j : 0] -= m1 * t[u].x;
                }
                if (j + 1 > k && j + 1 < NC) {
                    a0[j + 1 < NC ? j + 1 : 0] -= m0 * t[u].y;
                    a1[j + 1 < NC ? j + 1 : 0] -= m1 * t[u].y;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for_each_index(step, std::make_integer_sequence<int, NR>{});
    // the row that was the pivot of step k holds unknown k: x[k][c] = a[row][NR + c] / a[row][k]
    const double sign = p.negate ? -1.0 : 1.0;
    const double inv0 = 1.0 / piv0, inv1 = 1.0 / piv1;
    const int row0 = failed ? i : col0, row1 = failed ? i + G : col1;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    if (p.info && i == 0 && act0) p.info[b0 + g] = failed ? 1 : 0;
    lds_barrier();
    auto store_x = [&](auto cc) {
        constexpr int c0 = decltype(cc)::value * CW, cw = NRHS - c0 < CW ? NRHS - c0 : CW;
#pragma unroll
        for (int jj = 0; jj < cw; ++jj) {
            if (act0) stage[(jj * NR + row0) * LDI + g] = failed ? nan : sign * (a0[NR + c0 + jj] * inv0);
            if (act1) stage[(jj * NR + row1) * LDI + g] = failed ? nan : sign * (a1[NR + c0 + jj] * inv1);
        }
        lds_barrier();
        constexpr int count = cw * NR, e0 = c0 * NR;  // X is NR x NRHS, column-major: the piece is one contiguous run of elements
        if (p.W > 1) {
            const int inst = tid % IPB;
            if (inst < live) {
                double *xp = p.X + elem_index(b0 + inst, 0, ER, p.W);
                for (int idx = tid / IPB; idx < count; idx += 256 / IPB) xp[(long long)(e0 + idx) * p.W] = stage[idx * LDI + inst];
            }
        } else {
            for (int t = tid; t < live * count; t += 256) {
                const int inst = t / count, idx = t - inst * count;
                p.X[(b0 + inst) * ER + e0 + idx] = stage[idx * LDI + inst];
            }
        }
        lds_barrier();
    };
    for_each_index(store_x, std::make_integer_sequence<int, (NRHS + CW - 1) / CW>{});
}

template <int NR, int NRHS, int RR>
int launch_rows2(const SolveArgs &a, hipStream_t s) {
    constexpr int NC = NR + NRHS, NCP = (NC + 1) & ~1, CE = 5 * NR;
    const long long blocks = (a.batch + 15) / 16;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    const size_t lds = ((((size_t)CE * 17 + 1) & ~(size_t)1) + (size_t)16 * NCP) * sizeof(double);
    hipLaunchKernelGGL((solve_rows2_kernel<NR, NRHS, RR>), dim3((unsigned)blocks), dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

template <int NR, int NRHS, int RR>
int launch_rows(const SolveArgs &a, hipStream_t s) {
    constexpr int NC = NR + NRHS, NCP = (NC + 1) & ~1, ES = NR * (NR > NRHS ? NR : NRHS);
    const int ipb = kRowsIpb;  // 64-byte runs in the interleaved layouts; measured: 4 and 16 instances are 2 % and 7 % slower
    const long long blocks = ((a.batch + ipb - 1) / ipb + 15) / 16 * 16;  // whole groups of 16: see the kernel's workgroup numbering
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    const size_t lds = ((((size_t)ES * (ipb | 1) + 1) & ~(size_t)1) + (size_t)ipb * NCP) * sizeof(double);
    hipLaunchKernelGGL((solve_rows_kernel<NR, NRHS, RR>), dim3((unsigned)blocks), dim3(ipb * 32), lds, s, a, ipb);
    return (int)hipGetLastError();
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

template <int G>
int launch_small(const SolveArgs &a, hipStream_t s) {
    const int nc = a.nrow + a.nrhs;
    int ldr = nc + 1;
    while ((ldr & 3) != 2) ++ldr;  // even (16-byte pairs) and 2 mod 4 (rows land on distinct LDS banks)
    const size_t per_inst = (size_t)a.nrow * ldr * sizeof(double);
    // instances per workgroup: the choice that keeps most instances resident on a CU (160 KB of LDS, 32 wavefronts);
    // on a tie the larger group, whose accesses to the interleaved layouts are wider
    int ipb = 0;
    size_t best = 0;
    for (int c = 16; c >= 1; c >>= 1) {
        if (per_inst * c > 64 * 1024 || c * G > 1024) continue;
        const size_t by_lds = (160 * 1024) / (per_inst * c), by_waves = (size_t)(32 * 64) / (size_t)(c * G < 64 ? 64 : c * G);
        const size_t resident = (by_lds < by_waves ? by_lds : by_waves) * c;
        if (resident > best) { best = resident; ipb = c; }
    }
    if (ipb == 0) return MPCX_ERR_UNSUPPORTED;
    const long long blocks = (a.batch + ipb - 1) / ipb;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    int threads = ipb * G;
    if (threads < 64) threads = 64;
    hipLaunchKernelGGL(solve_small_kernel<G>, dim3((unsigned)blocks), dim3(threads), per_inst * ipb, s, a, ipb, ldr);
    return (int)hipGetLastError();
}

int launch_solve(const SolveArgs &a, hipStream_t s) {
    if (a.nrow == 1) return launch_tiny<1>(a, s);
    if (a.nrow == 2) return launch_tiny<2>(a, s);
    if (a.nrow == 3) return launch_tiny<3>(a, s);
    if (a.nrow == 4) return launch_tiny<4>(a, s);
    // the reference's horizon with rows in registers: gain of a single-output plant, and the solve
    // (a single right-hand side: two rows per lane, 2.86 -> 2.57 ms on 2^20 systems; with 20 right-hand sides the
    //  80 row elements of a lane leave too few wavefronts resident and the one-row kernel is twice as fast)
#ifndef MPCX_ROWS1  // (-DMPCX_ROWS1: the one-row-per-lane kernel, for comparison)
    if (a.nrow == 20 && a.nrhs == 1 && !a.Q) return launch_rows2<20, 1, 0>(a, s);
#endif
    if (a.nrow == 20 && a.nrhs == 20 && a.Q && a.r == 1 && a.R != a.X) return launch_rows<20, 20, 1>(a, s);
    // (in place is fine here: a workgroup reads the right-hand sides of its own instances into LDS before it writes them)
    if (a.nrow == 20 && a.nrhs == 1 && !a.Q) return launch_rows<20, 1, 0>(a, s);
    if (a.nrow <= kSmallMaxRows) {
        const int rc = a.nrow <= 16 ? launch_small<16>(a, s) : a.nrow <= 32 ? launch_small<32>(a, s) : launch_small<64>(a, s);
        if (rc != MPCX_ERR_UNSUPPORTED) return rc;
    }
    return MPCX_ERR_UNSUPPORTED;  // the blocked solver takes it from here
}

// The plain one-wavefront kernel with the whole system in LDS.  The blocked solver is several times faster at every size
// (64 unknowns x 192 right-hand sides: 5.8 against 108 ms for 16 384 instances), so this is only the last resort for what
// the blocked solver refuses (a weight block wider than it serves, a workspace too small for one instance).  Takes a ready right-hand side.
int launch_solve_general(const SolveArgs &a, hipStream_t s) {
    if (a.Q) return MPCX_ERR_UNSUPPORTED;
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

// True when one of the one-wavefront kernels above takes the system (no device workspace needed); everything else goes
// to the blocked solver.
bool solve_fits_on_chip(int nr, int nrhs) {
    int ldr = nr + nrhs + 1;
    while ((ldr & 3) != 2) ++ldr;
    return nr <= kSmallMaxRows && (size_t)nr * ldr * sizeof(double) <= 64 * 1024;
}

int launch_qp_solve(const double *H, const double *fvec, double *u_seq, int *info, long long batch, long long W, int vm,
                    double *ws, size_t ws_doubles, hipStream_t s) {
    SolveArgs a{H, fvec, u_seq, info, batch, W, vm, 1, 1, nullptr, 0, 0};
    int rc = launch_solve(a, s);
    if (rc != MPCX_ERR_UNSUPPORTED) return rc;
    rc = launch_solve_large(H, fvec, nullptr, 0, u_seq, info, batch, W, vm, 1, 1, ws, ws_doubles, s);
    return rc != MPCX_ERR_UNSUPPORTED ? rc : launch_solve_general(a, s);
}

// X = H^-1 R with R handed over in X (overwritten in place), any size the solvers take.
int launch_solve_inplace(const double *H, double *X, int *info, long long batch, long long W, int nr, int nrhs, double *ws,
                         size_t ws_doubles, hipStream_t s, int transposed) {
    SolveArgs a{H, X, X, info, batch, W, nr, nrhs, 0, nullptr, 0, transposed};  // every kernel reads R completely before it writes X
    int rc = launch_solve(a, s);
    if (rc != MPCX_ERR_UNSUPPORTED) return rc;
    rc = launch_solve_large(H, X, nullptr, 0, X, info, batch, W, nr, nrhs, 0, ws, ws_doubles, s, transposed);
    return rc != MPCX_ERR_UNSUPPORTED ? rc : launch_solve_general(a, s);
}

namespace {
// R[k][c] = sum_row Gamma[row][k] W4[row][c] for a dense (f r) x (f r) weight: the literal M^T W4 of controller.rs:96.
__global__ __launch_bounds__(256) void gamma_t_w4_dense_kernel(const double *Gamma, const double *W4, double *out, long long batch,
                                                               long long W, int nr, int fr) {
    const long long per = (long long)nr * fr;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= batch * per) return;
    const long long b = idx / per, e = idx - b * per;
    const int c = (int)(e / nr), k = (int)(e - (long long)c * nr);
    double s = 0.0;
    for (int row = 0; row < fr; ++row) s += Gamma[elem_index(b, (long long)k * fr + row, per, W)] * W4[(long long)c * fr + row];
    out[elem_index(b, e, per, W)] = s;
}

}  // namespace

// Dense W4 ((f r) x (f r), the literal argument of form_lifted_matrices): compatibility path, right-hand side formed by
// a plain kernel, then solved in place.
int launch_qp_gain_dense(const double *H, const double *Gamma, const double *W4, double *gain, int *info, long long batch,
                         long long W, int m, int r, int f, int v, double *ws, size_t ws_doubles, hipStream_t s) {
    const int nr = v * m, nrhs = f * r;
    const long long total = batch * (long long)nr * nrhs;
    if ((total + 255) / 256 > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gamma_t_w4_dense_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Gamma, W4, gain, batch, W,
                       nr, nrhs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return launch_solve_inplace(H, gain, info, batch, W, nr, nrhs, ws, ws_doubles, s);
}

int launch_qp_gain(const double *H, const double *Gamma, const double *Q, double *gain, int *info, long long batch,
                   long long W, int m, int r, int f, int v, double *ws, size_t ws_doubles, hipStream_t s) {
    const int nr = v * m, nrhs = f * r;
    if (nr <= 64) {  // small systems: Gamma^T W4 is formed while the workgroup loads its instances
        SolveArgs a{H, Gamma, gain, info, batch, W, nr, nrhs, 0, Q, r, 0};
        const int rc = launch_solve(a, s);
        if (rc != MPCX_ERR_UNSUPPORTED) return rc;
    }
    const int rc = launch_solve_large(H, Gamma, Q, r, gain, info, batch, W, nr, nrhs, 0, ws, ws_doubles, s);
    if (rc != MPCX_ERR_UNSUPPORTED) return rc;
    // last resort (a weight block wider than the blocked solver serves): Gamma^T blockdiag(Q) formed in place, then the
    // plain LDS kernel where the system fits it
    const long long total = batch * (long long)nr * nrhs;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    SolveArgs a{H, gain, gain, info, batch, W, nr, nrhs, 0, nullptr, 0, 0};  // in place: R is read into LDS before X is written
    if ((size_t)(nr | 1) * (nr + nrhs) * sizeof(double) > 160 * 1024) return MPCX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gamma_t_q_kernel, dim3((unsigned)blocks), dim3(256), 0, s, Gamma, Q, gain, batch, W, m, r, f, v);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return launch_solve_general(a, s);
}

}  // namespace mpcx
