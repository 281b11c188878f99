// Open-loop rollout X[:,t+1] = A X[:,t] + B U[:,t], Y[:,t] = C X[:,t]  (reference: src/control/simulator.rs:5-35,
// and with T = 1 the step of src/control/controller.rs:101-115).
//
// Time is serial; parallelism comes from the batch.  Two kernels:
//   rollout_lane   tiny plants (same (n, m, r) set as the lane QP kernel): one lane per instance, A/B/C and the
//                  state in registers, every access of a wave one contiguous segment when W >= 64.
//   rollout_group  any n, m, r <= 64: 16, 32 or 64 lanes per instance, A/B/C of the instance in LDS, lane i owns
//                  state row i (and output row i); the state is exchanged through LDS between steps.
#include "mpcx_internal.h"

namespace mpcx {
bool qp_lane_supports(int n, int m, int r);
namespace {

constexpr int kBlock = 64;

template <int N, int M, int R>
__global__ __launch_bounds__(kBlock) void rollout_lane_kernel(RolloutArgs p) {
    const long long b = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (b >= p.batch) return;
    const long long W = p.W, tile = b / W, lane = b - tile * W, T = p.T;
#define IN(ptr, E, e) ((ptr)[tile * (long long)(E) * W + (long long)(e) * W + lane])
    double A[N][N], Bm[N][M], Cm[R][N], x[N];
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) A[i][j] = IN(p.A, N * N, j * N + i);
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) Bm[i][j] = IN(p.B, N * M, j * N + i);
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < R; ++i) Cm[i][j] = IN(p.C, R * N, j * R + i);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = IN(p.x0, N, i);
    const long long EX = (long long)N * (T + 1), EY = (long long)R * T, EU = (long long)M * T;
    if (p.X) {
        // simulator.rs:13-17: X is zero-initialised and x0 is copied inside the loop, so T = 0 leaves column 0 zero
#pragma unroll
        for (int i = 0; i < N; ++i) IN(p.X, EX, i) = (T > 0) ? x[i] : 0.0;
    }
    // Eight steps at a time: their inputs are fetched first, the steps run in registers, and the 8 (N + R) outputs leave
    // back to back (stores issued in bursts reach a higher write rate than stores spaced by the arithmetic that feeds
    // them, and a per-instance input fetched inside a step would wait for every store issued before it).
    constexpr int CH = 8;
    for (long long t0 = 0; t0 < T; t0 += CH) {
        const int steps = (int)(T - t0 < CH ? T - t0 : CH);  // wave-uniform
        double u[CH][M], xs[CH][N], ys[CH][R];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const long long t = t0 + (q < steps ? q : steps - 1);
#pragma unroll
            for (int k = 0; k < M; ++k) u[q][k] = p.u_shared ? p.U[t * M + k] : IN(p.U, EU, t * M + k);
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            if (q >= steps) break;
#pragma unroll
            for (int a = 0; a < R; ++a) {
                double y = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) y += Cm[a][j] * x[j];
                ys[q][a] = y;
            }
            double xn[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) s += A[i][j] * x[j];
                double qq = 0.0;
#pragma unroll
                for (int k = 0; k < M; ++k) qq += Bm[i][k] * u[q][k];
                xn[i] = s + qq;
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                x[i] = xn[i];
                xs[q][i] = xn[i];
            }
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            if (q >= steps) break;
            if (p.Y) {
#pragma unroll
                for (int a = 0; a < R; ++a) IN(p.Y, EY, (t0 + q) * R + a) = ys[q][a];
            }
            if (p.X) {
#pragma unroll
                for (int i = 0; i < N; ++i) IN(p.X, EX, (t0 + q + 1) * N + i) = xs[q][i];
            }
        }
    }
#undef IN
}

template <int N, int M, int R>
int launch_lane_one(const RolloutArgs &a, hipStream_t s) {
    const long long blocks = (a.batch + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((rollout_lane_kernel<N, M, R>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
    return (int)hipGetLastError();
}

// G lanes per instance (64 / G instances per wavefront); any interleave width.  A, B, C of each instance sit in LDS,
// lane i of a group owns state row i and output row i.  Work proceeds in chunks of kChunk steps: the inputs of the
// next chunk are fetched into registers before the outputs of the current chunk are flushed from their LDS staging
// area, so no load ever waits behind a store that was issued less than a chunk ago (loads and stores retire through
// one in-order counter), and each flush writes runs of kChunk columns.
#ifndef MPCX_RO_CHUNK
#define MPCX_RO_CHUNK 8
#endif
constexpr int kChunk = MPCX_RO_CHUNK;
// The line-aligned flush of the instance-major outputs relies on a full chunk advancing a row by a whole number of
// 128-byte lines: kChunk * rows doubles with even rows, i.e. kChunk a multiple of 8.
static_assert(kChunk % 8 == 0, "MPCX_RO_CHUNK must be a multiple of 8 (whole 128-byte lines per chunk for even row counts)");

// LDS doubles per instance, kept even (16-byte aligned pairs).  (Staggering the instances of a wavefront over the LDS
// banks by padding was measured: no gain, and the extra bytes cost resident wavefronts.)
__host__ __device__ constexpr int pad_instance(int doubles) { return (doubles + 1) & ~1; }
__host__ __device__ constexpr int ring_size(int need) {
    int s = 64;
    while (s < need) s *= 2;
    return s;
}

// HX / HY: X / Y is stored.  Every chunk issues exactly kChunk stores per output, whatever the lane, the group or the
// step: lanes beyond the last row repeat row 0's store, groups beyond the batch repeat the last instance's stores and
// the steps beyond T in the last chunk repeat the last step's (same address, same value).  With a fixed count the
// compiler can wait for the input loads alone (vmcnt(#stores since)) instead of draining the stores as well.
// NP: state entries a lane reads per step and columns of A, C it holds (n <= NP <= G, a multiple of 4); MB: columns of
// B it holds (m <= MB).  The narrow instantiations (NP <= 12, MB = 4) fit 128 registers (four waves per SIMD) with a
// shared input sequence and 168 (three waves) with per-instance inputs, which need prefetch registers.
// SHU: one input sequence shared by every instance (simulator::step's unit input): it is read with wave-uniform loads
// straight into the B u products, with no per-lane prefetch registers and no LDS staging.
// FF: instance-major outputs with even n, r and 16-byte aligned bases: a row of X or Y is one contiguous stream per
// instance, staged in an LDS ring and flushed as 16-byte pairs, the group's lanes side by side (256-byte runs,
// kChunk NP / 32 store instructions per output instead of kChunk).  Flushes end on the 128-byte line boundaries of
// the row in memory (the remainder waits in the ring): runs that start or end inside a line cost a fifth of the
// write rate (4.5 against 5.5 TB/s for this access shape, tools/stream_write.hip).
// Orders the LDS traffic of ONE wavefront: the LDS executes a wavefront's instructions in issue order, so a value one
// lane stores is what another lane of the same wavefront reads next without waiting for the store to complete; only the
// compiler has to be kept from moving accesses across this point.  (A workgroup here is a single wavefront.)
__device__ __forceinline__ void wave_lds_order() {
#ifdef MPCX_RO_FULL_BARRIER
    lds_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// One 16-byte pair of a row's stream, from the ring to memory, for the first and the last flush of a row: `a` is the
// pair's aligned coordinate (even), `f` its unwrapped ring position (below 3 RS); entries outside [lo, hi) or at and
// beyond the flush's end are not stored, and a pair with one live entry becomes an 8-byte store.
__device__ __forceinline__ void flush_pair_masked(double *row, const double *ring, int a, int f, int RS, int lo, int hi, int end) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    f &= RS - 1;  // the ring size is a power of two
    const bool in0 = a >= lo && a < end && a < hi;
    const bool in1 = a + 1 >= lo && a + 1 < end && a + 1 < hi;
    if (in0 && in1) *(v2d *)(row + a) = *(const v2d *)(ring + f);
    else if (in0) row[a] = ring[f];
    else if (in1) row[a + 1] = ring[f + 1];
}

template <int G, int NP, int MB, bool HX, bool HY, bool SHU, bool FF>
__global__ __launch_bounds__(kBlock, (NP <= 12 && MB <= 4) ? (SHU ? 4 : 3) : 1) void rollout_group_kernel(RolloutArgs p) {
    extern __shared__ double lds[];
    constexpr int IPW = kBlock / G;
    constexpr bool kRegs = G <= 32;  // rows of A, B, C in registers, zero-padded to G columns
    const int g = threadIdx.x / G, i = threadIdx.x % G;
    long long b = (long long)blockIdx.x * IPW + g;
    if (b >= p.batch) b = p.batch - 1;  // idle groups shadow the last instance
    const int n = p.n, m = p.m, r = p.r;
    const long long W = p.W, tile = b / W, li = b - tile * W, T = p.T;
#define IN(ptr, E, e) ((ptr)[tile * (long long)(E) * W + (long long)(e) * W + li])
    // LDS per instance: [A | B | C when not in registers] x[2][G] u[kChunk][SU] out[kChunk][2][G]
    constexpr int SU = (SHU && G <= 32) ? 0 : ((G <= 32) ? MB : G);  // inputs kept per step (none: shared inputs come through scalar loads)
    const int mats = kRegs ? 0 : n * n + n * m + r * n;
    // staging of a chunk's outputs: two rings of RS doubles (X, Y) for the line-aligned paired flush, else [step][2][G]
    // Ring of a row's stream: a full chunk, the part of a line that waits for the next flush, and the G - 1 entries
    // beyond a step's rows that the surplus lanes of a group overwrite (see the step), rounded up to a power of two so
    // that a position wraps with one AND (the step loop is bound by instruction issue: every integer instruction of the
    // addressing costs as much as a multiply-add).
    constexpr int RS = ring_size(kChunk * NP + 16 + G);
    const int stage = FF ? 2 * RS : kChunk * 2 * G;
    const int per_inst = pad_instance(mats + 2 * G + kChunk * SU + stage);
    double *sA = lds + (size_t)g * per_inst, *sB = sA + (kRegs ? 0 : n * n), *sC = sB + (kRegs ? 0 : n * m);
    double *sx = sA + mats, *su = sx + 2 * G, *sOut = su + kChunk * SU;  // step q: [x_{t+1} (G) | y_t (G)]
    constexpr int RG = kRegs ? NP : 1, RB = kRegs ? MB : 1;
    double rA[RG], rB[RB], rC[RG];
    if constexpr (kRegs) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            rA[j] = (i < n && j < n) ? IN(p.A, n * n, j * n + i) : 0.0;
            rC[j] = (i < r && j < n) ? IN(p.C, r * n, j * r + i) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < MB; ++j) rB[j] = (i < n && j < m) ? IN(p.B, n * m, j * n + i) : 0.0;
    } else {
        for (int e = i; e < n * n; e += G) sA[e] = IN(p.A, n * n, e);
        for (int e = i; e < n * m; e += G) sB[e] = IN(p.B, n * m, e);
        for (int e = i; e < r * n; e += G) sC[e] = IN(p.C, r * n, e);
    }
    const long long EX = (long long)n * (T + 1), EY = (long long)r * T, EU = (long long)m * T;
    {
        const double x0 = (i < n) ? IN(p.x0, n, i) : 0.0;
        sx[i] = x0;
        sx[G + i] = 0.0;
        // simulator.rs:13-17: X is zero-initialised and x0 is copied inside the loop, so T = 0 leaves column 0 zero
        if (HX && i < n) IN(p.X, EX, i) = (T > 0) ? x0 : 0.0;
    }
    // Input fetches are branch-free: a lane beyond m or a step beyond T reads a clamped (valid, unused) address, so the
    // compiler never has to wait for one load before issuing the next.  Lanes >= m only ever meet zero coefficients.
    const int ic = i < m ? i : 0, ix = i < n ? i : 0, iy = i < r ? i : 0;
    const double *ubase = p.u_shared ? p.U + ic : &IN(p.U, EU, ic);
    const long long ustep = p.u_shared ? (long long)m : (long long)m * W;
    // the shared input sequence is read-only for the whole launch: the constant address space makes its wave-uniform
    // reads scalar loads (no vector registers, no entry in the vector-memory counter)
    typedef const double __attribute__((address_space(4))) *ConstPtr;
    const ConstPtr ush = (ConstPtr)(unsigned long long)p.U;
    double pre[SHU ? 1 : kChunk];
    if (!SHU && T > 0) {
#pragma unroll
        for (int q = 0; q < kChunk; ++q) pre[q] = ubase[(q < T ? q : T - 1) * ustep];
    }
    // Paired flush: rows of X and Y as streams of doubles.  "Aligned coordinate" of a row = element index + mis, where
    // mis is the row start's offset within its 128-byte line, so that multiples of 16 are line boundaries; rowx / rowy
    // point at aligned coordinate 0 (a line start).  X's stream starts at element n (column 0 is stored above), Y's
    // at 0.  Everything else (first unflushed line, ring positions) follows from the chunk's first step.
    typedef double v2d __attribute__((ext_vector_type(2)));
    double *rowx = nullptr, *rowy = nullptr;
    int misx = 0, misy = 0;
    if constexpr (FF) {
        double *bx = p.X ? p.X + b * EX : nullptr, *by = p.Y ? p.Y + b * EY : nullptr;
        misx = (int)(((uintptr_t)bx >> 3) & 15);
        misy = (int)(((uintptr_t)by >> 3) & 15);
        rowx = bx - misx;
        rowy = by - misy;
    }
    lds_barrier();
    // One chunk.  The first chunk is peeled off the loop so that every entry to the loop body has the same history
    // (kChunk loads, then the chunk's stores): the wait on `pre` then only covers the loads.
#ifdef MPCX_RO_STAMPS
    long long st[5] = {0, 0, 0, 0, 0};
#define RSTAMP(k) st[k] = clock64()
#else
#define RSTAMP(k)
#endif
    auto chunk = [&](const long long t0) {
        const int steps = (int)((T - t0 < kChunk) ? T - t0 : kChunk);
        // (FF) aligned coordinates of the first entries this chunk produces, and their ring positions
        const int prodx = ((int)t0 + 1) * n + misx, prody = (int)t0 * r + misy;
        int cx = prodx + i, cy = prody + i;  // this lane's entry of step 0, unwrapped
        // (opaque to the optimiser: otherwise it re-associates cx + q n into chunk-invariant per-step lane offsets, sixteen
        // more live registers, and the step loop spills)
        asm volatile("" : "+v"(cx), "+v"(cy));
        RSTAMP(0);
        if constexpr (!SHU) {
#pragma unroll
            for (int q = 0; q < kChunk; ++q)
                if (i < SU) su[q * SU + i] = pre[q];
#pragma unroll
            for (int q = 0; q < kChunk; ++q) {
                const long long t = t0 + kChunk + q;
                pre[q] = ubase[(t < T ? t : T - 1) * ustep];
            }
            lds_barrier();
        }
        RSTAMP(1);
        // B u_t of the whole chunk first: it does not depend on the state, so it stays off the serial path
        double bu[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            double qv = 0.0;
            if constexpr (SHU && kRegs) {
                const long long tq = (t0 + q < T) ? t0 + q : T - 1;  // wave-uniform
#pragma unroll
                for (int k = 0; k < MB; ++k) qv += rB[k] * ush[tq * m + (k < m ? k : m - 1)];  // rB is zero past m
            } else if constexpr (kRegs) {
#pragma unroll
                for (int k0 = 0; k0 < MB; k0 += 4) {
                    if (k0 >= m) break;  // uniform
                    const double u0 = su[q * SU + k0], u1 = su[q * SU + k0 + 1], u2 = su[q * SU + k0 + 2], u3 = su[q * SU + k0 + 3];
                    qv += rB[k0] * u0;
                    qv += rB[k0 + 1] * u1;
                    qv += rB[k0 + 2] * u2;
                    qv += rB[k0 + 3] * u3;
                }
            } else {
                const int ii = i < n ? i : 0;
#pragma unroll 4
                for (int k = 0; k < m; ++k) qv += sB[k * n + ii] * su[q * SU + k];
            }
            bu[q] = qv;
        }
        // (pinned: left to itself the compiler sinks each product into its step, and the step then waits for the scalar
        // loads of the inputs on the counter its state reads use)
#pragma unroll
        for (int q = 0; q < kChunk; ++q) asm volatile("" : "+v"(bu[q]));
        RSTAMP(2);
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (q >= steps) break;  // uniform
            const long long t = t0 + q;
            const double *xc = sx + (t & 1) * G;
            double *xn = sx + ((t + 1) & 1) * G;
            double *out = sOut + q * 2 * G;
            double s = 0.0, y = 0.0;
            if constexpr (kRegs) {
                // all state reads go out before the first multiply-add: one LDS latency per step, not one per pair
                double xr[NP];
                typedef double v2d __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int j = 0; j < NP; j += 2) {  // 16-byte reads: half the LDS cycles of paired 8-byte reads
                    const v2d pr = *(const v2d *)(xc + j);
                    xr[j] = pr.x;
                    xr[j + 1] = pr.y;
                }
                __builtin_amdgcn_sched_barrier(0);
                // two partial sums per dot product: four independent multiply-add chains instead of two
                double s1 = 0.0, y1 = 0.0;
#pragma unroll
                for (int j = 0; j < NP; j += 2) {
                    s += rA[j] * xr[j];
                    s1 += rA[j + 1] * xr[j + 1];
                    y += rC[j] * xr[j];
                    y1 += rC[j + 1] * xr[j + 1];
                }
                s += s1;
                y += y1;
            } else {
                const int ii = i < n ? i : 0, ir = i < r ? i : 0;
#pragma unroll 4
                for (int j = 0; j < n; ++j) {
                    const double xj = xc[j];
                    s += sA[j * n + ii] * xj;
                    y += sC[j * r + ir] * xj;
                }
                if (i >= n) s = 0.0;
            }
            s += bu[q];
            if (!kRegs && i >= n) s = 0.0;
            // rows at and beyond n (r) carry zero coefficients, so every lane writes: no divergence inside the step
            xn[i] = s;
            if constexpr (FF) {
                // Every lane stores: the lanes beyond row n - 1 (r - 1) carry zero coefficients and write zeros over the
                // positions of the next step's first rows, which that step (later in this wavefront's program order)
                // overwrites; past the last step of a chunk they land in the ring's spare entries, which are never flushed
                // before the next chunk has written them.  No branch and no compare inside the step.
                sOut[(cx + q * n) & (RS - 1)] = s;
                sOut[RS + ((cy + q * r) & (RS - 1))] = y;
            } else {
                out[i] = s;
                out[G + i] = y;
            }
            wave_lds_order();
        }
        RSTAMP(3);
        if constexpr (FF) {
            // Flush up to the last 128-byte line boundary of each row that the chunk has reached; what lies beyond it
            // (less than a line) stays in the ring for the next chunk.  A full chunk advances a row by kChunk n doubles,
            // a whole number of lines (n is even), so between the first and the last chunk every flush is the same
            // number of whole lines.  The first flush starts inside a line (a row starts wherever the previous one ended)
            // and the last one ends inside one: those two mask by element.
            const bool last = t0 + steps >= T;
            const bool edge = last || t0 == 0;  // wave-uniform
            const int ax = prodx & ~15, ay = prody & ~15;  // first unflushed line of each row
            const int fx = ax & (RS - 1), fy = ay & (RS - 1);
            if (edge) {
                constexpr int NSTE = (kChunk * NP + 16 + 2 * G - 1) / (2 * G);  // one more line than a full chunk
                const int endx = (int)EX + misx, endy = (int)EY + misy;
                const int ax_new = last ? ((endx + 15) & ~15) : ((prodx + steps * n) & ~15);
                const int ay_new = last ? ((endy + 15) & ~15) : ((prody + steps * r) & ~15);
#pragma unroll
                for (int k = 0; k < NSTE; ++k) {
                    const int e2 = 2 * (i + k * G);  // offset of this lane's pair from the first unflushed line
                    if (HX) flush_pair_masked(rowx, sOut, ax + e2, fx + e2, RS, n + misx, last ? endx : ax_new, ax_new);
                    if (HY) flush_pair_masked(rowy, sOut + RS, ay + e2, fy + e2, RS, misy, last ? endy : ay_new, ay_new);
                }
            } else {
                constexpr int NST = (kChunk * NP + 2 * G - 1) / (2 * G);  // pair stores per lane and output
                // every ring read first, then the stores: one LDS latency per flush instead of one per store
                v2d vx[NST], vy[NST];
#pragma unroll
                for (int k = 0; k < NST; ++k) {
                    const int e2 = 2 * (i + k * G);
                    const int f = (fx + e2) & (RS - 1), h = (fy + e2) & (RS - 1);
                    if (HX) vx[k] = *(const v2d *)(sOut + (e2 < kChunk * n ? f : 0));
                    if (HY) vy[k] = *(const v2d *)(sOut + RS + (e2 < kChunk * r ? h : 0));
                }
#pragma unroll
                for (int k = 0; k < NST; ++k) {
                    const int e2 = 2 * (i + k * G);
                    if (HX && e2 < kChunk * n) *(v2d *)(rowx + ax + e2) = vx[k];
                    if (HY && e2 < kChunk * r) *(v2d *)(rowy + ay + e2) = vy[k];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < kChunk; ++q) {
                const int qc = q < steps ? q : steps - 1;
                if (HX) IN(p.X, EX, (t0 + 1 + qc) * n + ix) = sOut[qc * 2 * G + ix];
                if (HY) IN(p.Y, EY, (t0 + qc) * r + iy) = sOut[qc * 2 * G + G + iy];
            }
        }
        lds_barrier();
        RSTAMP(4);
    };
    if (T > 0) {
        chunk(0);
        for (long long t0 = kChunk; t0 < T; t0 += kChunk) chunk(t0);
    }
#ifdef MPCX_RO_STAMPS
    if (threadIdx.x == 0 && HY) {
        double *o = p.Y + (long long)blockIdx.x * 8;
        for (int k = 0; k < 4; ++k) o[k] = (double)(st[k + 1] - st[k]);
    }
#endif
#undef IN
}

constexpr int kQuadStage = 192;  // steps of the shared input sequence staged in LDS at a time (a multiple of kChunk)
static_assert(kQuadStage % kChunk == 0, "whole chunks");
// FOUR lanes per instance (sixteen instances per wavefront) for instance-major batches of plants with n, r <= 4 RP
// states / outputs, m <= 4 inputs and a shared input sequence.  Lane pp of a quad owns state rows and output rows
// RP pp .. RP pp + RP - 1 with their rows of A, B, C in registers; the state is exchanged inside the quad with
// data-parallel-primitive moves (quad_perm broadcasts), so a step is multiply-adds and register moves only: no LDS
// access, no barrier and no idle lane on the serial path (the 16-lane form spends six LDS reads per lane and step on
// state that every lane of a group reads again, and is bound by that round trip).  The outputs of a chunk of kChunk
// steps stay in registers, go to the same per-instance LDS rings as in the 16-lane form afterwards and are flushed by
// the same line-aligned paired stores, 16 lanes per instance, four instances per pass.
template <int RP, bool HX, bool HY>
__global__ __launch_bounds__(kBlock) void rollout_quad_kernel(RolloutArgs p) {
    extern __shared__ double lds[];
    typedef double v2d __attribute__((ext_vector_type(2)));
    constexpr int NP = 4 * RP, MB = 4, IPW = kBlock / 4, G = 16;
    constexpr int LB = ring_size(kChunk * NP + 16 + G);  // a chunk, the part of a line carried over, the edge flush's overshoot
    constexpr int per_inst = 2 * LB;                     // staging buffers of X and Y
    const int lane = threadIdx.x, qi = lane >> 2, pp = lane & 3;
    const int n = p.n, m = p.m, r = p.r;
    const long long T = p.T, EX = (long long)n * (T + 1), EY = (long long)r * T;
    long long b = (long long)blockIdx.x * IPW + qi;
    if (b >= p.batch) b = p.batch - 1;  // idle quads shadow the last instance (same values to the same addresses)
    double rA[RP][NP], rC[RP][NP], rB[RP][MB], xo[RP];
    {
        const double *Ab = p.A + b * n * n, *Bb = p.B + b * n * m, *Cb = p.C + b * r * n, *x0b = p.x0 + b * n;
#pragma unroll
        for (int j = 0; j < RP; ++j) {
            const int row = RP * pp + j;
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                rA[j][c] = (row < n && c < n) ? Ab[c * n + row] : 0.0;
                rC[j][c] = (row < r && c < n) ? Cb[c * r + row] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < MB; ++k) rB[j][k] = (row < n && k < m) ? Bb[k * n + row] : 0.0;
            xo[j] = row < n ? x0b[row] : 0.0;
            // simulator.rs:13-17: X is zero-initialised and x0 is copied inside the loop, so T = 0 leaves column 0 zero
            if (HX && row < n) p.X[b * EX + row] = (T > 0) ? xo[j] : 0.0;
        }
    }
    // the shared input sequence, padded to four inputs per step, is staged in LDS kQuadStage steps at a time
    double *sU = lds + (size_t)IPW * per_inst;
    // Streams of the instance this lane computes for ("aligned coordinate" = element index + offset of the row start within
    // its 128-byte line, as in the 16-lane form).  Each stream has a LINEAR staging buffer of LB entries per instance:
    // entry 0 is the first unflushed line boundary of the row, the part of a line that a chunk leaves unflushed (fewer
    // than 16 entries) is moved to the front before the next chunk, and a chunk's entries follow it.  No position wraps,
    // so a step's stores are "base + step stride" with the lane's rows at immediate offsets; a lane whose row the plant
    // does not have carries zero coefficients and writes a zero where the next step (later in program order) or the next
    // chunk writes again, or beyond everything that is flushed.
    const int misx = HX ? (int)(((uintptr_t)(p.X + b * EX) >> 3) & 15) : 0;
    const int misy = HY ? (int)(((uintptr_t)(p.Y + b * EY) >> 3) & 15) : 0;
    const int bufX = qi * per_inst + RP * pp, bufY = bufX + LB;  // offsets into lds[] (LDS addressing stays visible)
    // at flush time a lane stores pairs of four other instances (instance 4 ig + lane / 16 of the wavefront, pair lane % 16)
    const int i = lane & 15;
    auto chunk = [&](const long long t0) {
        const int steps = (int)((T - t0 < kChunk) ? T - t0 : kChunk);
        if (t0 % kQuadStage == 0) {  // (a multiple of kChunk; the previous chunk's flush has ended with a barrier)
            const long long cnt = ((T - t0 < kQuadStage) ? T - t0 : kQuadStage) * MB;
            for (long long idx = lane; idx < cnt; idx += kBlock) {
                const long long t = idx / MB;
                const int k = (int)(idx - t * MB);
                sU[idx] = k < m ? p.U[(t0 + t) * m + k] : 0.0;
            }
            lds_barrier();
        }
        const double *sUc = sU + (t0 % kQuadStage) * MB;
        // where this chunk's first step lands: behind the entries carried over from the previous chunk
        const int wx = bufX + ((((int)t0 + 1) * n + misx) & 15), wy = bufY + (((int)t0 * r + misy) & 15);
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            if (q >= steps) break;  // uniform
            // (formed here and kept opaque: the optimiser would otherwise carry one address per step and output through
            // the whole chunk)
            int wxq = wx + q * n, wyq = wy + q * r;
            asm volatile("" : "+v"(wxq), "+v"(wyq));
            // the step's input: one broadcast read of the staged sequence, issued ahead of the multiply-adds that hide it
            double uq[MB];
            {
                const v2d u01 = *(const v2d *)(sUc + q * MB), u23 = *(const v2d *)(sUc + q * MB + 2);
                uq[0] = u01.x;
                uq[1] = u01.y;
                uq[2] = u23.x;
                uq[3] = u23.y;
            }
            double xf[NP];  // the whole state, gathered from the quad
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                xf[j] = dpp_f64<0x00>(xo[j]);
                xf[RP + j] = dpp_f64<0x55>(xo[j]);
                xf[2 * RP + j] = dpp_f64<0xAA>(xo[j]);
                xf[3 * RP + j] = dpp_f64<0xFF>(xo[j]);
            }
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                double s0 = 0.0, s1 = 0.0, y0 = 0.0, y1 = 0.0;
#pragma unroll
                for (int c = 0; c < NP; c += 2) {
                    s0 += rA[j][c] * xf[c];
                    s1 += rA[j][c + 1] * xf[c + 1];
                    if (HY) {
                        y0 += rC[j][c] * xf[c];
                        y1 += rC[j][c + 1] * xf[c + 1];
                    }
                }
                double bu = 0.0;
#pragma unroll
                for (int k = 0; k < MB; ++k) bu += rB[j][k] * uq[k];
                xo[j] = (s0 + s1) + bu;
                // into the staging buffers of this lane's instance: fire and forget, nothing on the serial path reads them
                if (HX) lds[wxq + j] = xo[j];
                if (HY) lds[wyq + j] = y0 + y1;
            }
            // (steps are not interleaved by the scheduler: what one step keeps live is all a step needs)
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
        // flush: up to the last 128-byte line boundary each row has reached; the first and the last flush of a row mask
        // by element (see the 16-lane form); what lies beyond the boundary moves to the front of the buffer
        const bool last = t0 + steps >= T;
        const bool edge = last || t0 == 0;  // wave-uniform
#pragma unroll
        for (int ig = 0; ig < 4; ++ig) {
            double *sOut = lds + (size_t)(ig * 4 + (lane >> 4)) * per_inst;
            long long bf = (long long)blockIdx.x * IPW + ig * 4 + (lane >> 4);
            if (bf >= p.batch) bf = p.batch - 1;
            double *bx = HX ? p.X + bf * EX : nullptr, *by = HY ? p.Y + bf * EY : nullptr;
            const int fmx = (int)(((uintptr_t)bx >> 3) & 15), fmy = (int)(((uintptr_t)by >> 3) & 15);
            double *rowx = bx - fmx, *rowy = by - fmy;
            const int prodx = ((int)t0 + 1) * n + fmx, prody = (int)t0 * r + fmy;
            const int ax = prodx & ~15, ay = prody & ~15;  // first unflushed line of each row = entry 0 of its buffer
            const int lx = ((prodx & 15) + steps * n) & ~15, ly = ((prody & 15) + steps * r) & ~15;  // entries flushed now
            if (edge) {
                constexpr int NSTE = (kChunk * NP + 16 + 2 * G - 1) / (2 * G);  // one more line than a full chunk
                const int endx = (int)EX + fmx, endy = (int)EY + fmy;
                const int ax_new = last ? ((endx + 15) & ~15) : ax + lx;
                const int ay_new = last ? ((endy + 15) & ~15) : ay + ly;
#pragma unroll
                for (int k = 0; k < NSTE; ++k) {
                    const int e2 = 2 * (i + k * G);
                    if (HX) flush_pair_masked(rowx, sOut, ax + e2, e2, LB, n + fmx, last ? endx : ax_new, ax_new);
                    if (HY) flush_pair_masked(rowy, sOut + LB, ay + e2, e2, LB, fmy, last ? endy : ay_new, ay_new);
                }
            } else {
                constexpr int NST = (kChunk * NP + 2 * G - 1) / (2 * G);  // pair stores per lane and output
                v2d vx[NST], vy[NST];
#pragma unroll
                for (int k = 0; k < NST; ++k) {
                    const int e2 = 2 * (i + k * G);
                    if (HX) vx[k] = *(const v2d *)(sOut + (e2 < kChunk * n ? e2 : 0));
                    if (HY) vy[k] = *(const v2d *)(sOut + LB + (e2 < kChunk * r ? e2 : 0));
                }
#pragma unroll
                for (int k = 0; k < NST; ++k) {
                    const int e2 = 2 * (i + k * G);
                    if (HX && e2 < kChunk * n) *(v2d *)(rowx + ax + e2) = vx[k];
                    if (HY && e2 < kChunk * r) *(v2d *)(rowy + ay + e2) = vy[k];
                }
            }
            if (!last) {  // carry: the (fewer than 16) entries behind the flushed lines go to the front
                const double cxv = HX ? sOut[lx + i] : 0.0, cyv = HY ? sOut[LB + ly + i] : 0.0;
                if (HX) sOut[i] = cxv;
                if (HY) sOut[LB + i] = cyv;
            }
        }
        lds_barrier();
    };
    for (long long t0 = 0; t0 < T; t0 += kChunk) chunk(t0);
}

template <int RP>
int launch_quad(const RolloutArgs &a, hipStream_t s) {
    constexpr int IPW = kBlock / 4, LB = ring_size(kChunk * 4 * RP + 16 + 16);
    const long long blocks = (a.batch + IPW - 1) / IPW;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    const size_t lds_bytes = ((size_t)IPW * 2 * LB + (size_t)kQuadStage * 4) * sizeof(double);  // staging buffers, staged inputs: 38 KB, four per CU
    if (a.X && a.Y) hipLaunchKernelGGL((rollout_quad_kernel<RP, true, true>), dim3((unsigned)blocks), dim3(kBlock), lds_bytes, s, a);
    else if (a.X) hipLaunchKernelGGL((rollout_quad_kernel<RP, true, false>), dim3((unsigned)blocks), dim3(kBlock), lds_bytes, s, a);
    else hipLaunchKernelGGL((rollout_quad_kernel<RP, false, true>), dim3((unsigned)blocks), dim3(kBlock), lds_bytes, s, a);
    return (int)hipGetLastError();
}

template <int G, int NP, int MB, bool HX, bool HY, bool SHU, bool FF>
int launch_group_inst(const RolloutArgs &a, hipStream_t s) {
    constexpr int IPW = kBlock / G;
    const long long blocks = (a.batch + IPW - 1) / IPW;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    const size_t mats = G <= 32 ? 0 : (size_t)a.n * a.n + (size_t)a.n * a.m + (size_t)a.r * a.n;
    const size_t su = (SHU && G <= 32) ? 0 : (G <= 32 ? MB : G);
    const size_t stage = FF ? (size_t)2 * ring_size(kChunk * NP + 16 + G) : (size_t)kChunk * 2 * G;
    const size_t per_inst = (size_t)pad_instance((int)(mats + 2 * G + (size_t)kChunk * su + stage));
    const size_t lds_bytes = per_inst * IPW * sizeof(double);
    if (lds_bytes > 160 * 1024) return MPCX_ERR_UNSUPPORTED;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)rollout_group_kernel<G, NP, MB, HX, HY, SHU, FF>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((rollout_group_kernel<G, NP, MB, HX, HY, SHU, FF>), dim3((unsigned)blocks), dim3(kBlock), lds_bytes, s, a);
    return (int)hipGetLastError();
}

template <int G, int NP, int MB>
int launch_group_np(const RolloutArgs &a, hipStream_t s) {
    if constexpr (G <= 32) {
        // (the paired flush indexes a row with 32-bit offsets)
        const bool short_rows = (long long)(a.n > a.r ? a.n : a.r) * (a.T + 2) < 0x7fffff00LL;
        const bool ff = a.W == 1 && a.n % 2 == 0 && a.r % 2 == 0 && (((uintptr_t)a.X | (uintptr_t)a.Y) & 15) == 0 && short_rows;
        if (a.u_shared && ff) {
            if (a.X && a.Y) return launch_group_inst<G, NP, MB, true, true, true, true>(a, s);
            if (a.X) return launch_group_inst<G, NP, MB, true, false, true, true>(a, s);
            return launch_group_inst<G, NP, MB, false, true, true, true>(a, s);
        }
        if (a.u_shared) {
            if (a.X && a.Y) return launch_group_inst<G, NP, MB, true, true, true, false>(a, s);
            if (a.X) return launch_group_inst<G, NP, MB, true, false, true, false>(a, s);
            return launch_group_inst<G, NP, MB, false, true, true, false>(a, s);
        }
        if (ff) {
            if (a.X && a.Y) return launch_group_inst<G, NP, MB, true, true, false, true>(a, s);
            if (a.X) return launch_group_inst<G, NP, MB, true, false, false, true>(a, s);
            return launch_group_inst<G, NP, MB, false, true, false, true>(a, s);
        }
    }
    if (a.X && a.Y) return launch_group_inst<G, NP, MB, true, true, false, false>(a, s);
    if (a.X) return launch_group_inst<G, NP, MB, true, false, false, false>(a, s);
    return launch_group_inst<G, NP, MB, false, true, false, false>(a, s);
}
template <int G>
int launch_group_one(const RolloutArgs &a, hipStream_t s) {
    if constexpr (G == 16) {
#ifndef MPCX_RO_NO_QUAD  // (-DMPCX_RO_NO_QUAD: the 16-lane form everywhere, for comparison)
        // instance-major, shared inputs, even row counts (the line-aligned paired flush): four lanes per instance
        const bool short_rows = (long long)(a.n > a.r ? a.n : a.r) * (a.T + 2) < 0x7fffff00LL;
        const bool ff = a.W == 1 && a.n % 2 == 0 && a.r % 2 == 0 && (((uintptr_t)a.X | (uintptr_t)a.Y) & 15) == 0 && short_rows;
        // ... from about one round of the 16-lane form's resident wavefronts on (16 per CU x 4 instances x 256 CUs): a lone
        // wavefront of the four-lane form takes longer per step (more instructions per lane), so small batches, which
        // are bound by that latency, stay with sixteen lanes.  Quadrotor plants, T = 200, 16-lane / four-lane form:
        // 2 048 instances 0.075 / 0.114 ms, 8 192 0.108 / 0.133, 16 384 0.186 / 0.165, 32 768 0.369 / 0.347 ms.
        if (ff && a.u_shared && a.m <= 4 && (a.X || a.Y) && a.batch >= 12288) {
            if (a.n <= 8 && a.r <= 8) return launch_quad<2>(a, s);
            if (a.n <= 12 && a.r <= 12) return launch_quad<3>(a, s);
        }
#endif
        // NP also bounds the rows of Y the compact flush stores (NST pair stores cover kChunk NP doubles per output),
        // so the narrow instantiations need r <= NP as well as n <= NP
        if (a.m <= 4 && a.n <= 8 && a.r <= 8) return launch_group_np<16, 8, 4>(a, s);
        if (a.m <= 4 && a.n <= 12 && a.r <= 12) return launch_group_np<16, 12, 4>(a, s);
    }
    return launch_group_np<G, G, G>(a, s);
}

}  // namespace

int launch_rollout_lane(const RolloutArgs &a, hipStream_t s) {
    if (!qp_lane_supports(a.n, a.m, a.r)) return MPCX_ERR_UNSUPPORTED;
#define CASE(N_, M_, R_) \
    if (a.n == N_ && a.m == M_ && a.r == R_) return launch_lane_one<N_, M_, R_>(a, s);
    CASE(1, 1, 1) CASE(2, 1, 1) CASE(3, 1, 1) CASE(4, 1, 1) CASE(5, 1, 1) CASE(6, 1, 1)
    CASE(2, 2, 2) CASE(4, 2, 2) CASE(6, 2, 2)
    CASE(3, 2, 1) CASE(4, 2, 1) CASE(3, 1, 2) CASE(4, 1, 2)
#undef CASE
    return MPCX_ERR_UNSUPPORTED;
}

int launch_rollout_group(const RolloutArgs &a, hipStream_t s) {
    const int big = a.n > a.m ? (a.n > a.r ? a.n : a.r) : (a.m > a.r ? a.m : a.r);
    if (big > kBlock) return MPCX_ERR_UNSUPPORTED;
    if (big <= 16) return launch_group_one<16>(a, s);
    if (big <= 32) return launch_group_one<32>(a, s);
    return launch_group_one<64>(a, s);
}

}  // namespace mpcx
