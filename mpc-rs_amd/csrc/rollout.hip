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
    for (long long t = 0; t < T; ++t) {
        double u[M];
#pragma unroll
        for (int k = 0; k < M; ++k) u[k] = p.u_shared ? p.U[t * M + k] : IN(p.U, EU, t * M + k);
        if (p.Y) {
#pragma unroll
            for (int a = 0; a < R; ++a) {
                double y = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) y += Cm[a][j] * x[j];
                IN(p.Y, EY, t * R + a) = y;
            }
        }
        double xn[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) s += A[i][j] * x[j];
            double q = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) q += Bm[i][k] * u[k];
            xn[i] = s + q;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = xn[i];
        if (p.X) {
#pragma unroll
            for (int i = 0; i < N; ++i) IN(p.X, EX, (t + 1) * N + i) = x[i];
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
constexpr int kChunk = 8;

// HX / HY: X / Y is stored.  Every chunk issues exactly kChunk stores per output, whatever the lane, the group or the
// step: lanes beyond the last row repeat row 0's store, groups beyond the batch repeat the last instance's stores and
// the steps beyond T in the last chunk repeat the last step's (same address, same value).  With a fixed count the
// compiler can wait for the input loads alone (vmcnt(#stores since)) instead of draining the stores as well.
// NP: state entries a lane reads per step and columns of A, C it holds (n <= NP <= G, a multiple of 4); MB: columns of
// B it holds (m <= MB).  The narrow instantiations (NP <= 12, MB = 4) fit 128 registers (four waves per SIMD) with a
// shared input sequence and 168 (three waves) with per-instance inputs, which need prefetch registers.
// SHU: one input sequence shared by every instance (simulator::step's unit input): it is read with wave-uniform loads
// straight into the B u products, with no per-lane prefetch registers and no LDS staging.
// FF: instance-major outputs with even n, r and 16-byte aligned bases: a chunk of X (kChunk n doubles) and of Y
// (kChunk r doubles) is one contiguous run per instance, staged compactly in LDS and flushed as 16-byte pairs, the
// group's lanes side by side (256-byte runs, kChunk NP / 32 store instructions per output instead of kChunk).
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
    constexpr int SU = (G <= 32) ? MB : G;  // inputs kept per step
    const int mats = kRegs ? 0 : n * n + n * m + r * n;
    const int per_inst = mats + 2 * G + kChunk * SU + kChunk * 2 * G;
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
#pragma unroll
                for (int j = 0; j < NP; ++j) xr[j] = xc[j];
                __builtin_amdgcn_sched_barrier(0);
                // two partial sums per dot product: four independent multiply-add chains instead of two
                double s1 = 0.0, y1 = 0.0;
#pragma unroll
                for (int j = 0; j < NP; j += 2) {
                    s += rA[j] * xr[j];
                    y += rC[j] * xr[j];
                    s1 += rA[j + 1] * xr[j + 1];
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
                if (i < n) sOut[q * n + i] = s;
                if (i < r) sOut[kChunk * G + q * r + i] = y;
            } else {
                out[i] = s;
                out[G + i] = y;
            }
            lds_barrier();
        }
        RSTAMP(3);
        if constexpr (FF) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            constexpr int NST = (kChunk * NP + 2 * G - 1) / (2 * G);  // pair stores per lane and output
            const int lastx = steps * n / 2 - 1, lasty = steps * r / 2 - 1;  // last pair of this chunk (n, r even)
            double *gx = p.X + b * EX + (t0 + 1) * n, *gy = p.Y + b * EY + t0 * r;
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int e = i + k * G;
                if (HX) { const int ex = e < lastx ? e : lastx; *(v2d *)(gx + 2 * ex) = *(const v2d *)(sOut + 2 * ex); }
                if (HY) { const int ey = e < lasty ? e : lasty; *(v2d *)(gy + 2 * ey) = *(const v2d *)(sOut + kChunk * G + 2 * ey); }
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

template <int G, int NP, int MB, bool HX, bool HY, bool SHU, bool FF>
int launch_group_inst(const RolloutArgs &a, hipStream_t s) {
    constexpr int IPW = kBlock / G;
    const long long blocks = (a.batch + IPW - 1) / IPW;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    const size_t mats = G <= 32 ? 0 : (size_t)a.n * a.n + (size_t)a.n * a.m + (size_t)a.r * a.n;
    const size_t per_inst = mats + 2 * G + (size_t)kChunk * (G <= 32 ? MB : G) + (size_t)kChunk * 2 * G;
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
        const bool ff = a.W == 1 && a.n % 2 == 0 && a.r % 2 == 0 && (((uintptr_t)a.X | (uintptr_t)a.Y) & 15) == 0;
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
