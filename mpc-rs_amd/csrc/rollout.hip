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
template <int G, bool HX, bool HY>
__global__ __launch_bounds__(kBlock) void rollout_group_kernel(RolloutArgs p) {
    extern __shared__ double lds[];
    constexpr int IPW = kBlock / G;
    constexpr bool kRegs = G <= 32;  // rows of A, B, C in registers, zero-padded to G columns
    const int g = threadIdx.x / G, i = threadIdx.x % G;
    long long b = (long long)blockIdx.x * IPW + g;
    if (b >= p.batch) b = p.batch - 1;  // idle groups shadow the last instance
    const int n = p.n, m = p.m, r = p.r;
    const long long W = p.W, tile = b / W, li = b - tile * W, T = p.T;
#define IN(ptr, E, e) ((ptr)[tile * (long long)(E) * W + (long long)(e) * W + li])
    // LDS per instance: [A | B | C when not in registers] x[2][G] u[kChunk][G] out[kChunk][2][G]
    const int mats = kRegs ? 0 : n * n + n * m + r * n;
    const int per_inst = mats + 2 * G + kChunk * G + kChunk * 2 * G;
    double *sA = lds + (size_t)g * per_inst, *sB = sA + (kRegs ? 0 : n * n), *sC = sB + (kRegs ? 0 : n * m);
    double *sx = sA + mats, *su = sx + 2 * G, *sOut = su + kChunk * G;  // step q: [x_{t+1} (G) | y_t (G)]
    constexpr int RG = kRegs ? G : 1;
    double rA[RG], rB[RG], rC[RG];
    if constexpr (kRegs) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            rA[j] = (i < n && j < n) ? IN(p.A, n * n, j * n + i) : 0.0;
            rB[j] = (i < n && j < m) ? IN(p.B, n * m, j * n + i) : 0.0;
            rC[j] = (i < r && j < n) ? IN(p.C, r * n, j * r + i) : 0.0;
        }
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
    double pre[kChunk];
    if (T > 0) {
#pragma unroll
        for (int q = 0; q < kChunk; ++q) pre[q] = ubase[(q < T ? q : T - 1) * ustep];
    }
    lds_barrier();
    // One chunk.  The first chunk is peeled off the loop so that every entry to the loop body has the same history
    // (kChunk loads, then the chunk's stores): the wait on `pre` then only covers the loads.
    auto chunk = [&](const long long t0) {
        const int steps = (int)((T - t0 < kChunk) ? T - t0 : kChunk);
#pragma unroll
        for (int q = 0; q < kChunk; ++q) su[q * G + i] = pre[q];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            const long long t = t0 + kChunk + q;
            pre[q] = ubase[(t < T ? t : T - 1) * ustep];
        }
        lds_barrier();
        // B u_t of the whole chunk first: it does not depend on the state, so it stays off the serial path
        double bu[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            double qv = 0.0;
            if constexpr (kRegs) {
#pragma unroll
                for (int k0 = 0; k0 < G; k0 += 4) {
                    if (k0 >= m) break;  // uniform
                    const double u0 = su[q * G + k0], u1 = su[q * G + k0 + 1], u2 = su[q * G + k0 + 2], u3 = su[q * G + k0 + 3];
                    qv += rB[k0] * u0;
                    qv += rB[k0 + 1] * u1;
                    qv += rB[k0 + 2] * u2;
                    qv += rB[k0 + 3] * u3;
                }
            } else {
                const int ii = i < n ? i : 0;
#pragma unroll 4
                for (int k = 0; k < m; ++k) qv += sB[k * n + ii] * su[q * G + k];
            }
            bu[q] = qv;
        }
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
                double xr[G];
#pragma unroll
                for (int j = 0; j < G; ++j) xr[j] = xc[j];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    s += rA[j] * xr[j];
                    y += rC[j] * xr[j];
                }
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
            out[i] = s;
            out[G + i] = y;
            lds_barrier();
        }
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            const int qc = q < steps ? q : steps - 1;
            if (HX) IN(p.X, EX, (t0 + 1 + qc) * n + ix) = sOut[qc * 2 * G + ix];
            if (HY) IN(p.Y, EY, (t0 + qc) * r + iy) = sOut[qc * 2 * G + G + iy];
        }
        lds_barrier();
    };
    if (T > 0) {
        chunk(0);
        for (long long t0 = kChunk; t0 < T; t0 += kChunk) chunk(t0);
    }
#undef IN
}

template <int G, bool HX, bool HY>
int launch_group_inst(const RolloutArgs &a, hipStream_t s) {
    constexpr int IPW = kBlock / G;
    const long long blocks = (a.batch + IPW - 1) / IPW;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    const size_t mats = G <= 32 ? 0 : (size_t)a.n * a.n + (size_t)a.n * a.m + (size_t)a.r * a.n;
    const size_t per_inst = mats + 2 * G + (size_t)kChunk * G + (size_t)kChunk * 2 * G;
    const size_t lds_bytes = per_inst * IPW * sizeof(double);
    if (lds_bytes > 160 * 1024) return MPCX_ERR_UNSUPPORTED;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)rollout_group_kernel<G, HX, HY>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((rollout_group_kernel<G, HX, HY>), dim3((unsigned)blocks), dim3(kBlock), lds_bytes, s, a);
    return (int)hipGetLastError();
}

template <int G>
int launch_group_one(const RolloutArgs &a, hipStream_t s) {
    if (a.X && a.Y) return launch_group_inst<G, true, true>(a, s);
    if (a.X) return launch_group_inst<G, true, false>(a, s);
    return launch_group_inst<G, false, true>(a, s);
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
