// Open-loop rollout X[:,t+1] = A X[:,t] + B U[:,t], Y[:,t] = C X[:,t]  (reference: src/control/simulator.rs:5-35,
// and with T = 1 the step of src/control/controller.rs:101-115).
//
// Time is serial; parallelism comes from the batch.  Two kernels:
//   rollout_lane   tiny plants (same (n, m, r) set as the lane QP kernel): one lane per instance, A/B/C and the
//                  state in registers, every access of a wave one contiguous segment when W >= 64.
//   rollout_group  any n, m, r <= 64: one wavefront per instance, A/B/C of the instance in LDS, lane i owns
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

// One wavefront per instance; any interleave width.
__global__ __launch_bounds__(kBlock) void rollout_group_kernel(RolloutArgs p) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long long b = blockIdx.x;
    const int n = p.n, m = p.m, r = p.r;
    const long long W = p.W, tile = b / W, li = b - tile * W, T = p.T;
#define IN(ptr, E, e) ((ptr)[tile * (long long)(E) * W + (long long)(e) * W + li])
    double *sA = lds, *sB = sA + n * n, *sC = sB + n * m, *sx = sC + r * n, *su = sx + 2 * n;
    for (int e = lane; e < n * n; e += kBlock) sA[e] = IN(p.A, n * n, e);
    for (int e = lane; e < n * m; e += kBlock) sB[e] = IN(p.B, n * m, e);
    for (int e = lane; e < r * n; e += kBlock) sC[e] = IN(p.C, r * n, e);
    const long long EX = (long long)n * (T + 1), EY = (long long)r * T, EU = (long long)m * T;
    if (lane < n) {
        const double x0 = IN(p.x0, n, lane);
        sx[lane] = x0;
        if (p.X) IN(p.X, EX, lane) = (T > 0) ? x0 : 0.0;
    }
    lds_barrier();
    for (long long t = 0; t < T; ++t) {
        const double *xc = sx + (t & 1) * n;
        double *xn = sx + ((t + 1) & 1) * n;
        if (lane < m) su[lane] = p.u_shared ? p.U[t * m + lane] : IN(p.U, EU, t * m + lane);
        lds_barrier();
        if (lane < n) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += sA[j * n + lane] * xc[j];
            double q = 0.0;
            for (int k = 0; k < m; ++k) q += sB[k * n + lane] * su[k];
            s += q;
            xn[lane] = s;
            if (p.X) IN(p.X, EX, (t + 1) * n + lane) = s;
        }
        if (p.Y && lane < r) {
            double y = 0.0;
            for (int j = 0; j < n; ++j) y += sC[j * r + lane] * xc[j];
            IN(p.Y, EY, t * r + lane) = y;
        }
        lds_barrier();
    }
#undef IN
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
    if (a.n > kBlock || a.m > kBlock || a.r > kBlock) return MPCX_ERR_UNSUPPORTED;
    if (a.batch > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    const size_t lds_bytes = sizeof(double) * ((size_t)a.n * a.n + (size_t)a.n * a.m + (size_t)a.r * a.n + 2 * a.n + a.m);
    hipLaunchKernelGGL(rollout_group_kernel, dim3((unsigned)a.batch), dim3(kBlock), lds_bytes, s, a);
    return (int)hipGetLastError();
}

}  // namespace mpcx
