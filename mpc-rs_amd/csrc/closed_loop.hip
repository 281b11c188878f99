// Batched receding-horizon closed loop: for t = 0 .. T-1
//     s   = ref[t .. t+f) - Phi x_t                  (controller.rs:119-127)
//     u_t = first m entries of gain s                (controller.rs:130-132; the reference is m = r = 1)
//     y_t = C x_t,  x_{t+1} = A x_t + B u_t          (controller.rs:101-115)
// which is the loop of src/main.rs:51-71 run for `batch` plants at once.  Phi and gain are the per-instance outputs of
// the QP build and of the gain solve; only the first m rows of the gain are read.
//
// The reference trajectory is r x (T + f), time along columns (for r = 1 this is the reference's T_total x 1
// desired_ctrl_traj_total with time_steps = T + f, main.rs:18-24); step t uses columns t .. t+f-1.
//
// Time is serial, parallelism comes from the batch.  Two kernels, as for the open-loop rollout:
//   closed_loop_lane   tiny plants: one lane per instance; A, B, C, x in registers; Phi and the first gain rows in a
//                      per-lane LDS column (no barriers: a lane only ever touches its own column).
//   closed_loop_group  any n, m, r <= 64: one workgroup of four wavefronts per instance, everything in LDS.
//   closed_loop_affine instance-major batches with m <= 4, n <= 12, r <= 16: the window products leave the recursion
//                      (u_t = K w_t - (K Phi) x_t), see the kernel.
// Loads and stores of a wave retire through one in-order counter, so a load issued after a store cannot be consumed
// before that store has reached memory.  Both kernels therefore fetch the per-instance reference a chunk of kChunk
// steps ahead: the wait for a chunk's loads only covers stores that are a whole chunk old.  A reference shared by the
// batch is read through the scalar cache and needs none of this.
#include "mpcx_internal.h"

namespace mpcx {
bool qp_lane_supports(int n, int m, int r);
namespace {

constexpr int kWave = 64;
constexpr int kChunk = 8;

// FS > 0: the horizon is the compile-time constant FS and Phi and the gain rows live in registers (no LDS traffic in
// the loop); FS = 0: any horizon, Phi and the gain rows in the lane's LDS column.
template <int N, int M, int R, bool SHARED, int FS>
__global__ __launch_bounds__(kWave) void closed_loop_lane_kernel(LoopArgs p, const double *__restrict__ ref_shared) {
    extern __shared__ double lds[];
    const int ln = threadIdx.x;
    const long long b = (long long)blockIdx.x * kWave + ln;
    if (b >= p.batch) return;
    const long long W = p.W, tile = b / W, lane = b - tile * W, T = p.T;
    const int F = FS > 0 ? FS : p.f, FR = F * R, VM = p.v * M;
    constexpr int kRegs = FS > 0 ? FS * R : 1;
    double rPhi[kRegs][N], rK[kRegs][M];
#define IN(ptr, E, e) ((ptr)[tile * (long long)(E) * W + (long long)(e) * W + lane])
    double *sPhi = lds + ln;                                       // element e at sPhi[e * 64]
    double *sK = sPhi + (FS > 0 ? 0 : (size_t)FR * N * kWave);     // (c, k) at sK[(c * M + k) * 64]
    double *sRing = sK + (FS > 0 ? 0 : (size_t)FR * M * kWave);    // column tau, row a at sRing[((tau % RC) * R + a) * 64]
    const int RC = F + kChunk;
    const long long L = T + F, ER = (long long)R * L;

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
    if constexpr (FS > 0) {
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int c = 0; c < kRegs; ++c) rPhi[c][j] = IN(p.Phi, kRegs * N, j * kRegs + c);
#pragma unroll
        for (int c = 0; c < kRegs; ++c)
#pragma unroll
            for (int k = 0; k < M; ++k) rK[c][k] = IN(p.gain, (long long)VM * kRegs, (long long)c * VM + k);
    } else {
        for (int e = 0; e < FR * N; ++e) sPhi[e * kWave] = IN(p.Phi, FR * N, e);
        for (int c = 0; c < FR; ++c)
#pragma unroll
            for (int k = 0; k < M; ++k) sK[(c * M + k) * kWave] = IN(p.gain, (long long)VM * FR, (long long)c * VM + k);
    }

    double pre[kChunk][R];
    if (!SHARED) {
        const long long first = (kChunk + F - 1 < L) ? kChunk + F - 1 : L;
        for (long long tau = 0; tau < first; ++tau)
#pragma unroll
            for (int a = 0; a < R; ++a) sRing[((int)(tau % RC) * R + a) * kWave] = IN(p.ref, ER, tau * R + a);
        // branch-free: a column beyond the trajectory reads the (valid, never used) last column instead
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            const long long tau = kChunk + F - 1 + q, tc = tau < L ? tau : L - 1;
#pragma unroll
            for (int a = 0; a < R; ++a) pre[q][a] = IN(p.ref, ER, tc * R + a);
        }
    }

    const long long EX = (long long)N * (T + 1), EY = (long long)R * T, EU = (long long)M * T;
    if (p.X) {
#pragma unroll
        for (int i = 0; i < N; ++i) IN(p.X, EX, i) = x[i];
    }
    for (long long t = 0; t < T; ++t) {
        if (!SHARED && t > 0 && (t % kChunk) == 0) {
            // columns [t + F - 1, t + F - 1 + kChunk) arrive from the registers filled one chunk ago; the next chunk's
            // loads go out before this chunk's stores
#pragma unroll
            for (int q = 0; q < kChunk; ++q) {
                const long long tau = t + F - 1 + q;
#pragma unroll
                for (int a = 0; a < R; ++a) sRing[((int)(tau % RC) * R + a) * kWave] = pre[q][a];
            }
#pragma unroll
            for (int q = 0; q < kChunk; ++q) {
                const long long tau = t + kChunk + F - 1 + q, tc = tau < L ? tau : L - 1;
#pragma unroll
                for (int a = 0; a < R; ++a) pre[q][a] = IN(p.ref, ER, tc * R + a);
            }
        }
        double u[M];
#pragma unroll
        for (int k = 0; k < M; ++k) u[k] = 0.0;
        int slot = (int)(t % RC);
        if constexpr (FS > 0) {
#pragma unroll
            for (int i = 0; i < FS; ++i) {
#pragma unroll
                for (int a = 0; a < R; ++a) {
                    const int c = i * R + a;
                    double ox = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) ox += rPhi[c][j] * x[j];
                    const double rv = SHARED ? ref_shared[(t + i) * R + a] : sRing[(slot * R + a) * kWave];
                    const double s = rv - ox;
#pragma unroll
                    for (int k = 0; k < M; ++k) u[k] += rK[c][k] * s;
                }
                slot = (slot + 1 == RC) ? 0 : slot + 1;
            }
        } else {
#pragma unroll 4
            for (int i = 0; i < F; ++i) {
#pragma unroll
                for (int a = 0; a < R; ++a) {
                    const int c = i * R + a;
                    double ox = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) ox += sPhi[(j * FR + c) * kWave] * x[j];
                    const double rv = SHARED ? ref_shared[(t + i) * R + a] : sRing[(slot * R + a) * kWave];
                    const double s = rv - ox;
#pragma unroll
                    for (int k = 0; k < M; ++k) u[k] += sK[(c * M + k) * kWave] * s;
                }
                slot = (slot + 1 == RC) ? 0 : slot + 1;
            }
        }
        if (p.U) {
#pragma unroll
            for (int k = 0; k < M; ++k) IN(p.U, EU, t * M + k) = u[k];
        }
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

// horizons with a register-resident instantiation (the reference scenario's f = 20, main.rs:17)
template <int N, int M, int R>
constexpr int static_horizon() { return (M == 1 && R == 1 && (N == 2 || N == 4)) ? 20 : 0; }

size_t lane_lds_bytes(int n, int m, int r, int f, bool shared, bool in_regs) {
    const size_t per_lane = (in_regs ? 0 : (size_t)f * r * (n + m)) + (shared ? 0 : (size_t)(f + kChunk) * r);
    return per_lane * kWave * sizeof(double);
}

template <int N, int M, int R, bool SHARED, int FS>
int launch_lane_inst(const LoopArgs &a, hipStream_t s) {
    const long long blocks = (a.batch + kWave - 1) / kWave;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    const size_t lds_bytes = lane_lds_bytes(N, M, R, a.f, SHARED, FS > 0);
    if (lds_bytes > 160 * 1024) return MPCX_ERR_UNSUPPORTED;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)closed_loop_lane_kernel<N, M, R, SHARED, FS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((closed_loop_lane_kernel<N, M, R, SHARED, FS>), dim3((unsigned)blocks), dim3(kWave), lds_bytes, s, a, a.ref);
    return (int)hipGetLastError();
}

template <int N, int M, int R>
int launch_lane_one(const LoopArgs &a, hipStream_t s) {
    constexpr int FS = static_horizon<N, M, R>();
    if constexpr (FS > 0) {
        if (a.f == FS) return a.ref_shared ? launch_lane_inst<N, M, R, true, FS>(a, s) : launch_lane_inst<N, M, R, false, FS>(a, s);
    }
    return a.ref_shared ? launch_lane_inst<N, M, R, true, 0>(a, s) : launch_lane_inst<N, M, R, false, 0>(a, s);
}

// One workgroup of four wavefronts per instance.  LDS: A, B, C, Phi, the first m gain rows, two state buffers, s, u,
// per-wavefront partial sums, the reference ring and a staging area holding kChunk steps of outputs.
// The last wavefront alone fetches the reference (a chunk ahead, into registers, then into the ring) and takes no part
// in flushing the outputs: its memory counter then only ever holds loads, so waiting for them never waits for a store.
constexpr int kGroupThreads = 256;
constexpr int kGroupWaves = kGroupThreads / kWave;
constexpr int kRowsPerPass = 4;  // rows of s a thread carries at once

__global__ __launch_bounds__(kGroupThreads) void closed_loop_group_kernel(LoopArgs p) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long b = blockIdx.x;
    const int n = p.n, m = p.m, r = p.r, F = p.f, FR = F * r, VM = p.v * m;
    const long long W = p.W, tile = b / W, li = b - tile * W, T = p.T;
    const long long L = T + F, ER = (long long)r * L;
    const int RC = F + kChunk;
#define IN(ptr, E, e) ((ptr)[tile * (long long)(E) * W + (long long)(e) * W + li])
    double *sA = lds, *sB = sA + n * n, *sC = sB + n * m, *sPhi = sC + r * n, *sK = sPhi + (size_t)FR * n;
    double *sx = sK + (size_t)FR * m, *ss = sx + 2 * n, *su = ss + FR, *sPart = su + m, *sOut = sPart + kGroupWaves * 4;
    const int out_per_step = n + m + r;
    double *sRing = sOut + kChunk * out_per_step;  // (tau % RC) * r + a
    for (int e = tid; e < n * n; e += kGroupThreads) sA[e] = IN(p.A, n * n, e);
    for (int e = tid; e < n * m; e += kGroupThreads) sB[e] = IN(p.B, n * m, e);
    for (int e = tid; e < r * n; e += kGroupThreads) sC[e] = IN(p.C, r * n, e);
    for (int e = tid; e < FR * n; e += kGroupThreads) sPhi[e] = IN(p.Phi, (long long)FR * n, e);
    for (int e = tid; e < FR * m; e += kGroupThreads) {  // sK[c * m + k] = gain(k, c)
        const int c = e / m, k = e - c * m;
        sK[e] = IN(p.gain, (long long)VM * FR, (long long)c * VM + k);
    }
    const long long EX = (long long)n * (T + 1), EY = (long long)r * T, EU = (long long)m * T;
    if (tid < n) {
        const double x0 = IN(p.x0, n, tid);
        sx[tid] = x0;
        if (p.X) IN(p.X, EX, tid) = x0;
    }
    // reference: element (a, tau) at rbase[tau * rstep] for lane a of the last wavefront
    const bool loader = wv == kGroupWaves - 1;
    const int la = lane < r ? lane : 0;
    const double *rbase = p.ref_shared ? p.ref + la : &IN(p.ref, ER, la);
    const long long rstep = p.ref_shared ? (long long)r : (long long)r * W;
    double pre[kChunk];
    if (loader) {
        const long long first = (kChunk + F - 1 < L) ? kChunk + F - 1 : L;
        for (long long tau = 0; tau < first; ++tau)
            if (lane < r) sRing[(int)(tau % RC) * r + lane] = rbase[tau * rstep];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {  // branch-free, clamped to valid (unused) addresses
            const long long tau = kChunk + F - 1 + q;
            pre[q] = rbase[(tau < L ? tau : L - 1) * rstep];
        }
    }
    __syncthreads();
    for (long long t0 = 0; t0 < T; t0 += kChunk) {
        const int steps = (int)((T - t0 < kChunk) ? T - t0 : kChunk);
        if (loader && t0 > 0) {
#pragma unroll
            for (int q = 0; q < kChunk; ++q) {
                const long long tau = t0 + F - 1 + q;
                if (lane < r) sRing[(int)(tau % RC) * r + lane] = pre[q];
            }
#pragma unroll
            for (int q = 0; q < kChunk; ++q) {
                const long long tau = t0 + kChunk + F - 1 + q;
                pre[q] = rbase[(tau < L ? tau : L - 1) * rstep];
            }
        }
        lds_barrier();
        for (int q = 0; q < steps; ++q) {
            const long long t = t0 + q;
            const int tmod = (int)(t % RC);
            const double *xc = sx + (t & 1) * n;
            double *xn = sx + ((t + 1) & 1) * n;
            double *out = sOut + q * out_per_step;  // [x_{t+1} (n) | u_t (m) | y_t (r)]
            // s = ref window - Phi x: threads along the rows of s, kRowsPerPass rows each per pass, the state read once
            // per pass; the partial sums of u = K s stay in the thread
            double uacc[4] = {0.0, 0.0, 0.0, 0.0};
            for (int c0 = tid; c0 < FR; c0 += kGroupThreads * kRowsPerPass) {
                double ox[kRowsPerPass];
                int cc[kRowsPerPass];
#pragma unroll
                for (int g = 0; g < kRowsPerPass; ++g) {
                    ox[g] = 0.0;
                    const int c = c0 + g * kGroupThreads;
                    cc[g] = c < FR ? c : FR - 1;  // clamped: surplus rows are computed and dropped
                }
#pragma unroll 4
                for (int j = 0; j < n; ++j) {
                    const double xj = xc[j];
#pragma unroll
                    for (int g = 0; g < kRowsPerPass; ++g) ox[g] += sPhi[j * FR + cc[g]] * xj;
                }
#pragma unroll
                for (int g = 0; g < kRowsPerPass; ++g) {
                    const int c = cc[g];
                    const bool live = c0 + g * kGroupThreads < FR;
                    // row c = i r + a of the window is column t + i, row a of the trajectory: ring index (t mod RC) r + c
                    int ri = tmod * r + c;
                    if (ri >= RC * r) ri -= RC * r;
                    const double sv = live ? sRing[ri] - ox[g] : 0.0;
                    if (m <= 4) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) uacc[k] += sK[c * m + (k < m ? k : 0)] * (k < m ? sv : 0.0);
                    } else if (live) {
                        ss[c] = sv;
                    }
                }
            }
            if (m <= 4) {
                // the four sums go down the tree together: four permutes in flight per level instead of one
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    double o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = __shfl_xor(uacc[k], off);
#pragma unroll
                    for (int k = 0; k < 4; ++k) uacc[k] += o[k];
                }
                if (lane < 4) sPart[wv * 4 + lane] = lane == 0 ? uacc[0] : lane == 1 ? uacc[1] : lane == 2 ? uacc[2] : uacc[3];
                lds_barrier();
                if (tid < m) {
                    double v = 0.0;
#pragma unroll
                    for (int w = 0; w < kGroupWaves; ++w) v += sPart[w * 4 + tid];
                    su[tid] = v;
                    out[n + tid] = v;
                }
            } else {
                lds_barrier();
                if (tid < m) {
                    double v = 0.0;
                    for (int c = 0; c < FR; ++c) v += sK[c * m + tid] * ss[c];
                    su[tid] = v;
                    out[n + tid] = v;
                }
            }
            lds_barrier();
            if (tid < n) {
                double sacc = 0.0;
#pragma unroll 4
                for (int j = 0; j < n; ++j) sacc += sA[j * n + tid] * xc[j];
                double qv = 0.0;
#pragma unroll 4
                for (int k = 0; k < m; ++k) qv += sB[k * n + tid] * su[k];
                sacc += qv;
                xn[tid] = sacc;
                out[tid] = sacc;
            } else if (tid >= kWave && tid < kWave + r) {  // outputs on the second wavefront
                const int a = tid - kWave;
                double y = 0.0;
#pragma unroll 4
                for (int j = 0; j < n; ++j) y += sC[j * r + a] * xc[j];
                out[n + m + a] = y;
            }
            lds_barrier();
        }
        // flush the chunk: X columns t0+1 .. t0+steps are contiguous per instance, likewise U and Y
        if (!loader) {
            constexpr int kFlush = kGroupThreads - kWave;
            if (p.X)
                for (int e = tid; e < steps * n; e += kFlush) IN(p.X, EX, (t0 + 1) * n + e) = sOut[(e / n) * out_per_step + (e % n)];
            if (p.U)
                for (int e = tid; e < steps * m; e += kFlush) IN(p.U, EU, t0 * m + e) = sOut[(e / m) * out_per_step + n + (e % m)];
            if (p.Y)
                for (int e = tid; e < steps * r; e += kFlush) IN(p.Y, EY, t0 * r + e) = sOut[(e / r) * out_per_step + n + m + (e % r)];
        }
        lds_barrier();
    }
#undef IN
}

// Register-resident variant of the group kernel for at most four inputs and f r <= 3 x 256 window rows: thread t owns
// rows t, t + 256, t + 512 of the window, with their rows of Phi (n values each) and their columns of the first m gain
// rows in registers for the whole loop.  A step is then: read the state (LDS broadcast), s_c = ref_c - Phi_c x,
// partial u = K s over the thread's rows, a wavefront reduction, one barrier, and the propagation by n threads that
// each add up the four wavefronts' partial sums themselves; a second barrier publishes the new state.  Nothing of
// Phi or the gain lives in LDS (about 10 KB per instance instead of 80), so several workgroups share a CU and hide
// one another's barriers.  The arithmetic order of the reference is kept: s first, then gain s.
template <int NP, int RPT>
__global__ __launch_bounds__(kGroupThreads, NP >= 8 ? 2 : 3) void closed_loop_rows_kernel(LoopArgs p) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long b = blockIdx.x;
    const int n = p.n, m = p.m, r = p.r, F = p.f, FR = F * r, VM = p.v * m;
    const long long W = p.W, tile = b / W, li = b - tile * W, T = p.T;
    const long long L = T + F, ER = (long long)r * L;
    const int RC = F + kChunk;
#define IN(ptr, E, e) ((ptr)[tile * (long long)(E) * W + (long long)(e) * W + li])
    double *sA = lds, *sB = sA + n * n, *sC = sB + n * m, *sx = sC + r * n, *sPart = sx + 2 * n;
    double *sOut = sPart + 2 * kGroupWaves * 4;
    const int out_per_step = n + m + r;
    double *sRing = sOut + kChunk * out_per_step;  // (tau % RC) * r + a
    for (int e = tid; e < n * n; e += kGroupThreads) sA[e] = IN(p.A, n * n, e);
    for (int e = tid; e < n * m; e += kGroupThreads) sB[e] = IN(p.B, n * m, e);
    for (int e = tid; e < r * n; e += kGroupThreads) sC[e] = IN(p.C, r * n, e);
    double phi[RPT][NP], kk[RPT][4];
    int rowc[RPT];
    bool live[RPT];
#pragma unroll
    for (int g = 0; g < RPT; ++g) {
        const int c = tid + g * kGroupThreads;
        live[g] = c < FR;
        rowc[g] = live[g] ? c : FR - 1;
#pragma unroll
        for (int j = 0; j < NP; ++j)
            phi[g][j] = (live[g] && j < n) ? IN(p.Phi, (long long)FR * n, (long long)(j < n ? j : 0) * FR + rowc[g]) : 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            kk[g][k] = (live[g] && k < m) ? IN(p.gain, (long long)VM * FR, (long long)rowc[g] * VM + (k < m ? k : 0)) : 0.0;
    }
    const long long EX = (long long)n * (T + 1), EY = (long long)r * T, EU = (long long)m * T;
    if (tid < n) {
        const double x0 = IN(p.x0, n, tid);
        sx[tid] = x0;
        if (p.X) IN(p.X, EX, tid) = x0;
    }
    // reference: element (a, tau) at rbase[tau * rstep] for lane a of the last wavefront
    const bool loader = wv == kGroupWaves - 1;
    const int la = lane < r ? lane : 0;
    const double *rbase = p.ref_shared ? p.ref + la : &IN(p.ref, ER, la);
    const long long rstep = p.ref_shared ? (long long)r : (long long)r * W;
    double pre[kChunk];
    if (loader) {
        const long long first = (kChunk + F - 1 < L) ? kChunk + F - 1 : L;
        for (long long tau = 0; tau < first; ++tau)
            if (lane < r) sRing[(int)(tau % RC) * r + lane] = rbase[tau * rstep];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {  // branch-free, clamped to valid (unused) addresses
            const long long tau = kChunk + F - 1 + q;
            pre[q] = rbase[(tau < L ? tau : L - 1) * rstep];
        }
    }
    __syncthreads();
    for (long long t0 = 0; t0 < T; t0 += kChunk) {
        const int steps = (int)((T - t0 < kChunk) ? T - t0 : kChunk);
        if (loader && t0 > 0) {
#pragma unroll
            for (int q = 0; q < kChunk; ++q) {
                const long long tau = t0 + F - 1 + q;
                if (lane < r) sRing[(int)(tau % RC) * r + lane] = pre[q];
            }
#pragma unroll
            for (int q = 0; q < kChunk; ++q) {
                const long long tau = t0 + kChunk + F - 1 + q;
                pre[q] = rbase[(tau < L ? tau : L - 1) * rstep];
            }
        }
        lds_barrier();
        for (int q = 0; q < steps; ++q) {
            const long long t = t0 + q;
            const int tmod = (int)(t % RC);
            const double *xc = sx + (t & 1) * n;
            double *xn = sx + ((t + 1) & 1) * n;
            double *part = sPart + (t & 1) * kGroupWaves * 4;
            double *out = sOut + q * out_per_step;  // [x_{t+1} (n) | u_t (m) | y_t (r)]
            double x[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) x[j] = xc[j < n ? j : n - 1];  // phi is zero past n
            double uacc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int g = 0; g < RPT; ++g) {
                // row c = i r + a of the window is column t + i, row a of the trajectory: ring index (t mod RC) r + c
                int ri = tmod * r + rowc[g];
                if (ri >= RC * r) ri -= RC * r;
                const double refv = sRing[ri];
                double ox = 0.0;
#pragma unroll
                for (int j = 0; j < NP; ++j) ox += phi[g][j] * x[j];
                const double sv = live[g] ? refv - ox : 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) uacc[k] += kk[g][k] * sv;
            }
            // Wavefront reduction of the four sums as a reduce-scatter: after two exchange levels a lane carries one of the
            // four sums for its group of four lanes (7 exchanges of a double instead of 24, five of them register moves).  Lane l ends up with sum
            // 2 (l & 1) + ((l >> 1) & 1).
            {
                const bool b0 = lane & 1, b1 = lane & 2;
                const double s0 = b0 ? uacc[0] : uacc[2], s1 = b0 ? uacc[1] : uacc[3];
                double k0 = b0 ? uacc[2] : uacc[0], k1 = b0 ? uacc[3] : uacc[1];
                k0 += dpp_f64<0xB1>(s0);  // lane ^ 1 (quad permutation 1,0,3,2)
                k1 += dpp_f64<0xB1>(s1);
                double mine = (b1 ? k1 : k0) + dpp_f64<0x4E>(b1 ? k0 : k1);  // lane ^ 2 (2,3,0,1)
                mine += dpp_f64<0x124>(mine);  // rotate by 4 within the row of 16 lanes
                mine += dpp_f64<0x128>(mine);  // rotate by 8: every lane of a row now holds its row's sum
                mine += __shfl_xor(mine, 16);
                mine += __shfl_xor(mine, 32);
                if (lane < 4) part[wv * 4 + 2 * (lane & 1) + ((lane >> 1) & 1)] = mine;
            }
            lds_barrier();
            if (tid < n) {
                double sacc = 0.0;
#pragma unroll
                for (int j = 0; j < NP; ++j) sacc += sA[(j < n ? j : 0) * n + tid] * (j < n ? x[j] : 0.0);
                double qv = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double u = 0.0;
#pragma unroll
                    for (int w = 0; w < kGroupWaves; ++w) u += part[w * 4 + k];
                    qv += (k < m ? sB[k * n + tid] : 0.0) * u;
                }
                sacc += qv;
                xn[tid] = sacc;
                out[tid] = sacc;
            } else if (tid >= kWave && tid < kWave + r) {  // outputs on the second wavefront
                const int a = tid - kWave;
                double y = 0.0;
#pragma unroll
                for (int j = 0; j < NP; ++j) y += sC[(j < n ? j : 0) * r + a] * (j < n ? x[j] : 0.0);
                out[n + m + a] = y;
            } else if (tid >= 2 * kWave && tid < 2 * kWave + m) {  // the move itself on the third
                const int k = tid - 2 * kWave;
                double u = 0.0;
#pragma unroll
                for (int w = 0; w < kGroupWaves; ++w) u += part[w * 4 + k];
                out[n + k] = u;
            }
            lds_barrier();
        }
        // flush the chunk: X columns t0+1 .. t0+steps are contiguous per instance, likewise U and Y
        if (!loader) {
            constexpr int kFlush = kGroupThreads - kWave;
            if (p.X)
                for (int e = tid; e < steps * n; e += kFlush) IN(p.X, EX, (t0 + 1) * n + e) = sOut[(e / n) * out_per_step + (e % n)];
            if (p.U)
                for (int e = tid; e < steps * m; e += kFlush) IN(p.U, EU, t0 * m + e) = sOut[(e / m) * out_per_step + n + (e % m)];
            if (p.Y)
                for (int e = tid; e < steps * r; e += kFlush) IN(p.Y, EY, t0 * r + e) = sOut[(e / r) * out_per_step + n + m + (e % r)];
        }
        lds_barrier();
    }
#undef IN
}

// The loop with the window products taken out of it.  u_t = K (w_t - Phi x_t) with w_t the reference window of step t is
// computed as u_t = K w_t - (K Phi) x_t: the m x n matrix K Phi is formed once per instance, and the products
// v_t = K w_t do not depend on the state, so all T of them are formed in parallel ahead of the serial recursion -- one
// lane per step, K shared through LDS -- instead of f r multiply-adds and a workgroup reduction inside every step.
// What stays serial is u_t = v_t - (K Phi) x_t (m x n), y_t = C x_t and x_{t+1} = A x_t + B u_t, run by one wavefront with
// the state broadcast lane to lane (no LDS, no barrier in the recursion).  Same trajectory as compute_control_input
// followed by propagate_dynamics (controller.rs:119-132, 101-115) up to the rounding of the regrouped product (the
// bound is that of K s itself: |K| |w| eps); A x + B u and C x keep the reference's order.
// Steps go in chunks of 64: reference columns -> LDS (transposed: a lane walks its window with unit stride), the four
// wavefronts each take a quarter of the window rows for all 64 steps, their partial sums are added, one wavefront runs
// the 64 steps into an LDS staging area, all flush it.  m <= 4, n <= 12, r <= 16.
__device__ __forceinline__ double lane_bcast(double x, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src), hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}
constexpr int kAffChunk = 64;
template <int NP>
__global__ __launch_bounds__(kGroupThreads, 3) void closed_loop_affine_kernel(LoopArgs p) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long b = blockIdx.x;
    const int n = p.n, m = p.m, r = p.r, F = p.f, FR = F * r, VM = p.v * m;
    const long long W = p.W, tile = b / W, li = b - tile * W, T = p.T;
    const long long L = T + F, ER = (long long)r * L;
    constexpr int TC = kAffChunk;
    const int RL = TC + F - 1, ops = n + m + r;  // reference columns a chunk touches; doubles a step leaves behind
    // The window of step tl is the run of F columns from column tl on: with the columns stored one after the other, row
    // c' of every window is at a fixed distance from the window's start and the products are one flat loop.  A column takes
    // rp = r | 1 doubles (an odd stride keeps the lanes, one per step, on distinct LDS banks); the pad entry of a column
    // is zero in the reference and in K.
    const int rp = r | 1, FP = F * rp;
#define IN(ptr, E, e) ((ptr)[tile * (long long)(E) * W + (long long)(e) * W + li])
    double *sA = lds, *sB = sA + n * n, *sC = sB + n * m, *sKP = sC + r * n;  // sKP[k * NP + j] = (K Phi)[k][j]
    double *sK = sKP + 4 * NP;                                                // sK[(i rp + a) * 4 + k] = K[k][i r + a] (zero for k >= m, a = r)
    double *sRef = sK + (size_t)FP * 4;                                       // sRef[(tau - t0) * rp + a]
    double *sV = sRef + (size_t)rp * RL;                                      // sV[tl * 4 + k]
    double *sOut = sV + TC * 4;                                               // [tl][x_{t+1} (n) | u_t (m) | y_t (r)]; also the partial sums
    double *sPart = sOut;                                                     // sPart[(w * TC + tl) * 4 + k], and K Phi's partial sums
    for (int e = tid; e < n * n; e += kGroupThreads) sA[e] = IN(p.A, n * n, e);
    for (int e = tid; e < n * m; e += kGroupThreads) sB[e] = IN(p.B, n * m, e);
    for (int e = tid; e < r * n; e += kGroupThreads) sC[e] = IN(p.C, r * n, e);
    for (int e = tid; e < FP * 4; e += kGroupThreads) {
        const int cp = e >> 2, k = e & 3, i = cp / rp, a = cp - i * rp;
        sK[e] = (k < m && a < r) ? IN(p.gain, (long long)VM * FR, (long long)(i * r + a) * VM + k) : 0.0;
    }
    if (rp > r)
        for (int e = tid; e < RL; e += kGroupThreads) sRef[e * rp + r] = 0.0;  // the pad entries
    const long long EX = (long long)n * (T + 1), EY = (long long)r * T, EU = (long long)m * T;
    // the reference columns of a chunk, requested a chunk ahead (element e of the chunk: column t0 + e / r, row e % r)
    constexpr int kPre = 6;  // registers per thread; chunks with more columns load the rest in place
    const int nref = r * RL;
    double pre[kPre];
    auto request_ref = [&](long long t0) {
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const int e = tid + q * kGroupThreads;
            long long tau = t0 + e / r;
            if (e >= nref || tau >= L) tau = L - 1;  // (a valid address, unused)
            const long long idx = tau * r + e % r;
            pre[q] = p.ref_shared ? p.ref[idx] : IN(p.ref, ER, idx);
        }
    };
    request_ref(0);
    __syncthreads();
    // Diagnostic build (-DMPCX_LOOP_STAMPS, tools/probe_loop.py): thread 0 adds up the time of each phase and leaves the sums
    // at the start of its instance's X.
#ifdef MPCX_LOOP_STAMPS
    long long ls_last = wall_clock64();
    double ls_sum[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#define LSTAMP(k) do { const long long now_ = wall_clock64(); ls_sum[k] += (double)(now_ - ls_last); ls_last = now_; } while (0)
#else
#define LSTAMP(k) do {} while (0)
#endif
    // ---- K Phi: a wavefront takes NP / 4 columns of Phi, lanes along the window rows (Phi read once, coalesced) ----
    {
        constexpr int NPW = NP / 4;
        double acc[4][NPW];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int jj = 0; jj < NPW; ++jj) acc[k][jj] = 0.0;
        for (int c = lane; c < FR; c += kWave) {
            double ph[NPW], kv[4];
#pragma unroll
            for (int jj = 0; jj < NPW; ++jj) {
                const int j = wv * NPW + jj;
                ph[jj] = j < n ? IN(p.Phi, (long long)FR * n, (long long)j * FR + c) : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) kv[k] = k < m ? IN(p.gain, (long long)VM * FR, (long long)c * VM + k) : 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int jj = 0; jj < NPW; ++jj) acc[k][jj] += kv[k] * ph[jj];
        }
        // rows of 16 lanes by lane-to-lane moves, the wavefront's four row sums through LDS
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int jj = 0; jj < NPW; ++jj) {
                double v = acc[k][jj];
                v += dpp_f64<0xB1>(v);
                v += dpp_f64<0x4E>(v);
                v += dpp_f64<0x124>(v);
                v += dpp_f64<0x128>(v);
                if ((lane & 15) == 0) sPart[(lane >> 4) * 4 * NP + k * NP + wv * NPW + jj] = v;
            }
        __syncthreads();
        if (tid < 4 * NP) sKP[tid] = (sPart[tid] + sPart[4 * NP + tid]) + (sPart[8 * NP + tid] + sPart[12 * NP + tid]);
        __syncthreads();
    }
    LSTAMP(0);
    // ---- the serial wavefront: lane l holds x_l (and, inside a chunk, row l of A, B, C and of K Phi) ----
    // Which wavefront runs the recursion differs from workgroup to workgroup (a hash of the workgroup's index): the
    // workgroups that share a CU would otherwise all run it on the same SIMD, one behind the other.
    const int sw = (int)((blockIdx.x * 2654435761u) >> 30);
    double xj = 0.0;
    if (wv == sw && lane < n) {
        xj = IN(p.x0, n, lane);
        if (p.X) IN(p.X, EX, lane) = xj;
    }
    for (long long t0 = 0; t0 < T; t0 += TC) {
        const int steps = (int)((T - t0 < TC) ? T - t0 : TC);
        // reference columns t0 .. t0 + steps + F - 2, transposed
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const int e = tid + q * kGroupThreads;
            if (e < nref) sRef[(e / r) * rp + e % r] = pre[q];
        }
        for (int e = tid + kPre * kGroupThreads; e < nref; e += kGroupThreads) {
            long long tau = t0 + e / r;
            if (tau >= L) tau = L - 1;
            const long long idx = tau * r + e % r;
            sRef[(e / r) * rp + e % r] = p.ref_shared ? p.ref[idx] : IN(p.ref, ER, idx);
        }
        __syncthreads();
        LSTAMP(1);
        // v_t = K w_t: lane tl is step t0 + tl, this wavefront's quarter of the (padded) window rows
        {
            const int c0 = (FP * wv) / kGroupWaves, c1 = (FP * (wv + 1)) / kGroupWaves;
            double v[4] = {0.0, 0.0, 0.0, 0.0};
            const double *rl = sRef + lane * rp;
            constexpr int UN = 8;  // rows per pass: every LDS read of the pass is issued before its multiply-adds
            int c = c0;
            for (; c + UN <= c1; c += UN) {
                double w[UN], kv[UN][4];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    w[u] = rl[c + u];
#pragma unroll
                    for (int k = 0; k < 4; ++k) kv[u][k] = sK[(c + u) * 4 + k];
                }
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += kv[u][k] * w[u];
            }
            for (; c < c1; ++c) {
                const double w = rl[c];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] += sK[c * 4 + k] * w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) sPart[(wv * TC + lane) * 4 + k] = v[k];
        }
        __syncthreads();
        LSTAMP(2);
        {
            double v = 0.0;  // thread (tl, k)
#pragma unroll
            for (int w = 0; w < kGroupWaves; ++w) v += sPart[w * TC * 4 + tid];
            sV[tid] = v;
        }
        __syncthreads();
        if (t0 + TC < T) request_ref(t0 + TC);  // (ahead of this chunk's stores)
        LSTAMP(3);
        if (wv == sw) {
            // Lanes 0..15 hold the rows of A (and of B), lanes 16..31 the rows of C, lanes 32..35 the rows of K Phi: a step
            // is one dot product of length n per lane, all three families side by side.  x_j lives in lane j.
            const int grp = lane >> 4, l15 = lane & 15;
            double row[NP], rowB[4];  // (read here, chunk by chunk: not live across the window products)
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                double v = 0.0;
                if (j < n) {
                    if (grp == 0 && l15 < n) v = sA[j * n + l15];
                    else if (grp == 1 && l15 < r) v = sC[j * r + l15];
                    else if (grp == 2 && l15 < 4) v = sKP[l15 * NP + j];
                }
                row[j] = v;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) rowB[k] = (lane < n && k < m) ? sB[k * n + lane] : 0.0;
            const int vk = (grp == 2 && l15 < 4) ? l15 : 0;
            // where this lane's result of a step goes ([x_{t+1} | u_t | y_t] of the step; idle lanes share a spare slot)
            const int slot = lane < n ? lane : (grp == 1 && l15 < r) ? n + m + l15 : (grp == 2 && l15 < m) ? n + l15 : -1;
            double vn = sV[vk];  // v of the first step, lanes 32 + k
            for (int q = 0; q < steps; ++q) {
                const double vt = vn;
                vn = sV[((q + 1 < steps) ? q + 1 : q) * 4 + vk];
                double xs[NP];  // the state, every entry in scalar registers before the first product
#pragma unroll
                for (int j = 0; j < NP; ++j) xs[j] = lane_bcast(xj, j);
                double dot = 0.0;  // lanes < 16: (A x)_l; 16..31: y_l = (C x)_l; 32..35: ((K Phi) x)_k
#pragma unroll
                for (int j = 0; j < NP; ++j) dot += row[j] * xs[j];
                const double uk = vt - dot;  // lanes 32 + k: the move
                double us[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) us[k] = lane_bcast(uk, 32 + k);
                double bu = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) bu += rowB[k] * us[k];
                const double xnew = dot + bu;
                if (lane < n) xj = xnew;
                const double res = grp == 0 ? xnew : grp == 1 ? dot : uk;
                sOut[slot >= 0 ? q * ops + slot : TC * ops] = res;
            }
        }
        __syncthreads();
        LSTAMP(4);
        // flush the chunk: X columns t0+1 .. t0+steps are contiguous per instance, likewise U and Y
        if (p.X)
            for (int e = tid; e < steps * n; e += kGroupThreads) IN(p.X, EX, (t0 + 1) * n + e) = sOut[(e / n) * ops + (e % n)];
        if (p.U)
            for (int e = tid; e < steps * m; e += kGroupThreads) IN(p.U, EU, t0 * m + e) = sOut[(e / m) * ops + n + (e % m)];
        if (p.Y)
            for (int e = tid; e < steps * r; e += kGroupThreads) IN(p.Y, EY, t0 * r + e) = sOut[(e / r) * ops + n + m + (e % r)];
        __syncthreads();
        LSTAMP(5);
    }
#ifdef MPCX_LOOP_STAMPS
    if (tid == 0 && p.X)
        for (int k = 0; k < 6; ++k) IN(p.X, EX, k) = ls_sum[k];
#endif
#undef IN
}

size_t affine_lds_bytes(const LoopArgs &a, int np) {
    const size_t out = (size_t)kAffChunk * (a.n + a.m + a.r) + 1, part = (size_t)kGroupWaves * kAffChunk * 4, kp = (size_t)4 * 4 * np;
    size_t stage = out > part ? out : part;
    if (kp > stage) stage = kp;
    const size_t rp = (size_t)(a.r | 1);
    size_t d = (size_t)a.n * a.n + (size_t)a.n * a.m + (size_t)a.r * a.n + 4 * np + (size_t)a.f * rp * 4 +
               rp * (kAffChunk + a.f - 1) + kAffChunk * 4 + stage;
    return d * sizeof(double);
}
template <int NP>
int launch_affine(const LoopArgs &a, hipStream_t s) {
    const size_t lds_bytes = affine_lds_bytes(a, NP);
    hipLaunchKernelGGL((closed_loop_affine_kernel<NP>), dim3((unsigned)a.batch), dim3(kGroupThreads), lds_bytes, s, a);
    return (int)hipGetLastError();
}

size_t rows_lds_bytes(const LoopArgs &a) {
    size_t d = (size_t)a.n * a.n + (size_t)a.n * a.m + (size_t)a.r * a.n + 2 * a.n + 2 * kGroupWaves * 4 +
               (size_t)kChunk * (a.n + a.m + a.r) + (size_t)(a.f + kChunk) * a.r;
    return d * sizeof(double);
}

template <int NP, int RPT>
int launch_rows_inst(const LoopArgs &a, hipStream_t s) {
    const size_t lds_bytes = rows_lds_bytes(a);
    if (lds_bytes > 64 * 1024) return MPCX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((closed_loop_rows_kernel<NP, RPT>), dim3((unsigned)a.batch), dim3(kGroupThreads), lds_bytes, s, a);
    return (int)hipGetLastError();
}
template <int NP>
int launch_rows_np(const LoopArgs &a, hipStream_t s) {
    const int rpt = (a.f * a.r + kGroupThreads - 1) / kGroupThreads;
    return rpt <= 1 ? launch_rows_inst<NP, 1>(a, s) : rpt == 2 ? launch_rows_inst<NP, 2>(a, s) : launch_rows_inst<NP, 3>(a, s);
}

size_t group_lds_bytes(const LoopArgs &a) {
    const size_t fr = (size_t)a.f * a.r;
    size_t d = (size_t)a.n * a.n + (size_t)a.n * a.m + (size_t)a.r * a.n + fr * a.n + fr * a.m + 2 * a.n + fr + a.m +
               kGroupWaves * 4 + (size_t)kChunk * (a.n + a.m + a.r) + (size_t)(a.f + kChunk) * a.r;
    return d * sizeof(double);
}

}  // namespace

int launch_closed_loop_lane(const LoopArgs &a, hipStream_t s) {
    if (!qp_lane_supports(a.n, a.m, a.r)) return MPCX_ERR_UNSUPPORTED;
#define CASE(N_, M_, R_) \
    if (a.n == N_ && a.m == M_ && a.r == R_) return launch_lane_one<N_, M_, R_>(a, s);
    CASE(1, 1, 1) CASE(2, 1, 1) CASE(3, 1, 1) CASE(4, 1, 1) CASE(5, 1, 1) CASE(6, 1, 1)
    CASE(2, 2, 2) CASE(4, 2, 2) CASE(6, 2, 2)
    CASE(3, 2, 1) CASE(4, 2, 1) CASE(3, 1, 2) CASE(4, 1, 2)
#undef CASE
    return MPCX_ERR_UNSUPPORTED;
}

int launch_closed_loop_group(const LoopArgs &a, hipStream_t s) {
    if (a.n > kWave || a.m > kWave || a.r > kWave) return MPCX_ERR_UNSUPPORTED;
    if (a.batch > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    // few inputs, a state of at most 16 entries: the window products leave the loop
#ifndef MPCX_NO_AFFINE
    if (a.m <= 4 && a.n <= 12 && a.r <= 16 && affine_lds_bytes(a, 12) <= 64 * 1024) {
        if (a.n <= 4) return launch_affine<4>(a, s);
        if (a.n <= 8) return launch_affine<8>(a, s);
        return launch_affine<12>(a, s);
    }
#endif
    // few inputs and a window of at most 768 rows: Phi and the gain rows live in registers
    if (a.m <= 4 && a.n <= 16 && (long long)a.f * a.r <= 3 * kGroupThreads && rows_lds_bytes(a) <= 64 * 1024) {
        if (a.n <= 4) return launch_rows_np<4>(a, s);
        if (a.n <= 8) return launch_rows_np<8>(a, s);
        if (a.n <= 12) return launch_rows_np<12>(a, s);
        return launch_rows_np<16>(a, s);
    }
    const size_t lds_bytes = group_lds_bytes(a);
    if (lds_bytes > 160 * 1024) return MPCX_ERR_UNSUPPORTED;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)closed_loop_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(closed_loop_group_kernel, dim3((unsigned)a.batch), dim3(kGroupThreads), lds_bytes, s, a);
    return (int)hipGetLastError();
}

}  // namespace mpcx
