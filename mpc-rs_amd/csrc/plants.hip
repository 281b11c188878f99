// Plant construction on the device: continuous-time matrices of the reference's two plants from per-instance
// parameters and the backward-Euler discretisation, batched.
//
// Reference: src/control/model.rs:47-58 (dc_motor::Model::new) and :128-141 (mpc::Model::new):
//     A = (I - Ac dt)^-1,   B = A (Bc dt),   C = Cc.
// The reference inverts with nalgebra's try_inverse and unwraps; here an instance whose matrix cannot be inverted
// (zero or non-finite pivot) gets info = 1 and NaN outputs, and the batch goes on.
//
//   plants_kernel<KIND>   one lane per instance: parameters -> Ac, Bc in registers -> A, B, C (n = 2 or 4, m = r = 1)
//   c2d_lane_kernel<N,M>  one lane per instance for the tiny shapes the lane QP kernel serves
//   general n             M = I - Ac dt is formed elementwise, inverted by the batched solver (right-hand side I, in
//                         place in the output A) and B = A (Bc dt) follows elementwise
#include "mpcx_internal.h"
#include "small_inverse.h"

#include <mutex>

// Scratch of the allocation-free discretisation (mpcx_c2d_backward_euler_f64_noalloc_dev): I - Ac dt of every instance
// of a call, in the batch layout.  One copy per GPU, at its own address on each, held until the process ends.  The
// name below is a host-side handle, not a device address: host code reaches the memory only through the pointer
// hipGetSymbolAddress returns for the current device (c2d_static_scratch).
#define MPCX_C2D_SCRATCH_DOUBLES ((size_t)1 << 21) /* 16 MiB */
static __device__ __align__(256) double mpcx_c2d_scratch[MPCX_C2D_SCRATCH_DOUBLES];

namespace mpcx {
bool qp_lane_supports(int n, int m, int r);
namespace {

constexpr int kBlock = 64;
constexpr double kNaN = __builtin_nan("");

// I - Ac dt with the product rounded before the subtraction, as `identity - mat_ac.scale(dt)` (model.rs:56, :139).
__device__ __forceinline__ double one_minus_scaled(double id, double ac, double dt) {
#pragma clang fp contract(off)
    const double scaled = ac * dt;
    return id - scaled;
}

// A = (I - Ac dt)^-1, all in registers; B = A (Bc dt).  Up to 4 states: the closed-form inverse nalgebra's try_inverse
// takes for these sizes (singular when the determinant is exactly zero); above: Gauss-Jordan with partial pivoting.
template <int N, int M>
__device__ __forceinline__ bool c2d_regs(const double (&Ac)[N][N], const double (&Bc)[N][M], double dt, double (&A)[N][N],
                                         double (&B)[N][M]) {
    double a[N][N], inv[N][N];
    if constexpr (N <= 4) {
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) a[i][j] = one_minus_scaled(i == j ? 1.0 : 0.0, Ac[i][j], dt);
        const bool ok = inverse_closed_form<N>(a, inv);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) A[i][j] = ok ? inv[i][j] : kNaN;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int k = 0; k < M; ++k) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) s += A[i][j] * (Bc[j][k] * dt);  // model.rs:57 / :140
                B[i][k] = s;
            }
        return ok;
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            a[i][j] = one_minus_scaled(i == j ? 1.0 : 0.0, Ac[i][j], dt);
            inv[i][j] = (i == j ? 1.0 : 0.0);
        }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double best = fabs(a[k][k]);
        int p = k;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const double v = fabs(a[i][k]);
            if (v > best) { best = v; p = i; }
        }
        if (!(best > 0.0) || !isfinite(best)) ok = false;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const bool sw = (p == i);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double t0 = a[k][j], t1 = a[i][j];
                a[k][j] = sw ? t1 : t0;
                a[i][j] = sw ? t0 : t1;
                const double u0 = inv[k][j], u1 = inv[i][j];
                inv[k][j] = sw ? u1 : u0;
                inv[i][j] = sw ? u0 : u1;
            }
        }
        const double rp = 1.0 / a[k][k];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i == k) continue;
            const double l = a[i][k] * rp;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j > k) a[i][j] -= l * a[k][j];
                inv[i][j] -= l * inv[k][j];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double rd = 1.0 / a[i][i];
#pragma unroll
        for (int j = 0; j < N; ++j) A[i][j] = ok ? inv[i][j] * rd : kNaN;
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < M; ++k) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) s += A[i][j] * (Bc[j][k] * dt);  // model.rs:57 / :140
            B[i][k] = s;
        }
    return ok;
}

#define LANE_PROLOGUE                                                         \
    const long long b = (long long)blockIdx.x * kBlock + threadIdx.x;         \
    if (b >= batch) return;                                                   \
    const long long tile = b / W, lane = b - tile * W;
#define AT(ptr, E, e) ((ptr)[tile * (long long)(E) * W + (long long)(e) * W + lane])

template <int N, int M>
__global__ __launch_bounds__(kBlock) void c2d_lane_kernel(const double *__restrict__ Acp, const double *__restrict__ Bcp, double dt,
                                                         double *__restrict__ Ap, double *__restrict__ Bp, int *info,
                                                         long long batch, long long W) {
    LANE_PROLOGUE
    double Ac[N][N], Bc[N][M], A[N][N], B[N][M];
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) Ac[i][j] = AT(Acp, N * N, j * N + i);
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) Bc[i][j] = AT(Bcp, N * M, j * N + i);
    const bool ok = c2d_regs<N, M>(Ac, Bc, dt, A, B);
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) AT(Ap, N * N, j * N + i) = A[i][j];
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) AT(Bp, N * M, j * N + i) = B[i][j];
    if (info) info[b] = ok ? 0 : 1;
}

// KIND 0: dc motor, parameters (b, j, k, l, r), n = 2 (model.rs:47-52).  KIND 1: two masses, (m1, m2, k1, k2, d1, d2),
// n = 4 (model.rs:128-135).  Parameters use the batch layout with 5 resp. 6 elements per instance.
template <int KIND>
__global__ __launch_bounds__(kBlock) void plants_kernel(const double *__restrict__ prm, double dt, double *__restrict__ Ap,
                                                       double *__restrict__ Bp, double *__restrict__ Cp, int *info,
                                                       long long batch, long long W) {
    constexpr int N = KIND == 0 ? 2 : 4, NP = KIND == 0 ? 5 : 6;
    LANE_PROLOGUE
    double q[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) q[e] = AT(prm, NP, e);
    double Ac[N][N], Bc[N][1], A[N][N], B[N][1];
    if constexpr (KIND == 0) {
        const double pb = q[0], pj = q[1], pk = q[2], pl = q[3], pr = q[4];
        Ac[0][0] = -pb / pj; Ac[0][1] = pk / pj;
        Ac[1][0] = -pk / pl; Ac[1][1] = -pr / pl;
        Bc[0][0] = 0.0; Bc[1][0] = 1.0 / pl;
    } else {
        const double m1 = q[0], m2 = q[1], k1 = q[2], k2 = q[3], d1 = q[4], d2 = q[5];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) Ac[i][j] = 0.0;
        Ac[0][1] = 1.0;
        Ac[1][0] = -(k1 + k2) / m1; Ac[1][1] = -(d1 + d2) / m1; Ac[1][2] = k2 / m1; Ac[1][3] = d2 / m1;
        Ac[2][3] = 1.0;
        Ac[3][0] = k2 / m2; Ac[3][1] = d2 / m2; Ac[3][2] = -k2 / m2; Ac[3][3] = -d2 / m2;
        Bc[0][0] = 0.0; Bc[1][0] = 0.0; Bc[2][0] = 0.0; Bc[3][0] = 1.0 / m2;
    }
    const bool ok = c2d_regs<N, 1>(Ac, Bc, dt, A, B);
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) AT(Ap, N * N, j * N + i) = A[i][j];
#pragma unroll
    for (int i = 0; i < N; ++i) AT(Bp, N, i) = B[i][0];
    if (Cp) {
#pragma unroll
        for (int j = 0; j < N; ++j) AT(Cp, N, j) = (j == 0) ? 1.0 : 0.0;  // model.rs:52 / :135
    }
    if (info) info[b] = ok ? 0 : 1;
}

// general n: Mt = I - Ac dt and A = I (the right-hand side the solver then overwrites in place)
__global__ __launch_bounds__(256) void c2d_m_kernel(const double *Acp, double dt, double *Mt, double *Ap, long long batch, long long W,
                                                    int n) {
    const long long E = (long long)n * n, idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= batch * E) return;
    const long long b = idx / E, e = idx - b * E;
    const int j = (int)(e / n), i = (int)(e - (long long)j * n);
    const long long at = elem_index(b, e, E, W);
    const double id = (i == j) ? 1.0 : 0.0;
    Mt[at] = one_minus_scaled(id, Acp[at], dt);  // product rounded before the subtraction: the closed forms test det == 0.0
    Ap[at] = id;
}

// B = A (Bc dt)
__global__ __launch_bounds__(256) void c2d_b_kernel(const double *Ap, const double *Bcp, double dt, double *Bp, long long batch,
                                                    long long W, int n, int m) {
    const long long E = (long long)n * m, idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= batch * E) return;
    const long long b = idx / E, e = idx - b * E;
    const int k = (int)(e / n), i = (int)(e - (long long)k * n);
    double s = 0.0;
    for (int j = 0; j < n; ++j)
        s += Ap[elem_index(b, (long long)j * n + i, (long long)n * n, W)] * (Bcp[elem_index(b, (long long)k * n + j, E, W)] * dt);
    Bp[elem_index(b, e, E, W)] = s;
}

template <int N, int M>
int launch_c2d_one(const double *Ac, const double *Bc, double dt, double *A, double *B, int *info, long long batch, long long W,
                   hipStream_t s) {
    const long long blocks = (batch + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((c2d_lane_kernel<N, M>), dim3((unsigned)blocks), dim3(kBlock), 0, s, Ac, Bc, dt, A, B, info, batch, W);
    return (int)hipGetLastError();
}

}  // namespace

bool c2d_lane_supports(int n, int m) { return (m == 1 && n >= 1 && n <= 6) || (m == 2 && (n == 2 || n == 3 || n == 4 || n == 6)); }

int launch_c2d_lane(const double *Ac, const double *Bc, double dt, double *A, double *B, int *info, long long batch, long long W,
                    int n, int m, hipStream_t s) {
#define CASE(N_, M_) \
    if (n == N_ && m == M_) return launch_c2d_one<N_, M_>(Ac, Bc, dt, A, B, info, batch, W, s);
    CASE(1, 1) CASE(2, 1) CASE(3, 1) CASE(4, 1) CASE(5, 1) CASE(6, 1)
    CASE(2, 2) CASE(3, 2) CASE(4, 2) CASE(6, 2)  // keep c2d_lane_supports in step with this list
#undef CASE
    return MPCX_ERR_UNSUPPORTED;
}

// tmp: n x n doubles per instance in the batch layout (for M = I - Ac dt).
int launch_c2d_general(const double *Ac, const double *Bc, double dt, double *A, double *B, int *info, long long batch,
                       long long W, int n, int m, double *tmp, double *ws, size_t ws_doubles, hipStream_t s) {
    const long long ea = batch * (long long)n * n, eb = batch * (long long)n * m;
    if ((ea + 255) / 256 > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(c2d_m_kernel, dim3((unsigned)((ea + 255) / 256)), dim3(256), 0, s, Ac, dt, tmp, A, batch, W, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int rc = launch_solve_inplace(tmp, A, info, batch, W, n, n, ws, ws_doubles, s);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(c2d_b_kernel, dim3((unsigned)((eb + 255) / 256)), dim3(256), 0, s, A, Bc, dt, B, batch, W, n, m);
    return (int)hipGetLastError();
}

int launch_plants(int kind, const double *params, double dt, double *A, double *B, double *C, int *info, long long batch,
                  long long W, hipStream_t s) {
    const long long blocks = (batch + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    if (kind == 0)
        hipLaunchKernelGGL(plants_kernel<0>, dim3((unsigned)blocks), dim3(kBlock), 0, s, params, dt, A, B, C, info, batch, W);
    else if (kind == 1)
        hipLaunchKernelGGL(plants_kernel<1>, dim3((unsigned)blocks), dim3(kBlock), 0, s, params, dt, A, B, C, info, batch, W);
    else
        return MPCX_ERR_UNSUPPORTED;
    return (int)hipGetLastError();
}

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
double *g_scratch_ptr[kMaxDevices];  // device address of mpcx_c2d_scratch on each GPU, resolved once per device
}  // namespace

size_t c2d_static_scratch_doubles() { return MPCX_C2D_SCRATCH_DOUBLES; }

// Device address of this GPU's copy of the static scratch.  `device` must be the current device.  The first call for a
// device resolves the symbol (which loads the code object there if needed); later calls are a table look-up.
int c2d_static_scratch(int device, double **ptr) {
    *ptr = nullptr;
    if (device < 0 || device >= kMaxDevices) return MPCX_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(mpcx_c2d_scratch));
        if (e != hipSuccess) return (int)e;
        if (!p) return (int)hipErrorInvalidSymbol;
        g_scratch_ptr[device] = (double *)p;
    }
    *ptr = g_scratch_ptr[device];
    return 0;
}

}  // namespace mpcx
