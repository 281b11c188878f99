// Internal declarations shared by the kernel files and the C ABI of libmpcx (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpcx.h"

namespace mpcx {

// HIP device index a context is bound to.
int ctx_device(const mpcx_ctx *ctx);

// Arguments of one batched QP build; every pointer is a device pointer, W is the resolved interleave width
// (1 <= W <= batch).
struct QpArgs {
    const double *A, *B, *C, *W3, *W4, *x0, *ref;
    double *Phi, *Gamma, *H, *fvec;
    long long batch, W;
    int n, m, r, f, v;
    int weights_mode;
    // two ints of device scratch in which the workgroup kernels measure the band structure of W3, or null
    void *w3win;
    size_t w3win_bytes;
    int cu_count;  // compute units of the device (filled in by the workgroup kernels' launcher; 0: unknown)
};

struct RolloutArgs {
    const double *A, *B, *C, *x0, *U;
    double *X, *Y;
    long long batch, W, T;
    int n, m, r;
    int u_shared;
};

// Receding-horizon closed loop: Phi (f r x n) and gain (v m x f r) per instance, ref r x (T + f) per instance or shared.
struct LoopArgs {
    const double *A, *B, *C, *x0, *Phi, *gain, *ref;
    double *X, *U, *Y;
    long long batch, W, T;
    int n, m, r, f, v;
    int ref_shared;
};

// Each launcher returns hipSuccess / a hipError_t, or MPCX_ERR_UNSUPPORTED when the dimensions are outside
// what that kernel family holds on chip (the caller then tries the next family).
int launch_qp_lane(const QpArgs &a, hipStream_t s);
int launch_qp_wg(const QpArgs &a, int hessian_algo, hipStream_t s);
// The same build without an on-chip size limit (block weights, instance-major): Markov vectors in a device workspace of
// qp_stream_ws_doubles(a) doubles per instance (fewer instances' worth means more launches).
bool qp_wg_takes(const QpArgs &a);
bool qp_stream_takes(const QpArgs &a);
size_t qp_stream_ws_doubles(const QpArgs &a);
int launch_qp_stream(const QpArgs &a, double *ws, size_t ws_doubles, hipStream_t s);
int launch_hessian_mfma(const QpArgs &a, hipStream_t s);
int launch_qp_dense(const QpArgs &a, hipStream_t s);
// ws / ws_doubles: device workspace for systems beyond 64 unknowns (solve_large_ws_doubles per instance; fewer
// instances' worth means more launches); ignored by the on-chip kernels.
int launch_qp_solve(const double *H, const double *fvec, double *u_seq, int *info, long long batch, long long W, int vm,
                    double *ws, size_t ws_doubles, hipStream_t s);
int launch_qp_gain(const double *H, const double *Gamma, const double *Q, double *gain, int *info, long long batch,
                   long long W, int m, int r, int f, int v, double *ws, size_t ws_doubles, hipStream_t s);
// transposed = 1 solves with H^T instead (H is read transposed)
int launch_solve_inplace(const double *H, double *X, int *info, long long batch, long long W, int nr, int nrhs, double *ws,
                         size_t ws_doubles, hipStream_t s, int transposed = 0);
// First m rows of the gain H^-1 Gamma^T W4 (m x f r per instance, column-major): m solves with H^T in `y` (v m x m per
// instance in the batch layout, scratch), then rows = (Gamma y)^T W4.  W4 is the r x r block or, dense != 0, the
// (f r) x (f r) matrix.
int launch_qp_gain_rows(const double *H, const double *Gamma, const double *W4, int dense, double *rows, int *info,
                        long long batch, long long W, int m, int r, int f, int v, double *y, double *ws, size_t ws_doubles,
                        hipStream_t s);
int launch_c2d_lane(const double *Ac, const double *Bc, double dt, double *A, double *B, int *info, long long batch, long long W,
                    int n, int m, hipStream_t s);
bool c2d_lane_supports(int n, int m);  // launch_c2d_lane takes the shape (no scratch needed)
int launch_c2d_general(const double *Ac, const double *Bc, double dt, double *A, double *B, int *info, long long batch,
                       long long W, int n, int m, double *tmp, double *ws, size_t ws_doubles, hipStream_t s);
// Static device scratch of the allocation-free discretisation: capacity in doubles, and its address on the current
// device (`device` must be current; returns 0, a hipError_t, or MPCX_ERR_UNSUPPORTED for a device index beyond the table).
size_t c2d_static_scratch_doubles();
int c2d_static_scratch(int device, double **ptr);
int launch_plants(int kind, const double *params, double dt, double *A, double *B, double *C, int *info, long long batch,
                  long long W, hipStream_t s);
bool solve_fits_on_chip(int nr, int nrhs);            // a one-wavefront register/LDS kernel takes it (up to 64 unknowns)
int launch_qp_gain_dense(const double *H, const double *Gamma, const double *W4, double *gain, int *info, long long batch,
                         long long W, int m, int r, int f, int v, double *ws, size_t ws_doubles, hipStream_t s);
size_t solve_large_ws_doubles(int nr, int nrhs);
bool solve_large_supports(int nr);
int launch_solve_large(const double *H, const double *R, const double *Q, int r, double *X, int *info, long long batch,
                       long long W, int nr, int nrhs, int negate, double *ws, size_t ws_doubles, hipStream_t s, int transposed = 0);
int launch_closed_loop_lane(const LoopArgs &a, hipStream_t s);
int launch_closed_loop_group(const LoopArgs &a, hipStream_t s);
int launch_rollout_lane(const RolloutArgs &a, hipStream_t s);
int launch_rollout_group(const RolloutArgs &a, hipStream_t s);
int launch_stats(const double *buf, long long count, mpcx_stats *out_dev, void *scratch, size_t scratch_bytes,
                 hipStream_t s);
size_t stats_scratch_bytes();
// `batch` instances, starting at instance src_first of the source array and written from instance dst_first of the
// destination array on (each array in its own interleave width)
int launch_relayout(long long batch, long long E, long long Wsrc, const double *src, long long Wdst, double *dst,
                    hipStream_t s, long long src_first = 0, long long dst_first = 0);
// True when the one-lane-per-instance build kernels take this problem (any batch layout); everything else runs on the
// one-workgroup-per-instance kernels, which take instance-major data.
bool qp_lane_takes(const QpArgs &a);

// Workgroup barrier that orders LDS traffic only.  `__syncthreads()` also waits for every outstanding global store
// of the wave (s_waitcnt vmcnt(0)), which in a loop that stores to HBM each step serialises the loop on the store
// latency; the kernels here never pass data between waves through global memory, so LDS ordering is all they need.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// A double moved across lanes by a data-parallel-primitive pattern (an ALU-speed exchange inside a row of 16 lanes;
// CTRL: quad permutation 0x00..0xFF, row rotate right by k = 0x120 + k).  Every lane has a valid source lane.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    // (mov_dpp: no "old" operand to initialise, one instruction per half)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Index of element e of instance b in an array with E elements per instance and interleave width W.
__host__ __device__ inline long long elem_index(long long b, long long e, long long E, long long W) {
    const long long tile = b / W;
    return tile * E * W + e * W + (b - tile * W);
}

}  // namespace mpcx
