// First-move rows of the gain: the first m rows of H^-1 Gamma^T W4.
//
// Reference: src/control/controller.rs:93-96 builds the whole gain_matrix ((v m) x (f r)), but compute_control_input
// (:130-132) applies only its first row (first m rows for m inputs): u = gain[0:m, :] s.  Row i of H^-1 R is
// (H^-T e_i)^T R, so those rows cost m solves with H^T instead of f r solves with H:
//     H^T y_i = e_i  (i < m);      rows[i, :] = y_i^T Gamma^T W4 = ((Gamma y_i)^T) W4.
// The solves run on the batched solver (same pivoting rule and the same per-instance failure flag as the full gain;
// systems of 1..4 unknowns invert H itself in closed form, as try_inverse does, and use the inverse transposed); the
// product reads Gamma once and is bound by that read.
#include "mpcx_internal.h"

namespace mpcx {
namespace {

// y <- the first m columns of the identity (v m x m per instance, batch layout)
__global__ __launch_bounds__(256) void unit_columns_kernel(double *y, long long batch, long long W, int nr, int m) {
    const long long E = (long long)nr * m, idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= batch * E) return;
    const long long b = idx % batch, e = idx / batch;  // instance fastest: contiguous in the interleaved layouts
    const int c = (int)(e / nr), k = (int)(e - (long long)c * nr);
    y[elem_index(b, e, E, W)] = (k == c) ? 1.0 : 0.0;
}

// One lane per instance (interleaved layouts, small systems): y in the lane's LDS column,
//     rows[i, (blk, a2)] = sum_k y[k, i] sum_a Gamma[blk r + a, k] Q[a][a2].
// Every global access of a wavefront is the same element of 64 consecutive instances.
__global__ __launch_bounds__(64) void rows_lane_kernel(const double *__restrict__ Gamma, const double *__restrict__ Q,
                                                      const double *__restrict__ y, double *__restrict__ rows, long long batch,
                                                      long long W, int m, int r, int f, int v) {
    extern __shared__ double ylds[];  // [i * VM + k][64]: y (v m x m, column-major), one column of slots per lane
    const int tid = threadIdx.x;
    const long long b = (long long)blockIdx.x * 64 + tid;
    if (b >= batch) return;
    const int VM = v * m, FR = f * r;
    const long long EY = (long long)VM * m, EG = (long long)FR * VM, ER = (long long)m * FR;
    for (int e = 0; e < VM * m; ++e) ylds[e * 64 + tid] = y[elem_index(b, e, EY, W)];
    const double *g = Gamma + elem_index(b, 0, EG, W);
    double *out = rows + elem_index(b, 0, ER, W);
    for (int blk = 0; blk < f; ++blk)
        for (int a2 = 0; a2 < r; ++a2)
            for (int i = 0; i < m; ++i) {
                double acc = 0.0;
                if (r == 1) {  // one load per term: keep several in flight
                    const double q = Q[0];
                    const double *gp = g + (long long)blk * W;
#pragma unroll 8
                    for (int k = 0; k < VM; ++k) acc += ylds[(i * VM + k) * 64 + tid] * (gp[(long long)k * FR * W] * q);
                } else {
#pragma unroll 2
                    for (int k = 0; k < VM; ++k) {
                        double t = 0.0;
                        for (int a = 0; a < r; ++a) t += g[((long long)k * FR + blk * r + a) * W] * Q[a2 * r + a];
                        acc += ylds[(i * VM + k) * 64 + tid] * t;
                    }
                }
                out[((long long)(blk * r + a2) * m + i) * W] = acc;
            }
}

// One workgroup per instance and chunk of CR rows of Gamma (CR a multiple of r): thread `row` forms T[row, :] = Gamma[row, :] y
// with lanes along the rows (for instance-major data every read of Gamma is a contiguous run of a column), the chunk of T
// goes to LDS and the weight block is applied from there.  dense: one chunk holds all f r rows and the weight is the whole
// (f r) x (f r) matrix.
__global__ __launch_bounds__(256) void rows_wg_kernel(const double *__restrict__ Gamma, const double *__restrict__ W4,
                                                     const double *__restrict__ y, double *__restrict__ rows, long long W, int m,
                                                     int r, int f, int v, int CR, int chunks, int dense) {
    extern __shared__ double sm[];
    const int VM = v * m, FR = f * r, tid = threadIdx.x;
    const long long b = blockIdx.x / chunks;
    const int row0 = (int)(blockIdx.x - b * chunks) * CR, nrows = (FR - row0 < CR) ? FR - row0 : CR;
    double *ys = sm, *ts = sm + (size_t)VM * m;  // ys[i * VM + k], ts[i * CR + row]
    const long long EY = (long long)VM * m, EG = (long long)FR * VM, ER = (long long)m * FR;
    for (int e = tid; e < VM * m; e += 256) ys[e] = y[elem_index(b, e, EY, W)];
    __syncthreads();
    for (int row = tid; row < nrows; row += 256) {
        for (int i0 = 0; i0 < m; i0 += 4) {  // four columns of y at a time: Gamma is re-read (from cache) once per four inputs
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            const double *g = Gamma + elem_index(b, row0 + row, EG, W);
            for (int k = 0; k < VM; ++k) {
                const double gv = g[(long long)k * FR * W];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u < m) acc[u] += gv * ys[(i0 + u) * VM + k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u < m) ts[(i0 + u) * CR + row] = acc[u];
        }
    }
    __syncthreads();
    if (dense) {  // rows[i, c] = sum_row T[row, i] W4[row, c]
        for (int e = tid; e < m * FR; e += 256) {
            const int c = e / m, i = e - c * m;
            double acc = 0.0;
            for (int row = 0; row < FR; ++row) acc += ts[i * CR + row] * W4[(long long)c * FR + row];
            rows[elem_index(b, e, ER, W)] = acc;
        }
    } else {  // rows[i, (blk, a2)] = sum_a T[blk r + a, i] Q[a][a2]
        for (int e = tid; e < m * nrows; e += 256) {
            const int c = e / m, i = e - c * m;  // c: row of Gamma within the chunk = column of the output
            const int blk = c / r, a2 = c - blk * r;
            double acc = 0.0;
            for (int a = 0; a < r; ++a) acc += ts[i * CR + blk * r + a] * W4[a2 * r + a];
            rows[elem_index(b, (long long)(row0 + c) * m + i, ER, W)] = acc;
        }
    }
}

}  // namespace

int launch_qp_gain_rows(const double *H, const double *Gamma, const double *W4, int dense, double *rows, int *info,
                        long long batch, long long W, int m, int r, int f, int v, double *y, double *ws, size_t ws_doubles,
                        hipStream_t s) {
    const int VM = v * m, FR = f * r;
    const long long ey = batch * (long long)VM * m;
    if ((ey + 255) / 256 > 0x7fffffffLL || batch > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(unit_columns_kernel, dim3((unsigned)((ey + 255) / 256)), dim3(256), 0, s, y, batch, W, VM, m);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int rc = launch_solve_inplace(H, y, info, batch, W, VM, m, ws, ws_doubles, s, /*transposed=*/1);
    if (rc != 0) return rc;
    // the product
    const size_t lane_lds = (size_t)VM * m * 64 * sizeof(double);
    if (!dense && W != 1 && lane_lds <= 64 * 1024 && (batch + 63) / 64 <= 0x7fffffffLL) {
        hipLaunchKernelGGL(rows_lane_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), lane_lds, s, Gamma, W4, y, rows, batch, W,
                           m, r, f, v);
        return (int)hipGetLastError();
    }
    // chunk of rows per workgroup: a multiple of r near 256 (all rows with a dense weight)
    int CR = dense ? FR : r * (256 / r > 0 ? 256 / r : 1);
    if (CR > FR) CR = FR;
    const size_t lds = ((size_t)VM * m + (size_t)CR * m) * sizeof(double);
    if (lds > 160 * 1024) return MPCX_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        e = hipFuncSetAttribute((const void *)rows_wg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const int chunks = (FR + CR - 1) / CR;
    if (batch * chunks > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(rows_wg_kernel, dim3((unsigned)(batch * chunks)), dim3(256), lds, s, Gamma, W4, y, rows, W, m, r, f, v, CR,
                       chunks, dense);
    return (int)hipGetLastError();
}

}  // namespace mpcx
