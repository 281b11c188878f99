// Solve with the condensed Hessian for systems too large for LDS (v m from 65 up to about 1200 unknowns): blocked
// elimination with partial pivoting on the augmented matrix [H | R] held in a device workspace, one workgroup per
// instance, trailing updates on the FP64 matrix cores.
//
// Reference: src/control/controller.rs:93-96 (try_inverse of the Hessian, then H^-1 M^T W4).  Pivot rule and failure
// test are those of an LU with partial pivoting, as nalgebra's: at column k the largest |a[i][k]| among the rows not
// yet used as a pivot; a zero or non-finite pivot marks the instance failed (the reference's "Matrix inversion failed").
//
// Algorithm, per panel of kNB = 16 columns starting at k0:
//   1. the whole panel (all rows, 16 columns) is loaded into LDS and factored there: for column j the pivot is searched
//      in rows >= k0 + j, exchanged into row k0 + j, and every other row outside the panel's pivot block, the rows
//      above k0 included, gets its multiplier l = a / pivot and is reduced in the remaining panel columns.
//      Eliminating above as well as below (block Gauss-Jordan) means there is no backward sweep over the matrix.
//   2. each wavefront takes blocks of 16 columns to the right (the rest of H and the right-hand sides): it applies the
//      panel's row exchanges, forms U12 = L11^-1 T for the 16 pivot rows (one 16x16x16 product with the inverse of the
//      panel's unit lower triangle) and subtracts (multipliers) x U12 from every other 16-row tile.  Products are
//      computed transposed, so that a lane's four accumulators are four columns of one row and a tile is read and
//      written as 128-byte runs of the column-major workspace.
// What remains is block diagonal with the 16x16 upper triangles U11 of the panels: lu_unpack solves those per
// right-hand side and scatters the result into the caller's layout.
#include "mpcx_internal.h"

namespace mpcx {
namespace {

using d4 = __attribute__((ext_vector_type(4))) double;
constexpr int kNB = 16;
constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxTouched = 2 * kNB;

struct LuArgs {
    double *ws;      // per instance: column-major nrp x ncp, leading dimension nrp
    int *failed;     // per instance
    long long count; // instances in this launch
    int nrp, ncp;
};

// ---- pack: workspace <- [H padded with an identity block | right-hand side padded with zeros] ----
struct PackArgs {
    const double *H, *R, *Q;  // Q != nullptr: R is Gamma (fr x nr per instance) and the right-hand side is Gamma^T blockdiag(Q)
    double *ws;
    long long first, count, W;  // instances [first, first + count) of the batch
    int nr, nrhs, r, nrp, ncp;
};

// Elementwise part: the H block (with its identity padding) and, when the right-hand side is given as such (Q == nullptr),
// the right-hand side columns.  `ncols` = nrp or ncp accordingly.  Lanes run along the rows of a column on both sides.
__global__ __launch_bounds__(256) void lu_pack_kernel(PackArgs p, int ncols) {
    const long long per = (long long)p.nrp * ncols;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.count * per) return;
    const long long li = idx / per, e = idx - li * per;
    const int col = (int)(e / p.nrp), row = (int)(e - (long long)col * p.nrp);
    const long long b = p.first + li;
    double v;
    if (col < p.nrp) {
        v = (row < p.nr && col < p.nr) ? p.H[elem_index(b, (long long)col * p.nr + row, (long long)p.nr * p.nr, p.W)]
                                       : (row == col ? 1.0 : 0.0);
    } else {
        const int c = col - p.nrp;
        v = (row < p.nr && c < p.nrhs) ? p.R[elem_index(b, (long long)c * p.nr + row, (long long)p.nr * p.nrhs, p.W)] : 0.0;
    }
    p.ws[li * (long long)p.nrp * p.ncp + e] = v;
}

// Right-hand side Gamma^T blockdiag(Q) (controller.rs:96, M^T W4): column c = blk r + a2, row k is
// sum_a Gamma[blk r + a][k] Q[a][a2].  Gamma is (f r) x (v m) column-major, so its rows are the contiguous direction
// while the workspace wants k contiguous: a workgroup moves a tile of 16 unknowns x TC rows of Gamma (TC a multiple of r)
// through LDS, reading along Gamma's rows and writing along the workspace's.
constexpr int kTileK = 16, kTileCMax = 64;
__global__ __launch_bounds__(256) void lu_pack_gamma_kernel(PackArgs p, int TC, int CT, int KT) {
    __shared__ double tile[kTileK][kTileCMax + 1];
    long long t = blockIdx.x;
    const int kt = (int)(t % KT);
    t /= KT;
    const int ct = (int)(t % CT);
    const long long li = t / CT, b = p.first + li;
    const int k0 = kt * kTileK, c0 = ct * TC, r = p.r;
    const long long ER = (long long)p.nr * p.nrhs;
    for (int e = threadIdx.x; e < kTileK * TC; e += 256) {
        const int kk = e / TC, cc = e - kk * TC, k = k0 + kk, cp = c0 + cc;
        tile[kk][cc] = (k < p.nr && cp < p.nrhs) ? p.R[elem_index(b, (long long)k * p.nrhs + cp, ER, p.W)] : 0.0;
    }
    __syncthreads();
    double *out = p.ws + li * (long long)p.nrp * p.ncp + (long long)p.nrp * p.nrp;
    const int ncr = p.ncp - p.nrp;  // padded right-hand side columns
    for (int e = threadIdx.x; e < kTileK * TC; e += 256) {
        const int cc = e / kTileK, kk = e - cc * kTileK, c = c0 + cc, k = k0 + kk;
        if (c >= ncr) continue;
        double v = 0.0;
        if (k < p.nr && c < p.nrhs) {
            const int blk = cc / r, a2 = cc - blk * r;
            for (int a = 0; a < r; ++a) v += tile[kk][blk * r + a] * p.Q[a2 * r + a];
        }
        out[(long long)c * p.nrp + k] = v;
    }
}

// ---- factor / eliminate: one workgroup per instance ----
__global__ __launch_bounds__(kThreads) void lu_block_kernel(LuArgs p) {
    extern __shared__ __align__(16) double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nrp = p.nrp, ncp = p.ncp;
    double *panel = sm;                                  // [16][nrp]: panel[j * nrp + row]
    double *linv = panel + (size_t)kNB * nrp;            // [16][16]: linv[i * 16 + k] = (L11^-1)[i][k]
    double *tiles = linv + kNB * kNB;                    // per wavefront [16][16]: T[row * 16 + col]
    double *red_val = tiles + kWaves * kNB * kNB;        // [kWaves]
    int *red_idx = (int *)(red_val + kWaves);            // [kWaves]
    int *ipiv = red_idx + kWaves;                        // [16]
    int *trow = ipiv + kNB;                              // [32] rows whose content changes with the panel's exchanges
    int *tsrc = trow + kMaxTouched;                      // [32] the row whose old content each of them receives
    int *tcount = tsrc + kMaxTouched;                    // [1]
    double *A = p.ws + (size_t)blockIdx.x * nrp * ncp;
    double *mytile = tiles + wv * kNB * kNB;

    for (int k0 = 0; k0 < nrp; k0 += kNB) {
        // 1. panel -> LDS (16 adjacent columns are one contiguous run of the workspace)
        {
            const double *src = A + (size_t)k0 * nrp;
            for (int e = tid; e < kNB * nrp; e += kThreads) panel[e] = src[e];
        }
        __syncthreads();
        for (int j = 0; j < kNB; ++j) {
            double *col = panel + (size_t)j * nrp;
            const int kk = k0 + j;
            double best = -1.0;
            int bi = kk;
            for (int r = kk + tid; r < nrp; r += kThreads) {
                double a = fabs(col[r]);
                if (!(a == a)) a = INFINITY;  // a NaN must be chosen (and fail), not skipped
                if (a > best) { best = a; bi = r; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off);
                const int oi = __shfl_xor(bi, off);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (lane == 0) { red_val[wv] = best; red_idx[wv] = bi; }
            __syncthreads();
            best = red_val[0];
            bi = red_idx[0];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) {
                const double ob = red_val[w];
                const int oi = red_idx[w];
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (!(best > 0.0) || !isfinite(best)) {  // uniform over the workgroup
                if (tid == 0) p.failed[blockIdx.x] = 1;
                return;
            }
            if (tid < kNB) {
                const double t = panel[(size_t)tid * nrp + kk];
                panel[(size_t)tid * nrp + kk] = panel[(size_t)tid * nrp + bi];
                panel[(size_t)tid * nrp + bi] = t;
                if (tid == 0) ipiv[j] = bi;
            }
            __syncthreads();
            const double inv = 1.0 / col[kk];
            for (int r = tid; r < nrp; r += kThreads) {
                if (r >= k0 && r <= kk) continue;  // the panel's earlier pivot rows keep their entry: it belongs to U11
                const double l = col[r] * inv;
                col[r] = l;
#pragma unroll
                for (int jj = 1; jj < kNB; ++jj)
                    if (j + jj < kNB) panel[(size_t)(j + jj) * nrp + r] -= l * panel[(size_t)(j + jj) * nrp + kk];
            }
            __syncthreads();
        }
        // 2. panel back to the workspace (U11 is needed by the final solve), inverse of the unit lower triangle, and the
        //    net effect of the 16 exchanges: which rows end up with whose old content
        {
            double *dst = A + (size_t)k0 * nrp;
            for (int e = tid; e < kNB * nrp; e += kThreads) dst[e] = panel[e];
        }
        if (tid < kNB) {  // column tid of L11^-1 by forward substitution on a unit vector
            double x[kNB];
#pragma unroll
            for (int i = 0; i < kNB; ++i) {
                double s = (i == tid) ? 1.0 : 0.0;
#pragma unroll
                for (int q = 0; q < kNB; ++q)
                    if (q < i) s -= panel[(size_t)q * nrp + k0 + i] * x[q];
                x[i] = s;
                linv[i * kNB + tid] = s;
            }
        }
        if (tid == 64) {  // another wavefront than the one inverting L11
            int n = kNB;
            for (int q = 0; q < kNB; ++q) { trow[q] = k0 + q; tsrc[q] = k0 + q; }
            for (int j = 0; j < kNB; ++j) {
                const int b = ipiv[j];
                if (b == k0 + j) continue;
                int ib = -1;
                for (int q = 0; q < n; ++q)
                    if (trow[q] == b) ib = q;
                if (ib < 0) { ib = n; trow[n] = b; tsrc[n] = b; ++n; }
                const int t = tsrc[j];
                tsrc[j] = tsrc[ib];
                tsrc[ib] = t;
            }
            *tcount = n;
        }
        __syncthreads();
        // 3. the columns to the right, one block of 16 per wavefront at a time
        const int ntouched = *tcount;
        d4 la;  // A operand of the triangular solve: (L11^-1)[lane & 15][(lane >> 4) + 4 kk]
#pragma unroll
        for (int q = 0; q < 4; ++q) la[q] = linv[(lane & 15) * kNB + (lane >> 4) + 4 * q];
        for (int c0 = k0 + kNB + wv * kNB; c0 < ncp; c0 += kWaves * kNB) {
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
            d4 u12 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double tb = mytile[((lane >> 4) + 4 * q) * kNB + (lane & 15)];
                u12 = __builtin_amdgcn_mfma_f64_16x16x4f64(la[q], tb, u12, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) A[(size_t)(c0 + (lane & 15)) * nrp + k0 + (lane >> 4) + 4 * q] = u12[q];
            // every other 16-row tile: C^T -= U12^T L^T.  A operand [i' = column][k] = U12[k][column] is u12 as it stands.
            const d4 nu = -u12;
            // Tiles are taken four at a time; the next four are fetched before the current four are stored, so the
            // loads of a group are in flight while the previous group is computed and written back.
            const int pidx = k0 / kNB, nt = nrp / kNB - 1, nt4 = nt & ~3;
            double *cbase = A + (size_t)(c0 + (lane >> 4)) * nrp + (lane & 15);  // + r0, + 4 q columns
            auto row0 = [&](int t) { return (t < pidx ? t : t + 1) * kNB; };
            d4 buf[4], nbuf[4];
            if (nt4 > 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) buf[u][q] = cbase[(size_t)(4 * q) * nrp + row0(u)];
            }
            for (int t = 0; t < nt4; t += 4) {
                const int tn = (t + 4 < nt4) ? t + 4 : t;  // the last round re-reads its own tiles; the values are not used
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) nbuf[u][q] = cbase[(size_t)(4 * q) * nrp + row0(tn + u)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r0 = row0(t + u);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double lb = panel[(size_t)((lane >> 4) + 4 * q) * nrp + r0 + (lane & 15)];
                        buf[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(nu[q], lb, buf[u], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) cbase[(size_t)(4 * q) * nrp + row0(t + u)] = buf[u][q];
#pragma unroll
                for (int u = 0; u < 4; ++u) buf[u] = nbuf[u];
            }
            for (int t = nt4; t < nt; ++t) {
                const int r0 = row0(t);
                d4 acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = cbase[(size_t)(4 * q) * nrp + r0];
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
    }
}

// ---- unpack: x_p = U11_p^-1 y_p per panel and right-hand side, scattered into the caller's layout ----
struct UnpackArgs {
    const double *ws;
    const int *failed;
    double *X;
    int *info;
    long long first, count, W;
    int nr, nrhs, nrp, ncp, negate;
};

__global__ __launch_bounds__(256) void lu_unpack_kernel(UnpackArgs p) {
    const int panels = p.nrp / kNB;
    const long long per = (long long)panels * p.nrhs;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.count * per) return;
    const long long li = idx / per, rem = idx - li * per;
    const int c = (int)(rem / panels), pn = (int)(rem - (long long)c * panels), k0 = pn * kNB;  // panels fastest: rows contiguous
    const long long b = p.first + li;
    const double *A = p.ws + (size_t)li * p.nrp * p.ncp;
    const bool bad = p.failed[li] != 0;
    if (p.info && rem == 0) p.info[b] = bad ? 1 : 0;
    double x[kNB];
#pragma unroll
    for (int i = 0; i < kNB; ++i) x[i] = A[(size_t)(p.nrp + c) * p.nrp + k0 + i];
#pragma unroll
    for (int i = kNB - 1; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int q = kNB - 1; q >= 0; --q)
            if (q > i) s -= A[(size_t)(k0 + q) * p.nrp + k0 + i] * x[q];
        x[i] = s / A[(size_t)(k0 + i) * p.nrp + k0 + i];
    }
    const double sign = p.negate ? -1.0 : 1.0;
    const long long ER = (long long)p.nr * p.nrhs;
#pragma unroll
    for (int i = 0; i < kNB; ++i)
        if (k0 + i < p.nr)
            p.X[elem_index(b, (long long)c * p.nr + k0 + i, ER, p.W)] = bad ? __longlong_as_double(0x7ff8000000000000LL) : sign * x[i];
}

size_t lu_lds_bytes(int nrp) {
    size_t d = (size_t)kNB * nrp + kNB * kNB + kWaves * kNB * kNB + kWaves;
    return d * sizeof(double) + sizeof(int) * (kWaves + kNB + 2 * kMaxTouched + 4);
}

}  // namespace

// Doubles of workspace one instance needs, and the ints for the failure flags ride behind them.
size_t solve_large_ws_doubles(int nr, int nrhs) {
    const size_t nrp = (size_t)((nr + 15) & ~15), ncp = nrp + (size_t)((nrhs + 15) & ~15);
    return nrp * ncp + 1;  // + one slot for the instance's failure flag
}

bool solve_large_supports(int nr) { return lu_lds_bytes((nr + 15) & ~15) <= 160 * 1024; }

// R / Q as in the small kernels (Q != nullptr: R is Gamma and the right-hand side Gamma^T blockdiag(Q)).  `ws` holds
// `ws_doubles` doubles; the batch is processed in as many launches as that takes.
int launch_solve_large(const double *H, const double *R, const double *Q, int r, double *X, int *info, long long batch,
                       long long W, int nr, int nrhs, int negate, double *ws, size_t ws_doubles, hipStream_t s) {
    const int nrp = (nr + 15) & ~15, ncp = nrp + ((nrhs + 15) & ~15);
    const size_t lds_bytes = lu_lds_bytes(nrp);
    if (lds_bytes > 160 * 1024) return MPCX_ERR_UNSUPPORTED;
    const size_t per = solve_large_ws_doubles(nr, nrhs);
    const long long chunk = (long long)(ws_doubles / per);
    if (chunk < 1) return MPCX_ERR_UNSUPPORTED;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)lu_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    for (long long first = 0; first < batch; first += chunk) {
        const long long count = (batch - first < chunk) ? batch - first : chunk;
        int *failed = (int *)(ws + (size_t)count * nrp * ncp);  // behind the matrices of this launch
        hipError_t e = hipMemsetAsync(failed, 0, (size_t)count * sizeof(int), s);
        if (e != hipSuccess) return (int)e;
        const long long elems = count * (long long)nrp * ncp;
        const long long pblocks = (elems + 255) / 256;
        if (pblocks > 0x7fffffffLL || count > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
        PackArgs pa{H, R, Q, ws, first, count, W, nr, nrhs, r, nrp, ncp};
        if (Q != nullptr && r <= kTileCMax) {
            const long long hblocks = (count * (long long)nrp * nrp + 255) / 256;
            hipLaunchKernelGGL(lu_pack_kernel, dim3((unsigned)hblocks), dim3(256), 0, s, pa, nrp);
            const int TC = r * (kTileCMax / r), CT = (ncp - nrp + TC - 1) / TC, KT = nrp / kTileK;
            const long long tiles = count * (long long)CT * KT;
            if (tiles > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
            hipLaunchKernelGGL(lu_pack_gamma_kernel, dim3((unsigned)tiles), dim3(256), 0, s, pa, TC, CT, KT);
        } else if (Q != nullptr) {
            return MPCX_ERR_UNSUPPORTED;  // weight blocks wider than 64 outputs are not served by the blocked solver
        } else {
            hipLaunchKernelGGL(lu_pack_kernel, dim3((unsigned)pblocks), dim3(256), 0, s, pa, ncp);
        }
        LuArgs la{ws, failed, count, nrp, ncp};
        hipLaunchKernelGGL(lu_block_kernel, dim3((unsigned)count), dim3(kThreads), lds_bytes, s, la);
        const long long uthreads = count * (long long)(nrp / kNB) * nrhs;
        UnpackArgs ua{ws, failed, X, info, first, count, W, nr, nrhs, nrp, ncp, negate};
        hipLaunchKernelGGL(lu_unpack_kernel, dim3((unsigned)((uthreads + 255) / 256)), dim3(256), 0, s, ua);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // namespace mpcx
