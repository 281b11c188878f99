// Condensed-QP build without an on-chip size limit: the streaming form.
//
// The one-workgroup-per-instance kernels (qp_wg.hip) keep an instance's Markov vectors G, S, Q G, Q S (4 m f r doubles)
// and two column blocks of H and W3 in the 160 KB of LDS; Controller::form_lifted_matrices (controller.rs:16-99) has no
// such limit.  Beyond it the same quantities go through a device workspace instead, in three kernels per call:
//
//   markov_kernel   one workgroup per instance, the two O(f) recurrences step by step (w <- A w, P <- P A; A, B, C are
//                   read from memory each step -- they stay in cache -- and only the two states live in LDS):
//                   Phi to the caller; G_k = C A^k B, S_k = G_0 + .. + G_k, Q G, Q S and z = Q (Phi x0 - ref) to the
//                   workspace, one vector per input column: G[c][k r + a] = G_k[a][c].
//   gamma_kernel    one workgroup per column of Gamma: column (k, c) is G[c] shifted down by k r (the last block column:
//                   S[c] shifted by (v-1) r), a copy with lanes along the rows; fvec[(k, c)] = column . z is reduced on
//                   the way (fixed tree, deterministic).
//   hsweep_kernel   H = Gamma^T blockdiag(Q) Gamma + W3 by the block-Toeplitz recurrence (controller.rs:92-93 evaluated as
//                       Ht[a][b] = Ht[a+1][b+1] + G_(f-1-a)^T Q G_(f-1-b)
//                   over the Gram matrix Ht of plain-shift columns bordered by a virtual block row and column of index v,
//                   as in qp_wg.hip).  A thread owns one block DIAGONAL d = b - a and one row i of its blocks: its running
//                   sums never leave its registers, and at sweep step b the lanes of consecutive diagonals write
//                   consecutive rows of column block b -- contiguous runs of a column of H.  With held inputs (f > v) a
//                   thread starts from its border value (a sum of f - v terms) and H's last block row and column, which
//                   involve S, are summed directly by hedge_kernel.
// Instance-major data only (interleaved batches reach it through the instance-major copy, like the workgroup kernels).
#include "mpcx_internal.h"

namespace mpcx {
namespace {

constexpr int kT = 256;

struct StreamWs {
    double *G, *S, *QG, *QS, *z;  // per instance: m f r (x4), f r
};

__device__ __forceinline__ StreamWs ws_of(double *ws, long long li, int m, int FR) {
    const size_t per = (size_t)4 * m * FR + FR;
    double *base = ws + (size_t)li * per;
    return {base, base + (size_t)m * FR, base + (size_t)2 * m * FR, base + (size_t)3 * m * FR, base + (size_t)4 * m * FR};
}

struct StreamArgs {
    QpArgs q;
    double *ws;
    long long first, count;  // instances [first, first + count) of the batch; workspace slot = instance - first
};

__global__ __launch_bounds__(kT) void markov_kernel(StreamArgs p) {
    extern __shared__ double sm[];
    const QpArgs &a = p.q;
    const int n = a.n, m = a.m, r = a.r, f = a.f, FR = f * r, tid = threadIdx.x;
    const long long li = blockIdx.x, b = p.first + li;
    const double *A = a.A + b * (long long)n * n, *B = a.B + b * (long long)n * m, *C = a.C + b * (long long)r * n;
    const bool want_s = a.fvec != nullptr;
    const double *x0 = want_s ? a.x0 + b * (long long)n : nullptr, *ref = want_s ? a.ref + b * (long long)FR : nullptr;
    double *Phi = a.Phi ? a.Phi + b * (long long)FR * n : nullptr;
    const StreamWs w = ws_of(p.ws, li, m, FR);
    // states ping-pong: W[slot][c * n + i], P[slot][j * r + a]; x0; running sums S (entry t belongs to the thread of task t)
    double *sW = sm, *sP = sW + 2 * n * m, *sx = sP + 2 * r * n, *sRun = sx + n;
    for (int e = tid; e < r * m; e += kT) sRun[e] = 0.0;
    for (int e = tid; e < n * m; e += kT) sW[e] = B[e];                     // W_0 = B
    for (int e = tid; e < r * n; e += kT) sP[e] = C[e];                     // P_(-1) = C, column-major r x n
    for (int e = tid; e < n; e += kT) sx[e] = want_s ? x0[e] : 0.0;
    __syncthreads();
    const int nG = r * m, nW = n * m, nP = r * n, nS = want_s ? r : 0, tasks = nG + nW + nP + nS;
    for (int k = 0; k < f; ++k) {
        const double *Wc = sW + (k & 1) * n * m, *Pc = sP + (k & 1) * r * n;
        double *Wn = sW + ((k + 1) & 1) * n * m, *Pn = sP + ((k + 1) & 1) * r * n;
        for (int t = tid; t < tasks; t += kT) {
            if (t < nG) {  // G_k[a][c] = C[a][:] W_k[:, c], and S_k = S_(k-1) + G_k
                const int c = t / r, aa = t - c * r;
                double g = 0.0;
                for (int j = 0; j < n; ++j) g += C[j * r + aa] * Wc[c * n + j];
                w.G[(size_t)c * FR + k * r + aa] = g;
                const double run = sRun[t] + g;
                sRun[t] = run;
                w.S[(size_t)c * FR + k * r + aa] = run;
            } else if (t < nG + nW) {  // W_(k+1) = A W_k
                const int u = t - nG, c = u / n, i = u - c * n;
                double s = 0.0;
                for (int j = 0; j < n; ++j) s += A[(long long)j * n + i] * Wc[c * n + j];
                Wn[c * n + i] = s;
            } else if (t < nG + nW + nP) {  // P_k = P_(k-1) A  (= C A^(k+1), the Phi block k)
                const int u = t - nG - nW, j = u / r, aa = u - j * r;
                double s = 0.0;
                for (int q = 0; q < n; ++q) s += Pc[q * r + aa] * A[(long long)j * n + q];
                Pn[j * r + aa] = s;
                if (Phi) Phi[(long long)j * FR + k * r + aa] = s;
            } else if (k > 0) {  // s_(k-1) = P_(k-1) x0 - ref_(k-1), from the block the previous step produced
                const int aa = t - nG - nW - nP;
                double s = 0.0;
                for (int j = 0; j < n; ++j) s += Pc[j * r + aa] * sx[j];
                w.z[(k - 1) * r + aa] = s - ref[(k - 1) * r + aa];
            }
        }
        __syncthreads();
    }
    if (want_s) {  // the last block's s
        const double *Pc = sP + (f & 1) * r * n;
        for (int aa = tid; aa < r; aa += kT) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += Pc[j * r + aa] * sx[j];
            w.z[(f - 1) * r + aa] = s - ref[(f - 1) * r + aa];
        }
    }
    __syncthreads();  // the workgroup's own writes to the workspace are visible to it from here on
    if (a.H != nullptr) {  // Q G and Q S, element-wise over the vectors just written
        const double *Q = a.W4;  // r x r, column-major
        for (int e = tid; e < m * FR; e += kT) {
            const int c = e / FR, row = e - c * FR, k = row / r, aa = row - k * r;
            double g = 0.0, sq = 0.0;
            for (int q = 0; q < r; ++q) {
                g += Q[q * r + aa] * w.G[(size_t)c * FR + k * r + q];
                sq += Q[q * r + aa] * w.S[(size_t)c * FR + k * r + q];
            }
            w.QG[e] = g;
            w.QS[e] = sq;
        }
    }
}

// z = Q s per block, out of place (s and z are different arrays): s in w.z, z in `zout`
__global__ __launch_bounds__(kT) void zq_kernel(StreamArgs p, double *zout) {
    const QpArgs &a = p.q;
    const int r = a.r, FR = a.f * r;
    const long long idx = (long long)blockIdx.x * kT + threadIdx.x;
    if (idx >= p.count * FR) return;
    const long long li = idx / FR;
    const int row = (int)(idx - li * FR), k = row / r, aa = row - k * r;
    const StreamWs w = ws_of(p.ws, li, a.m, FR);
    double z = 0.0;
    for (int q = 0; q < r; ++q) z += a.W4[q * r + aa] * w.z[k * r + q];
    zout[idx] = z;
}

// One workgroup per (column of Gamma, instance).
__global__ __launch_bounds__(kT) void gamma_kernel(StreamArgs p, const double *zq) {
    __shared__ double red[kT];
    const QpArgs &a = p.q;
    const int m = a.m, r = a.r, f = a.f, v = a.v, FR = f * r, VM = v * m, tid = threadIdx.x;
    const long long li = blockIdx.x / VM, b = p.first + li;
    const int col = (int)(blockIdx.x - li * VM), kb = col / m, c = col - kb * m;
    const StreamWs w = ws_of(p.ws, li, m, FR);
    const double *src = (kb == v - 1 ? w.S : w.G) + (size_t)c * FR;
    const double *z = zq ? zq + li * (long long)FR : nullptr;
    double *gcol = a.Gamma ? a.Gamma + b * (long long)FR * VM + (long long)col * FR : nullptr;
    const int shift = kb * r;
    double acc = 0.0;
    for (int row = tid; row < FR; row += kT) {
        const double val = row >= shift ? src[row - shift] : 0.0;
        if (gcol) gcol[row] = val;
        if (z) acc += val * z[row];
    }
    if (a.fvec) {
        red[tid] = acc;
        __syncthreads();
        for (int s = kT / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) a.fvec[b * (long long)VM + col] = red[0];
    }
}

// Thread (d, i): block diagonal d = bb - aa in -(v-1) .. v-1 and row i of its m x m blocks; acc[j] for every column j
// (JT columns at a time: a launch covers columns j0 .. j0 + JT).  Lanes are ordered so that at a sweep step consecutive
// lanes write consecutive rows: lane index u = (v - 1 - d) m + i  <=>  row aa m + i with aa = bb - d ascending in -d.
template <int JT>
__global__ __launch_bounds__(kT) void hsweep_kernel(StreamArgs p, int j0) {
    const QpArgs &a = p.q;
    const int m = a.m, r = a.r, f = a.f, v = a.v, FR = f * r, VM = v * m;
    const bool held = f > v;
    const int per_inst = (2 * v - 1) * m, blocks_per = (per_inst + kT - 1) / kT;
    const long long li = blockIdx.x / blocks_per, b = p.first + li;
    const int u = (int)(blockIdx.x - li * blocks_per) * kT + threadIdx.x;
    if (u >= per_inst) return;
    const int dd = u / m, i = u - dd * m, d = (v - 1) - dd;  // d descending with the lane index
    const StreamWs w = ws_of(p.ws, li, m, FR);
    const double *Gi = w.G + (size_t)i * FR;
    double *H = a.H + b * (long long)VM * VM;
    double acc[JT];
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) acc[jj] = 0.0;
    // start of the chain: block (aa, bb) = (v-1-d, v-1) for d >= 0, (v-1, v-1+d) for d < 0; its predecessor on the
    // diagonal is the border block (aa+1, bb+1), one of whose indices is v: a sum over the rows i2 = v .. f-1
    const int bb_first = d >= 0 ? v - 1 : v - 1 + d;
    if (held) {
        const int pa = bb_first - d + 1, pb = bb_first + 1;  // border block (pa, pb), max(pa, pb) = v
        for (int i2 = v; i2 < f; ++i2)
            for (int q = 0; q < r; ++q) {
                const double g = Gi[(i2 - pa) * r + q];
#pragma unroll
                for (int jj = 0; jj < JT; ++jj)
                    if (j0 + jj < m) acc[jj] += g * w.QG[(size_t)(j0 + jj) * FR + (i2 - pb) * r + q];
            }
    }
    // every thread walks the same column blocks at the same time (that is what makes a wavefront's stores contiguous);
    // a thread is idle until its diagonal enters the matrix and after it has left
    for (int bb = v - 1; bb >= 0; --bb) {
        const int aa = bb - d;
        if (aa < 0 || aa > v - 1) continue;
        const double *ga = Gi + (f - 1 - aa) * r;
        for (int q = 0; q < r; ++q) {
            const double g = ga[q];
#pragma unroll
            for (int jj = 0; jj < JT; ++jj)
                if (j0 + jj < m) acc[jj] += g * w.QG[(size_t)(j0 + jj) * FR + (f - 1 - bb) * r + q];
        }
        if (held && (aa == v - 1 || bb == v - 1)) continue;  // H's own last block row / column: hedge_kernel
#pragma unroll
        for (int jj = 0; jj < JT; ++jj)
            if (j0 + jj < m) {
                const long long at = (long long)(bb * m + j0 + jj) * VM + aa * m + i;
                H[at] = acc[jj] + a.W3[at];
            }
    }
}

// f > v: H's last block column (which = 0) and last block row (which = 1), where the held-input column S stands in for a
// plain shift: element = sum over i2 >= v-1 of X_a(i2)[:, i] . Q X_b(i2)[:, j], X_(v-1)(i2) = S_(i2-v+1), X_k(i2) = G_(i2-k).
__global__ __launch_bounds__(kT) void hedge_kernel(StreamArgs p) {
    const QpArgs &a = p.q;
    const int m = a.m, r = a.r, f = a.f, v = a.v, FR = f * r, VM = v * m;
    const long long per = 2LL * VM * m, idx = (long long)blockIdx.x * kT + threadIdx.x;
    if (idx >= p.count * per) return;
    const long long li = idx / per, b = p.first + li;
    const int e = (int)(idx - li * per), which = e / (VM * m), rem = e - which * VM * m;
    // which = 0: column (v-1, j), row = rem % VM (lanes along rows); which = 1: row (v-1, i), column = rem / m
    int row, col;
    if (which == 0) { col = (v - 1) * m + rem / VM; row = rem % VM; }
    else            { row = (v - 1) * m + rem % m; col = rem / m; }
    const int aa = row / m, i = row - aa * m, bb = col / m, j = col - bb * m;
    const StreamWs w = ws_of(p.ws, li, m, FR);
    const bool la = aa == v - 1, lb = bb == v - 1;
    const double *xa = (la ? w.S : w.G) + (size_t)i * FR, *xb = (lb ? w.QS : w.QG) + (size_t)j * FR;
    const int sa = la ? v - 1 : aa, sb = lb ? v - 1 : bb;  // X(i2) = vector[(i2 - shift) r ..]
    double acc = 0.0;
    for (int i2 = v - 1; i2 < f; ++i2)
        for (int q = 0; q < r; ++q) acc += xa[(i2 - sa) * r + q] * xb[(i2 - sb) * r + q];
    const long long at = (long long)col * VM + row;
    a.H[b * (long long)VM * VM + at] = acc + a.W3[at];
}

}  // namespace

// Doubles of workspace per instance: G, S, Q G, Q S (m f r each), s = Phi x0 - ref and z = Q s (f r each).
size_t qp_stream_ws_doubles(const QpArgs &a) { return (size_t)4 * a.m * a.f * a.r + (size_t)2 * a.f * a.r; }

// The shapes the streaming form takes: block weights, instance-major, the two recurrence states in LDS.
bool qp_stream_takes(const QpArgs &a) {
    if (a.W != 1) return false;
    if (a.weights_mode != MPCX_WEIGHTS_BLOCK && (a.H || a.fvec)) return false;
    const size_t lds = ((size_t)2 * a.n * a.m + (size_t)2 * a.r * a.n + a.n + (size_t)a.r * a.m) * sizeof(double);
    return lds <= 160 * 1024 && (long long)a.f * a.r * a.v * a.m <= (1LL << 40);
}

int launch_qp_stream(const QpArgs &a, double *ws, size_t ws_doubles, hipStream_t s) {
    if (!qp_stream_takes(a)) return MPCX_ERR_UNSUPPORTED;
    const int m = a.m, FR = a.f * a.r, VM = a.v * a.m;
    const size_t per = qp_stream_ws_doubles(a);
    const long long chunk = (long long)(ws_doubles / per);
    if (chunk < 1) return MPCX_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)2 * a.n * a.m + (size_t)2 * a.r * a.n + a.n + (size_t)a.r * a.m) * sizeof(double);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)markov_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    for (long long first = 0; first < a.batch; first += chunk) {
        const long long count = a.batch - first < chunk ? a.batch - first : chunk;
        // this launch's instances use the first `count` slots: G S QG QS s per slot (4 m FR + FR), z behind all slots
        StreamArgs p{a, ws, first, count};
        double *zq = ws + (size_t)count * ((size_t)4 * m * FR + FR);
        if (count > 0x7fffffffLL || count * (long long)VM > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(markov_kernel, dim3((unsigned)count), dim3(kT), lds, s, p);
        if (a.fvec) {
            const long long zb = (count * FR + kT - 1) / kT;
            if (zb > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
            hipLaunchKernelGGL(zq_kernel, dim3((unsigned)zb), dim3(kT), 0, s, p, zq);
        }
        if (a.Gamma || a.fvec)
            hipLaunchKernelGGL(gamma_kernel, dim3((unsigned)(count * VM)), dim3(kT), 0, s, p, a.fvec ? zq : nullptr);
        if (a.H) {
            const long long blocks_per = ((2LL * a.v - 1) * m + kT - 1) / kT;
            if (count * blocks_per > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
            for (int j0 = 0; j0 < m; j0 += 4)
                hipLaunchKernelGGL(hsweep_kernel<4>, dim3((unsigned)(count * blocks_per)), dim3(kT), 0, s, p, j0);
            if (a.f > a.v) {
                const long long eb = (count * 2LL * VM * m + kT - 1) / kT;
                if (eb > 0x7fffffffLL) return MPCX_ERR_UNSUPPORTED;
                hipLaunchKernelGGL(hedge_kernel, dim3((unsigned)eb), dim3(kT), 0, s, p);
            }
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // namespace mpcx
