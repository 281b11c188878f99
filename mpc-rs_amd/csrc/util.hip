// Shard statistics (what ranks exchange instead of results) and batch re-layout between interleave widths.
#include "mpcx_internal.h"

namespace mpcx {
namespace {

constexpr int kStatBlocks = 1024;
constexpr int kStatThreads = 256;

struct Partial {
    long long non_finite;
    double sum, sum_sq, max_abs;
    unsigned long long checksum;
};

__device__ inline unsigned long long mix64(unsigned long long x) {  // splitmix64 finaliser
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

__device__ inline void combine(Partial &a, const Partial &b) {
    a.non_finite += b.non_finite;
    a.sum += b.sum;
    a.sum_sq += b.sum_sq;
    a.max_abs = fmax(a.max_abs, b.max_abs);
    a.checksum += b.checksum;
}

__device__ inline Partial block_reduce(Partial v) {
    __shared__ Partial sh[kStatThreads];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = kStatThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) combine(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    return sh[0];
}

__global__ __launch_bounds__(kStatThreads) void stats_partial_kernel(const double *buf, long long count, Partial *out) {
    Partial v = {0, 0.0, 0.0, 0.0, 0ULL};
    const long long stride = (long long)gridDim.x * kStatThreads;
    for (long long i = (long long)blockIdx.x * kStatThreads + threadIdx.x; i < count; i += stride) {
        const double x = buf[i];
        if (isfinite(x)) {
            v.sum += x;
            v.sum_sq += x * x;
            v.max_abs = fmax(v.max_abs, fabs(x));
        } else {
            v.non_finite += 1;
        }
        v.checksum += mix64((unsigned long long)__double_as_longlong(x));
    }
    v = block_reduce(v);
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

__global__ __launch_bounds__(kStatThreads) void stats_final_kernel(const Partial *parts, int nparts, long long count,
                                                                   mpcx_stats *out) {
    Partial v = {0, 0.0, 0.0, 0.0, 0ULL};
    for (int i = threadIdx.x; i < nparts; i += kStatThreads) combine(v, parts[i]);
    v = block_reduce(v);
    if (threadIdx.x == 0) {
        out->count = count;
        out->non_finite = v.non_finite;
        out->sum = v.sum;
        out->sum_sq = v.sum_sq;
        out->max_abs = v.max_abs;
        out->checksum = v.checksum;
    }
}

constexpr int kTile = 64;

// dst and src hold the same (batch x E) values under two interleave widths.  A 64 x 64 (instance x element)
// tile goes through LDS so that both the read and the write side run along their own contiguous direction.
// Instance b of the call is instance sb0 + b of the source array and db0 + b of the destination array.
__global__ __launch_bounds__(256) void relayout_kernel(long long batch, long long E, long long Ws, const double *src,
                                                       long long sb0, long long Wd, double *dst, long long db0) {
    __shared__ double tile[kTile][kTile + 1];  // [instance][element]
    const long long b0 = (long long)blockIdx.x * kTile, e0 = (long long)blockIdx.y * kTile;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const bool src_elem_fast = (Ws == 1), dst_elem_fast = (Wd == 1);
    for (int k = ty; k < kTile; k += 4) {
        const long long b = src_elem_fast ? b0 + k : b0 + tx;
        const long long e = src_elem_fast ? e0 + tx : e0 + k;
        if (b < batch && e < E) tile[b - b0][e - e0] = src[elem_index(sb0 + b, e, E, Ws)];
    }
    __syncthreads();
    for (int k = ty; k < kTile; k += 4) {
        const long long b = dst_elem_fast ? b0 + k : b0 + tx;
        const long long e = dst_elem_fast ? e0 + tx : e0 + k;
        if (b < batch && e < E) dst[elem_index(db0 + b, e, E, Wd)] = tile[b - b0][e - e0];
    }
}

}  // namespace

size_t stats_scratch_bytes() { return sizeof(Partial) * kStatBlocks; }

int launch_stats(const double *buf, long long count, mpcx_stats *out_dev, void *scratch, size_t scratch_bytes,
                 hipStream_t s) {
    if (scratch_bytes < stats_scratch_bytes()) return MPCX_ERR_UNSUPPORTED;
    Partial *parts = (Partial *)scratch;
    long long want = (count + kStatThreads - 1) / kStatThreads;
    int blocks = (int)(want < 1 ? 1 : (want > kStatBlocks ? kStatBlocks : want));
    hipLaunchKernelGGL(stats_partial_kernel, dim3(blocks), dim3(kStatThreads), 0, s, buf, count, parts);
    hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(kStatThreads), 0, s, parts, blocks, count, out_dev);
    return (int)hipGetLastError();
}

int launch_relayout(long long batch, long long E, long long Wsrc, const double *src, long long Wdst, double *dst,
                    hipStream_t s, long long src_first, long long dst_first) {
    if (batch == 0 || E == 0) return 0;
    const long long gx = (batch + kTile - 1) / kTile, gy = (E + kTile - 1) / kTile;
    if (gx > 0x7fffffffLL || gy > 65535) return MPCX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(relayout_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, batch, E, Wsrc, src, src_first,
                       Wdst, dst, dst_first);
    return (int)hipGetLastError();
}

}  // namespace mpcx
