// C ABI of libmpcx: argument checking, kernel-family dispatch, and the host-pointer convenience paths.
// There is no CPU implementation behind these entry points.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "mpcx_internal.h"

struct mpcx_ctx {
    int device;
    hipStream_t stream;
    void *work;          // staging for host-pointer calls
    size_t work_bytes;
    void *lu_work;       // augmented matrices of the blocked solver (systems beyond 64 unknowns)
    size_t lu_bytes;
    void *aux;           // I - Ac dt of the general discretisation; the solutions H^-T e_i of the first-move gain rows
    size_t aux_bytes;
    void *stat_scratch;  // partial records of the stats reduction + one result record
    void *relay;         // instance-major copy of a chunk of the batch (interleaved data on the workgroup kernels)
    size_t relay_bytes;
    void *qp_ws;         // Markov vectors of the streaming build (shapes beyond the on-chip kernels)
    size_t qp_ws_bytes;
    void *w3win;         // band structure of W3, measured on the device by every build call of the workgroup kernels
    size_t w3win_bytes;
    size_t relay_cap;    // MPCX_OPT_RELAY_CAP_BYTES
    int force_stream;    // MPCX_OPT_QP_STREAMING
    char err[512];
};

namespace {

using namespace mpcx;

int fail(mpcx_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

int hip_fail(mpcx_ctx *ctx, hipError_t e, const char *what) {
    return fail(ctx, (int)e, "%s: %s", what, hipGetErrorString(e));
}

#define HIP_TRY(ctx, call)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

int check_common(mpcx_ctx *ctx, const mpcx_dims *d, long long *W) {
    if (!ctx) return MPCX_ERR_NULL;
    if (!d) return fail(ctx, MPCX_ERR_NULL, "dims is NULL");
    if (d->n < 1 || d->m < 1 || d->r < 1) return fail(ctx, MPCX_ERR_DIMS, "n, m, r must be >= 1 (got %d, %d, %d)", d->n, d->m, d->r);
    if (d->batch < 0) return fail(ctx, MPCX_ERR_DIMS, "batch must be >= 0");
    if (d->interleave < 0) return fail(ctx, MPCX_ERR_LAYOUT, "interleave must be >= 0");
    long long w = d->interleave == 0 ? d->batch : d->interleave;
    if (w < 1) w = 1;
    if (w != 1 && w != d->batch && (w % 64) != 0)
        return fail(ctx, MPCX_ERR_LAYOUT, "interleave %lld: must be 0 (= batch), 1, or a multiple of 64", (long long)d->interleave);
    *W = w;
    return MPCX_OK;
}

// Workspace of the blocked solver: room for the whole batch when that is at most kLuCap, else for as many instances
// as kLuCap holds (the solver then runs the batch in several launches); never less than one instance.
constexpr size_t kLuCap = (size_t)64 << 30;
int ensure_lu_work(mpcx_ctx *ctx, int nr, int nrhs, long long batch, size_t *doubles) {
    *doubles = 0;
    if (solve_fits_on_chip(nr, nrhs) || !solve_large_supports(nr)) return MPCX_OK;
    const size_t per = solve_large_ws_doubles(nr, nrhs) * sizeof(double);
    size_t want = per * (size_t)batch;
    if (want > kLuCap) want = (kLuCap / per ? kLuCap / per : 1) * per;
    if (want > ctx->lu_bytes) {
        if (ctx->lu_work) {
            HIP_TRY(ctx, hipFree(ctx->lu_work));
            ctx->lu_work = nullptr;
            ctx->lu_bytes = 0;
        }
        HIP_TRY(ctx, hipMalloc(&ctx->lu_work, want));
        ctx->lu_bytes = want;
    }
    *doubles = ctx->lu_bytes / sizeof(double);
    return MPCX_OK;
}

int ensure_work(mpcx_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->work_bytes) return MPCX_OK;
    if (ctx->work) {
        HIP_TRY(ctx, hipFree(ctx->work));
        ctx->work = nullptr;
        ctx->work_bytes = 0;
    }
    HIP_TRY(ctx, hipMalloc(&ctx->work, bytes));
    ctx->work_bytes = bytes;
    return MPCX_OK;
}

// Workspace of the streaming build: the whole batch when that is at most kQpWsCap, else as many instances as the cap
// holds (the build then runs in several launches); never less than one instance.
constexpr size_t kQpWsCap = (size_t)8 << 30;
int ensure_qp_ws(mpcx_ctx *ctx, const QpArgs &a) {
    const size_t per = qp_stream_ws_doubles(a) * sizeof(double);
    size_t want = per * (size_t)a.batch;
    if (want > kQpWsCap) want = (kQpWsCap / per ? kQpWsCap / per : 1) * per;
    if (want <= ctx->qp_ws_bytes) return MPCX_OK;
    if (ctx->qp_ws) {
        HIP_TRY(ctx, hipFree(ctx->qp_ws));
        ctx->qp_ws = nullptr;
        ctx->qp_ws_bytes = 0;
    }
    HIP_TRY(ctx, hipMalloc(&ctx->qp_ws, want));
    ctx->qp_ws_bytes = want;
    return MPCX_OK;
}

int ensure_aux(mpcx_ctx *ctx, size_t need) {
    if (need <= ctx->aux_bytes) return MPCX_OK;
    if (ctx->aux) {
        HIP_TRY(ctx, hipFree(ctx->aux));
        ctx->aux = nullptr;
        ctx->aux_bytes = 0;
    }
    HIP_TRY(ctx, hipMalloc(&ctx->aux, need));
    ctx->aux_bytes = need;
    return MPCX_OK;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct QpSizes {
    long long A, B, C, x0, ref, W3, W4, Phi, Gamma, H, fvec;  // elements
};

QpSizes qp_sizes(const mpcx_dims *d) {
    const long long n = d->n, m = d->m, r = d->r, f = d->f, v = d->v, bt = d->batch, il = d->interleave;
    QpSizes s;
    s.A = mpcx_array_elems(bt, il, n * n);
    s.B = mpcx_array_elems(bt, il, n * m);
    s.C = mpcx_array_elems(bt, il, r * n);
    s.x0 = mpcx_array_elems(bt, il, n);
    s.ref = mpcx_array_elems(bt, il, f * r);
    s.W3 = v * m * v * m;
    s.W4 = d->weights_mode == MPCX_WEIGHTS_DENSE ? f * r * f * r : r * r;
    s.Phi = mpcx_array_elems(bt, il, f * r * n);
    s.Gamma = mpcx_array_elems(bt, il, f * r * v * m);
    s.H = mpcx_array_elems(bt, il, v * m * v * m);
    s.fvec = mpcx_array_elems(bt, il, v * m);
    return s;
}

}  // namespace

namespace mpcx {
int ctx_device(const mpcx_ctx *ctx) { return ctx->device; }
}  // namespace mpcx

extern "C" {

int mpcx_version(void) { return MPCX_VERSION; }

int64_t mpcx_array_elems(int64_t batch, int64_t interleave, int64_t elems_per_instance) {
    if (batch <= 0 || elems_per_instance <= 0) return 0;
    const int64_t W = interleave == 0 ? batch : interleave;
    if (W < 1) return 0;
    const int64_t tiles = (batch + W - 1) / W;
    return tiles * W * elems_per_instance;
}

int mpcx_create(mpcx_ctx **out, int device_id) {
    if (!out) return MPCX_ERR_NULL;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) return MPCX_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return MPCX_ERR_NO_DEVICE;
    mpcx_ctx *ctx = new (std::nothrow) mpcx_ctx();
    if (!ctx) return MPCX_ERR_NULL;
    ctx->device = device_id;
    ctx->work = nullptr;
    ctx->work_bytes = 0;
    ctx->lu_work = nullptr;
    ctx->lu_bytes = 0;
    ctx->aux = nullptr;
    ctx->aux_bytes = 0;
    ctx->stat_scratch = nullptr;
    ctx->w3win = nullptr;
    ctx->w3win_bytes = 0;
    ctx->relay = nullptr;
    ctx->relay_bytes = 0;
    ctx->qp_ws = nullptr;
    ctx->qp_ws_bytes = 0;
    ctx->relay_cap = (size_t)2 << 30;
    ctx->force_stream = 0;
    ctx->err[0] = 0;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return (int)e;
    }
    e = hipMalloc(&ctx->stat_scratch, stats_scratch_bytes() + 256);
    if (e == hipSuccess) {  // band structure of W3, measured on the device by every build call of the workgroup kernels
        e = hipMalloc(&ctx->w3win, 256);
        ctx->w3win_bytes = 256;
    }
    if (e == hipSuccess) {
        // the allocation-free discretisation works in a static device array: resolve its address on this device now
        double *p = nullptr;
        const int rc = c2d_static_scratch(device_id, &p);
        if (rc > 0) e = (hipError_t)rc;  // a device index beyond the table only disables that entry point
    }
    if (e != hipSuccess) {
        if (ctx->stat_scratch) (void)hipFree(ctx->stat_scratch);
        if (ctx->w3win) (void)hipFree(ctx->w3win);
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return (int)e;
    }
    *out = ctx;
    return MPCX_OK;
}

void mpcx_destroy(mpcx_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->work) (void)hipFree(ctx->work);
    if (ctx->lu_work) (void)hipFree(ctx->lu_work);
    if (ctx->aux) (void)hipFree(ctx->aux);
    if (ctx->stat_scratch) (void)hipFree(ctx->stat_scratch);
    if (ctx->w3win) (void)hipFree(ctx->w3win);
    if (ctx->relay) (void)hipFree(ctx->relay);
    if (ctx->qp_ws) (void)hipFree(ctx->qp_ws);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *mpcx_last_error(const mpcx_ctx *ctx) { return ctx ? ctx->err : "context is NULL"; }

int mpcx_set_option(mpcx_ctx *ctx, int option, int64_t value) {
    if (!ctx) return MPCX_ERR_NULL;
    if (option == MPCX_OPT_RELAY_CAP_BYTES) {
        if (value < 1) return fail(ctx, MPCX_ERR_DIMS, "MPCX_OPT_RELAY_CAP_BYTES must be >= 1");
        ctx->relay_cap = (size_t)value;
        return MPCX_OK;
    }
    if (option == MPCX_OPT_QP_STREAMING) {
        if (value != 0 && value != 1) return fail(ctx, MPCX_ERR_DIMS, "MPCX_OPT_QP_STREAMING must be 0 or 1");
        ctx->force_stream = (int)value;
        return MPCX_OK;
    }
    return fail(ctx, MPCX_ERR_DIMS, "unknown option %d", option);
}

// Kernel-family dispatch of one build on prepared arguments.  The dense-weight kernel, the matrix-core contraction and
// the workgroup kernels take instance-major data (a.W == 1); the caller re-lays interleaved batches out for them.
static int qp_build_core(mpcx_ctx *ctx, const mpcx_dims *d, QpArgs a, hipStream_t s) {
    int rc;
    if (d->weights_mode == MPCX_WEIGHTS_DENSE && (a.H || a.fvec)) {
        // dense W4: H and fvec from the column-by-column kernel, Phi and Gamma from the build kernels
        rc = launch_qp_dense(a, s);
        if (rc == MPCX_ERR_UNSUPPORTED)
            return fail(ctx, rc, "weights_mode = DENSE: n=%d m=%d r=%d f=%d v=%d exceeds the kernel's on-chip working set",
                        d->n, d->m, d->r, d->f, d->v);
        if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "qp_dense launch");
        a.H = nullptr;
        a.fvec = nullptr;
        if (!a.Phi && !a.Gamma) return MPCX_OK;
    }
    if (a.H && d->hessian_algo == MPCX_HESSIAN_CONTRACT) {
        // dense contraction on the matrix cores: H from its own kernel, everything else from the build kernels
        rc = launch_hessian_mfma(a, s);
        if (rc == MPCX_ERR_UNSUPPORTED)
            return fail(ctx, rc, "hessian_algo = CONTRACT: n=%d m=%d r=%d f=%d v=%d weights_mode=%d exceeds the kernel's on-chip working set",
                        d->n, d->m, d->r, d->f, d->v, d->weights_mode);
        if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "hessian_mfma launch");
        a.H = nullptr;
        if (!a.Phi && !a.Gamma && !a.fvec) return MPCX_OK;
    }
    rc = (ctx->force_stream && qp_stream_takes(a)) ? MPCX_ERR_UNSUPPORTED : launch_qp_lane(a, s);
    if (rc == MPCX_ERR_UNSUPPORTED && !(ctx->force_stream && qp_stream_takes(a))) rc = launch_qp_wg(a, d->hessian_algo, s);
    if (rc == MPCX_ERR_UNSUPPORTED && qp_stream_takes(a)) {  // beyond the on-chip working set: the streaming form
        const int wrc = ensure_qp_ws(ctx, a);
        if (wrc != MPCX_OK) return wrc;
        rc = launch_qp_stream(a, (double *)ctx->qp_ws, ctx->qp_ws_bytes / sizeof(double), s);
    }
    if (rc == MPCX_ERR_LAYOUT)
        return fail(ctx, rc, "n=%d m=%d r=%d needs the workgroup kernels, which take instance-major data (interleave = 1)",
                    d->n, d->m, d->r);
    if (rc == MPCX_ERR_UNSUPPORTED)
        return fail(ctx, rc, "n=%d m=%d r=%d f=%d v=%d weights_mode=%d does not fit the on-chip working set of any kernel",
                    d->n, d->m, d->r, d->f, d->v, d->weights_mode);
    if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "qp_build launch");
    return MPCX_OK;
}

// Interleaved batch on kernels that take instance-major data: the batch goes through an instance-major copy in chunks
// of at most ctx->relay_cap bytes (inputs re-laid out into the context's scratch, built there, outputs re-laid out into the
// caller's arrays).  A compatibility path: it moves every output three times.
static int ensure_relay(mpcx_ctx *ctx, size_t want) {
    if (want <= ctx->relay_bytes) return MPCX_OK;
    if (ctx->relay) {
        HIP_TRY(ctx, hipFree(ctx->relay));
        ctx->relay = nullptr;
        ctx->relay_bytes = 0;
    }
    HIP_TRY(ctx, hipMalloc(&ctx->relay, want));
    ctx->relay_bytes = want;
    return MPCX_OK;
}

// instances per chunk of the instance-major copy: what the cap holds, in whole groups of 64, at least one
static long long relay_chunk(const mpcx_ctx *ctx, long long per_instance_doubles, long long batch) {
    long long cnt = (long long)(ctx->relay_cap / ((size_t)per_instance_doubles * sizeof(double)));
    if (cnt < 1) cnt = 1;
    if (cnt >= 64) cnt &= ~63LL;
    return cnt > batch ? batch : cnt;
}

static int qp_build_relayed(mpcx_ctx *ctx, const mpcx_dims *d, const QpArgs &a, hipStream_t s) {
    const long long n = a.n, m = a.m, r = a.r, FR = (long long)a.f * a.r, VM = (long long)a.v * a.m;
    const double *in[5] = {a.A, a.B, a.C, a.fvec ? a.x0 : nullptr, a.fvec ? a.ref : nullptr};
    const long long in_e[5] = {n * n, n * m, r * n, n, FR};
    double *out[4] = {a.Phi, a.Gamma, a.H, a.fvec};
    const long long out_e[4] = {FR * n, FR * VM, VM * VM, VM};
    long long per = 0;
    for (int i = 0; i < 5; ++i) per += in[i] ? in_e[i] : 0;
    for (int i = 0; i < 4; ++i) per += out[i] ? out_e[i] : 0;
    const long long cnt = relay_chunk(ctx, per, a.batch);
    {
        const int rc = ensure_relay(ctx, (size_t)cnt * (size_t)per * sizeof(double));
        if (rc != MPCX_OK) return rc;
    }
    for (long long c0 = 0; c0 < a.batch; c0 += cnt) {
        const long long cc = a.batch - c0 < cnt ? a.batch - c0 : cnt;
        double *w = (double *)ctx->relay;
        const double *win[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        double *wout[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < 5; ++i) {
            if (!in[i]) continue;
            const int rc = launch_relayout(cc, in_e[i], a.W, in[i], 1, w, s, c0, 0);
            if (rc == MPCX_ERR_UNSUPPORTED) return fail(ctx, rc, "interleaved batch: an input array is too large to re-lay out");
            if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "relayout launch");
            win[i] = w;
            w += cc * in_e[i];
        }
        for (int i = 0; i < 4; ++i) {
            if (!out[i]) continue;
            wout[i] = w;
            w += cc * out_e[i];
        }
        QpArgs c = a;
        c.A = win[0]; c.B = win[1]; c.C = win[2]; c.x0 = win[3]; c.ref = win[4];
        c.Phi = wout[0]; c.Gamma = wout[1]; c.H = wout[2]; c.fvec = wout[3];
        c.batch = cc; c.W = 1;
        int rc = qp_build_core(ctx, d, c, s);
        if (rc != MPCX_OK) return rc;
        for (int i = 0; i < 4; ++i) {
            if (!out[i]) continue;
            rc = launch_relayout(cc, out_e[i], 1, wout[i], a.W, out[i], s, 0, c0);
            if (rc == MPCX_ERR_UNSUPPORTED)
                return fail(ctx, rc, "interleaved batch: %lld elements per instance are too many to re-lay out (use interleave = 1)",
                            out_e[i]);
            if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "relayout launch");
        }
    }
    return MPCX_OK;
}

int mpcx_reserve(mpcx_ctx *ctx, const mpcx_dims *d) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->f < 1 || d->v < 1 || d->v > d->f)
        return fail(ctx, MPCX_ERR_DIMS, "need f >= 1 and 1 <= v <= f (got f=%d, v=%d)", d->f, d->v);
    if (d->batch == 0) return MPCX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t wsd = 0;
    // blocked solver: solve (one right-hand side), gain (f r of them), general discretisation (n x n)
    if ((rc = ensure_lu_work(ctx, d->v * d->m, 1, d->batch, &wsd)) != MPCX_OK) return rc;
    if ((rc = ensure_lu_work(ctx, d->v * d->m, d->f * d->r, d->batch, &wsd)) != MPCX_OK) return rc;
    if ((rc = ensure_lu_work(ctx, d->n, d->n, d->batch, &wsd)) != MPCX_OK) return rc;
    if ((rc = ensure_lu_work(ctx, d->v * d->m, d->m, d->batch, &wsd)) != MPCX_OK) return rc;  // first-move gain rows
    if ((rc = ensure_aux(ctx, (size_t)mpcx_array_elems(d->batch, d->interleave, (int64_t)d->v * d->m * d->m) * sizeof(double))) != MPCX_OK)
        return rc;
    QpArgs a{};
    a.n = d->n; a.m = d->m; a.r = d->r; a.f = d->f; a.v = d->v; a.batch = d->batch; a.W = W; a.weights_mode = d->weights_mode;
    if (!c2d_lane_supports(d->n, d->m)) {  // I - Ac dt of the general discretisation
        const size_t need = (size_t)mpcx_array_elems(d->batch, d->interleave, (int64_t)d->n * d->n) * sizeof(double);
        if ((rc = ensure_aux(ctx, need)) != MPCX_OK) return rc;
    }
    {  // shapes beyond the on-chip build kernels: the streaming build's workspace
        QpArgs w = a;
        w.W = 1;
        double dummy = 0.0;
        w.H = &dummy;  // (sizes do not depend on which outputs are asked for)
        if (!qp_lane_takes(a) && !qp_wg_takes(w) && qp_stream_takes(w)) {
            if (W != 1) {  // reached through the instance-major copy, a chunk at a time: bound the chunk by the inputs alone
                const long long n = d->n, m = d->m, r = d->r;
                w.batch = relay_chunk(ctx, n * n + n * m + r * n, d->batch);
            }
            if ((rc = ensure_qp_ws(ctx, w)) != MPCX_OK) return rc;
        }
    }
    if (W != 1 && (!qp_lane_takes(a) || d->hessian_algo == MPCX_HESSIAN_CONTRACT)) {  // instance-major copy of a build
        const long long n = d->n, m = d->m, r = d->r, FR = (long long)d->f * r, VM = (long long)d->v * m;
        const long long per = n * n + n * m + r * n + n + FR + FR * n + FR * VM + VM * VM + VM;
        // A later build may ask for fewer outputs: its per-instance size is smaller, its chunk holds more instances and
        // chunk x size can exceed what the all-outputs chunk takes (both stay under the cap, or one instance where one
        // instance exceeds the cap).  Reserve the bound that covers every subset: min(max(cap, one instance), whole batch).
        const size_t one = (size_t)per * sizeof(double), whole = one * (size_t)d->batch;
        const size_t bound = ctx->relay_cap > one ? ctx->relay_cap : one;
        if ((rc = ensure_relay(ctx, bound < whole ? bound : whole)) != MPCX_OK) return rc;
    }
    return MPCX_OK;
}

int mpcx_qp_build_f64_dev(mpcx_ctx *ctx, const mpcx_dims *d, const double *A, const double *B, const double *C,
                          const double *W3, const double *W4, const double *x0, const double *ref, double *Phi,
                          double *Gamma, double *H, double *fvec, void *stream) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->f < 1 || d->v < 1 || d->v > d->f)
        return fail(ctx, MPCX_ERR_DIMS, "need f >= 1 and 1 <= v <= f (got f=%d, v=%d)", d->f, d->v);
    if (d->weights_mode != MPCX_WEIGHTS_BLOCK && d->weights_mode != MPCX_WEIGHTS_DENSE)
        return fail(ctx, MPCX_ERR_DIMS, "unknown weights_mode %d", d->weights_mode);
    if (d->hessian_algo != MPCX_HESSIAN_AUTO && d->hessian_algo != MPCX_HESSIAN_CONTRACT &&
        d->hessian_algo != MPCX_HESSIAN_STRUCTURED)
        return fail(ctx, MPCX_ERR_DIMS, "unknown hessian_algo %d", d->hessian_algo);
    if (d->batch == 0) return MPCX_OK;
    if (!A || !B || !C) return fail(ctx, MPCX_ERR_NULL, "A, B and C are required");
    if ((H || fvec) && !W4) return fail(ctx, MPCX_ERR_NULL, "W4 is required when H or fvec is requested");
    if (H && !W3) return fail(ctx, MPCX_ERR_NULL, "W3 is required when H is requested");
    if (fvec && (!x0 || !ref)) return fail(ctx, MPCX_ERR_NULL, "x0 and ref are required when fvec is requested");
    if (!Phi && !Gamma && !H && !fvec) return MPCX_OK;

    QpArgs a;
    a.A = A; a.B = B; a.C = C; a.W3 = W3; a.W4 = W4; a.x0 = x0; a.ref = ref;
    a.Phi = Phi; a.Gamma = Gamma; a.H = H; a.fvec = fvec;
    a.batch = d->batch; a.W = W;
    a.n = d->n; a.m = d->m; a.r = d->r; a.f = d->f; a.v = d->v;
    a.weights_mode = d->weights_mode;
    hipStream_t s = (hipStream_t)stream;

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    a.w3win = nullptr;
    a.w3win_bytes = 0;
    if (H) {  // where the workgroup kernels measure the band structure of W3
        a.w3win = ctx->w3win;
        a.w3win_bytes = ctx->w3win_bytes;
    }
    // only the one-lane-per-instance kernels read and write interleaved batches directly
    const bool lane_all = qp_lane_takes(a) && !(H && d->hessian_algo == MPCX_HESSIAN_CONTRACT) && !ctx->force_stream;
    if (W != 1 && !lane_all) return qp_build_relayed(ctx, d, a, s);
    return qp_build_core(ctx, d, a, s);
}

int mpcx_qp_build_f64(mpcx_ctx *ctx, const mpcx_dims *d, const double *A, const double *B, const double *C,
                      const double *W3, const double *W4, const double *x0, const double *ref, double *Phi,
                      double *Gamma, double *H, double *fvec) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->f < 1 || d->v < 1 || d->v > d->f)
        return fail(ctx, MPCX_ERR_DIMS, "need f >= 1 and 1 <= v <= f (got f=%d, v=%d)", d->f, d->v);
    if (d->batch == 0) return MPCX_OK;
    if (!A || !B || !C) return fail(ctx, MPCX_ERR_NULL, "A, B and C are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const QpSizes z = qp_sizes(d);
    const double *hin[7] = {A, B, C, W3, W4, x0, ref};
    const long long nin[7] = {z.A, z.B, z.C, z.W3, z.W4, z.x0, z.ref};
    double *hout[4] = {Phi, Gamma, H, fvec};
    const long long nout[4] = {z.Phi, z.Gamma, z.H, z.fvec};
    size_t off_in[7], off_out[4], total = 0;
    for (int i = 0; i < 7; ++i) {
        off_in[i] = total;
        if (hin[i]) total += align256((size_t)nin[i] * sizeof(double));
    }
    for (int i = 0; i < 4; ++i) {
        off_out[i] = total;
        if (hout[i]) total += align256((size_t)nout[i] * sizeof(double));
    }
    rc = ensure_work(ctx, total);
    if (rc != MPCX_OK) return rc;
    char *base = (char *)ctx->work;
    const double *din[7];
    double *dout[4];
    for (int i = 0; i < 7; ++i) {
        din[i] = hin[i] ? (const double *)(base + off_in[i]) : nullptr;
        if (hin[i])
            HIP_TRY(ctx, hipMemcpyAsync((void *)din[i], hin[i], (size_t)nin[i] * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    for (int i = 0; i < 4; ++i) dout[i] = hout[i] ? (double *)(base + off_out[i]) : nullptr;
    rc = mpcx_qp_build_f64_dev(ctx, d, din[0], din[1], din[2], din[3], din[4], din[5], din[6], dout[0], dout[1], dout[2],
                               dout[3], (void *)ctx->stream);
    if (rc != MPCX_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    for (int i = 0; i < 4; ++i)
        if (hout[i])
            HIP_TRY(ctx, hipMemcpyAsync(hout[i], dout[i], (size_t)nout[i] * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MPCX_OK;
}

int mpcx_rollout_f64_dev(mpcx_ctx *ctx, const mpcx_dims *d, const double *A, const double *B, const double *C,
                         const double *x0, const double *U, double *X, double *Y, void *stream) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->T < 0) return fail(ctx, MPCX_ERR_DIMS, "T must be >= 0");
    if (d->batch == 0) return MPCX_OK;
    if (!A || !B || !C || !x0) return fail(ctx, MPCX_ERR_NULL, "A, B, C and x0 are required");
    if (d->T > 0 && !U) return fail(ctx, MPCX_ERR_NULL, "U is required when T > 0");
    if (!X && !Y) return MPCX_OK;
    RolloutArgs a;
    a.A = A; a.B = B; a.C = C; a.x0 = x0; a.U = U; a.X = X; a.Y = Y;
    a.batch = d->batch; a.W = W; a.T = d->T;
    a.n = d->n; a.m = d->m; a.r = d->r; a.u_shared = d->u_shared;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rc = (W == 1) ? MPCX_ERR_UNSUPPORTED : launch_rollout_lane(a, s);
    if (rc == MPCX_ERR_UNSUPPORTED) rc = launch_rollout_group(a, s);
    if (rc == MPCX_ERR_UNSUPPORTED && W == 1) rc = launch_rollout_lane(a, s);
    if (rc == MPCX_ERR_UNSUPPORTED)
        return fail(ctx, rc, "n=%d m=%d r=%d exceeds the rollout kernels' limit of 64 rows", d->n, d->m, d->r);
    if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "rollout launch");
    return MPCX_OK;
}

int mpcx_rollout_f64(mpcx_ctx *ctx, const mpcx_dims *d, const double *A, const double *B, const double *C,
                     const double *x0, const double *U, double *X, double *Y) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->T < 0) return fail(ctx, MPCX_ERR_DIMS, "T must be >= 0");
    if (d->batch == 0) return MPCX_OK;
    if (!A || !B || !C || !x0) return fail(ctx, MPCX_ERR_NULL, "A, B, C and x0 are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long long n = d->n, m = d->m, r = d->r, bt = d->batch, il = d->interleave, T = d->T;
    const double *hin[5] = {A, B, C, x0, U};
    const long long nin[5] = {mpcx_array_elems(bt, il, n * n), mpcx_array_elems(bt, il, n * m), mpcx_array_elems(bt, il, r * n),
                              mpcx_array_elems(bt, il, n), d->u_shared ? m * T : mpcx_array_elems(bt, il, m * T)};
    double *hout[2] = {X, Y};
    const long long nout[2] = {mpcx_array_elems(bt, il, n * (T + 1)), mpcx_array_elems(bt, il, r * T)};
    size_t off_in[5], off_out[2], total = 0;
    for (int i = 0; i < 5; ++i) {
        off_in[i] = total;
        if (hin[i]) total += align256((size_t)nin[i] * sizeof(double));
    }
    for (int i = 0; i < 2; ++i) {
        off_out[i] = total;
        if (hout[i]) total += align256((size_t)nout[i] * sizeof(double));
    }
    rc = ensure_work(ctx, total ? total : 256);
    if (rc != MPCX_OK) return rc;
    char *base = (char *)ctx->work;
    const double *din[5];
    double *dout[2];
    for (int i = 0; i < 5; ++i) {
        din[i] = hin[i] ? (const double *)(base + off_in[i]) : nullptr;
        if (hin[i] && nin[i] > 0)
            HIP_TRY(ctx, hipMemcpyAsync((void *)din[i], hin[i], (size_t)nin[i] * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    for (int i = 0; i < 2; ++i) dout[i] = hout[i] ? (double *)(base + off_out[i]) : nullptr;
    rc = mpcx_rollout_f64_dev(ctx, d, din[0], din[1], din[2], din[3], din[4], dout[0], dout[1], (void *)ctx->stream);
    if (rc != MPCX_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    for (int i = 0; i < 2; ++i)
        if (hout[i] && nout[i] > 0)
            HIP_TRY(ctx, hipMemcpyAsync(hout[i], dout[i], (size_t)nout[i] * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MPCX_OK;
}

namespace {
// Host-pointer round trip shared by the host variants: inputs up, outputs and the optional info array down.
struct HostIo {
    const double *hin[8];
    long long nin[8];
    int ni;
    double *hout[4];
    long long nout[4];
    int no;
    int32_t *hinfo;
    long long ninfo;
    const double *din[8];
    double *dout[4];
    int32_t *dinfo;
};
int host_stage(mpcx_ctx *ctx, HostIo &io) {
    size_t off_in[8], off_out[4], total = 0;
    for (int i = 0; i < io.ni; ++i) { off_in[i] = total; if (io.hin[i]) total += align256((size_t)io.nin[i] * sizeof(double)); }
    for (int i = 0; i < io.no; ++i) { off_out[i] = total; if (io.hout[i]) total += align256((size_t)io.nout[i] * sizeof(double)); }
    const size_t off_info = total;
    if (io.hinfo) total += align256((size_t)io.ninfo * sizeof(int32_t));
    int rc = ensure_work(ctx, total ? total : 256);
    if (rc != MPCX_OK) return rc;
    char *base = (char *)ctx->work;
    for (int i = 0; i < io.ni; ++i) {
        io.din[i] = io.hin[i] ? (const double *)(base + off_in[i]) : nullptr;
        if (io.hin[i] && io.nin[i] > 0)
            HIP_TRY(ctx, hipMemcpyAsync((void *)io.din[i], io.hin[i], (size_t)io.nin[i] * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    for (int i = 0; i < io.no; ++i) io.dout[i] = io.hout[i] ? (double *)(base + off_out[i]) : nullptr;
    io.dinfo = io.hinfo ? (int32_t *)(base + off_info) : nullptr;
    return MPCX_OK;
}
int host_collect(mpcx_ctx *ctx, HostIo &io, int rc) {
    if (rc != MPCX_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    for (int i = 0; i < io.no; ++i)
        if (io.hout[i] && io.nout[i] > 0)
            HIP_TRY(ctx, hipMemcpyAsync(io.hout[i], io.dout[i], (size_t)io.nout[i] * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (io.hinfo) HIP_TRY(ctx, hipMemcpyAsync(io.hinfo, io.dinfo, (size_t)io.ninfo * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MPCX_OK;
}
}  // namespace

int mpcx_qp_solve_f64_dev(mpcx_ctx *ctx, const mpcx_dims *d, const double *H, const double *fvec, double *u_seq,
                          int32_t *info, void *stream) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->v < 1) return fail(ctx, MPCX_ERR_DIMS, "need v >= 1");
    if (d->batch == 0) return MPCX_OK;
    if (!H || !fvec || !u_seq) return fail(ctx, MPCX_ERR_NULL, "H, fvec and u_seq are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t wsd = 0;
    rc = ensure_lu_work(ctx, d->v * d->m, 1, d->batch, &wsd);
    if (rc != MPCX_OK) return rc;
    rc = launch_qp_solve(H, fvec, u_seq, info, d->batch, W, d->v * d->m, (double *)ctx->lu_work, wsd, (hipStream_t)stream);
    if (rc == MPCX_ERR_UNSUPPORTED) return fail(ctx, rc, "solve: v m = %d is beyond what the solver holds on chip", d->v * d->m);
    if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "solve launch");
    return MPCX_OK;
}

int mpcx_qp_gain_f64_dev(mpcx_ctx *ctx, const mpcx_dims *d, const double *H, const double *Gamma, const double *W4,
                         double *gain, int32_t *info, void *stream) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->f < 1 || d->v < 1 || d->v > d->f) return fail(ctx, MPCX_ERR_DIMS, "need f >= 1 and 1 <= v <= f");
    if (d->weights_mode != MPCX_WEIGHTS_BLOCK && d->weights_mode != MPCX_WEIGHTS_DENSE)
        return fail(ctx, MPCX_ERR_DIMS, "unknown weights_mode %d", d->weights_mode);
    if (d->batch == 0) return MPCX_OK;
    if (!H || !Gamma || !W4 || !gain) return fail(ctx, MPCX_ERR_NULL, "H, Gamma, W4 and gain are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t wsd = 0;
    rc = ensure_lu_work(ctx, d->v * d->m, d->f * d->r, d->batch, &wsd);
    if (rc != MPCX_OK) return rc;
    rc = (d->weights_mode == MPCX_WEIGHTS_DENSE)
             ? launch_qp_gain_dense(H, Gamma, W4, gain, info, d->batch, W, d->m, d->r, d->f, d->v, (double *)ctx->lu_work, wsd,
                                    (hipStream_t)stream)
             : launch_qp_gain(H, Gamma, W4, gain, info, d->batch, W, d->m, d->r, d->f, d->v, (double *)ctx->lu_work, wsd,
                              (hipStream_t)stream);
    if (rc == MPCX_ERR_UNSUPPORTED) return fail(ctx, rc, "gain: v m = %d is beyond what the solver holds on chip", d->v * d->m);
    if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "gain launch");
    return MPCX_OK;
}

int mpcx_qp_gain_rows_f64_dev(mpcx_ctx *ctx, const mpcx_dims *d, const double *H, const double *Gamma, const double *W4,
                              double *gain_rows, int32_t *info, void *stream) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->f < 1 || d->v < 1 || d->v > d->f) return fail(ctx, MPCX_ERR_DIMS, "need f >= 1 and 1 <= v <= f");
    if (d->weights_mode != MPCX_WEIGHTS_BLOCK && d->weights_mode != MPCX_WEIGHTS_DENSE)
        return fail(ctx, MPCX_ERR_DIMS, "unknown weights_mode %d", d->weights_mode);
    if (d->batch == 0) return MPCX_OK;
    if (!H || !Gamma || !W4 || !gain_rows) return fail(ctx, MPCX_ERR_NULL, "H, Gamma, W4 and gain_rows are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t wsd = 0;
    rc = ensure_lu_work(ctx, d->v * d->m, d->m, d->batch, &wsd);
    if (rc != MPCX_OK) return rc;
    rc = ensure_aux(ctx, (size_t)mpcx_array_elems(d->batch, d->interleave, (int64_t)d->v * d->m * d->m) * sizeof(double));
    if (rc != MPCX_OK) return rc;
    rc = launch_qp_gain_rows(H, Gamma, W4, d->weights_mode == MPCX_WEIGHTS_DENSE, gain_rows, info, d->batch, W, d->m, d->r, d->f,
                             d->v, (double *)ctx->aux, (double *)ctx->lu_work, wsd, (hipStream_t)stream);
    if (rc == MPCX_ERR_UNSUPPORTED)
        return fail(ctx, rc, "gain rows: v m = %d, m = %d, f r = %d is beyond what the solver or the product kernel holds on chip",
                    d->v * d->m, d->m, d->f * d->r);
    if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "gain rows launch");
    return MPCX_OK;
}


int mpcx_qp_solve_f64(mpcx_ctx *ctx, const mpcx_dims *d, const double *H, const double *fvec, double *u_seq, int32_t *info) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->batch == 0) return MPCX_OK;
    if (!H || !fvec || !u_seq) return fail(ctx, MPCX_ERR_NULL, "H, fvec and u_seq are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long long vm = (long long)d->v * d->m;
    HostIo io{};
    io.ni = 2;
    io.hin[0] = H; io.nin[0] = mpcx_array_elems(d->batch, d->interleave, vm * vm);
    io.hin[1] = fvec; io.nin[1] = mpcx_array_elems(d->batch, d->interleave, vm);
    io.no = 1;
    io.hout[0] = u_seq; io.nout[0] = io.nin[1];
    io.hinfo = info; io.ninfo = d->batch;
    rc = host_stage(ctx, io);
    if (rc != MPCX_OK) return rc;
    rc = mpcx_qp_solve_f64_dev(ctx, d, io.din[0], io.din[1], io.dout[0], io.dinfo, (void *)ctx->stream);
    return host_collect(ctx, io, rc);
}

int mpcx_qp_gain_f64(mpcx_ctx *ctx, const mpcx_dims *d, const double *H, const double *Gamma, const double *W4, double *gain,
                     int32_t *info) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->batch == 0) return MPCX_OK;
    if (!H || !Gamma || !W4 || !gain) return fail(ctx, MPCX_ERR_NULL, "H, Gamma, W4 and gain are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long long vm = (long long)d->v * d->m, fr = (long long)d->f * d->r;
    HostIo io{};
    io.ni = 3;
    io.hin[0] = H; io.nin[0] = mpcx_array_elems(d->batch, d->interleave, vm * vm);
    io.hin[1] = Gamma; io.nin[1] = mpcx_array_elems(d->batch, d->interleave, fr * vm);
    io.hin[2] = W4; io.nin[2] = (d->weights_mode == MPCX_WEIGHTS_DENSE) ? fr * fr : (long long)d->r * d->r;
    io.no = 1;
    io.hout[0] = gain; io.nout[0] = mpcx_array_elems(d->batch, d->interleave, vm * fr);
    io.hinfo = info; io.ninfo = d->batch;
    rc = host_stage(ctx, io);
    if (rc != MPCX_OK) return rc;
    rc = mpcx_qp_gain_f64_dev(ctx, d, io.din[0], io.din[1], io.din[2], io.dout[0], io.dinfo, (void *)ctx->stream);
    return host_collect(ctx, io, rc);
}

int mpcx_qp_gain_rows_f64(mpcx_ctx *ctx, const mpcx_dims *d, const double *H, const double *Gamma, const double *W4, double *gain_rows,
                     int32_t *info) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->batch == 0) return MPCX_OK;
    if (!H || !Gamma || !W4 || !gain_rows) return fail(ctx, MPCX_ERR_NULL, "H, Gamma, W4 and gain_rows are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long long vm = (long long)d->v * d->m, fr = (long long)d->f * d->r;
    HostIo io{};
    io.ni = 3;
    io.hin[0] = H; io.nin[0] = mpcx_array_elems(d->batch, d->interleave, vm * vm);
    io.hin[1] = Gamma; io.nin[1] = mpcx_array_elems(d->batch, d->interleave, fr * vm);
    io.hin[2] = W4; io.nin[2] = (d->weights_mode == MPCX_WEIGHTS_DENSE) ? fr * fr : (long long)d->r * d->r;
    io.no = 1;
    io.hout[0] = gain_rows; io.nout[0] = mpcx_array_elems(d->batch, d->interleave, (long long)d->m * fr);
    io.hinfo = info; io.ninfo = d->batch;
    rc = host_stage(ctx, io);
    if (rc != MPCX_OK) return rc;
    rc = mpcx_qp_gain_rows_f64_dev(ctx, d, io.din[0], io.din[1], io.din[2], io.dout[0], io.dinfo, (void *)ctx->stream);
    return host_collect(ctx, io, rc);
}

int mpcx_c2d_backward_euler_f64_dev(mpcx_ctx *ctx, const mpcx_dims *d, const double *Ac, const double *Bc, double dt,
                                    double *A, double *B, int32_t *info, void *stream) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->batch == 0) return MPCX_OK;
    if (!Ac || !Bc || !A || !B) return fail(ctx, MPCX_ERR_NULL, "Ac, Bc, A and B are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    rc = launch_c2d_lane(Ac, Bc, dt, A, B, info, d->batch, W, d->n, d->m, s);
    if (rc == MPCX_ERR_UNSUPPORTED) {
        const size_t need = (size_t)mpcx_array_elems(d->batch, d->interleave, (int64_t)d->n * d->n) * sizeof(double);
        if (need > ctx->aux_bytes) {
            if (ctx->aux) { HIP_TRY(ctx, hipFree(ctx->aux)); ctx->aux = nullptr; ctx->aux_bytes = 0; }
            HIP_TRY(ctx, hipMalloc(&ctx->aux, need));
            ctx->aux_bytes = need;
        }
        size_t wsd = 0;
        rc = ensure_lu_work(ctx, d->n, d->n, d->batch, &wsd);
        if (rc != MPCX_OK) return rc;
        rc = launch_c2d_general(Ac, Bc, dt, A, B, info, d->batch, W, d->n, d->m, (double *)ctx->aux, (double *)ctx->lu_work, wsd, s);
    }
    if (rc == MPCX_ERR_UNSUPPORTED) return fail(ctx, rc, "c2d: n = %d is beyond what the solver holds on chip", d->n);
    if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "c2d launch");
    return MPCX_OK;
}

int mpcx_c2d_backward_euler_f64_noalloc_dev(mpcx_ctx *ctx, const mpcx_dims *d, const double *Ac, const double *Bc, double dt,
                                            double *A, double *B, int32_t *info, void *stream) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->batch == 0) return MPCX_OK;
    if (!Ac || !Bc || !A || !B) return fail(ctx, MPCX_ERR_NULL, "Ac, Bc, A and B are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    rc = launch_c2d_lane(Ac, Bc, dt, A, B, info, d->batch, W, d->n, d->m, s);  // tiny plants: registers only, no scratch
    if (rc == MPCX_ERR_UNSUPPORTED) {
        // the static scratch holds I - Ac dt of every instance and, behind it, the blocked solver's workspace for at least
        // one instance when the system is beyond the one-wavefront kernels (fewer instances' worth means more launches)
        const bool on_chip = solve_fits_on_chip(d->n, d->n);
        if (!on_chip && !solve_large_supports(d->n))
            return fail(ctx, MPCX_ERR_UNSUPPORTED, "c2d: n = %d is beyond what the solver holds on chip", d->n);
        const size_t cap = c2d_static_scratch_doubles();
        const size_t need = ((size_t)mpcx_array_elems(d->batch, d->interleave, (int64_t)d->n * d->n) + 31) & ~(size_t)31;
        const size_t per = on_chip ? 0 : solve_large_ws_doubles(d->n, d->n);
        if (need + per > cap)
            return fail(ctx, MPCX_ERR_UNSUPPORTED, "c2d (no-allocation entry point): batch x n^2 = %zu doubles%s exceeds the static "
                        "scratch of %zu doubles; split the batch or use mpcx_c2d_backward_euler_f64_dev", need,
                        on_chip ? "" : " plus one instance of solver workspace", cap);
        double *tmp = nullptr;
        rc = c2d_static_scratch(ctx->device, &tmp);
        if (rc == MPCX_ERR_UNSUPPORTED) return fail(ctx, rc, "c2d (no-allocation entry point): device index %d is beyond the table", ctx->device);
        if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "hipGetSymbolAddress(mpcx_c2d_scratch)");
        // every element of the scratch that the solver reads is written first by this call (the blocked solver packs its
        // matrices and clears its failure flags itself): nothing to clear here
        rc = launch_c2d_general(Ac, Bc, dt, A, B, info, d->batch, W, d->n, d->m, tmp, on_chip ? nullptr : tmp + need,
                                on_chip ? 0 : cap - need, s);
    }
    if (rc == MPCX_ERR_UNSUPPORTED) return fail(ctx, rc, "c2d: n = %d is beyond what the solver holds on chip", d->n);
    if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "c2d launch");
    return MPCX_OK;
}

int mpcx_plants_f64_dev(mpcx_ctx *ctx, int kind, int64_t batch, int64_t interleave, const double *params, double dt,
                        double *A, double *B, double *C, int32_t *info, void *stream) {
    if (!ctx) return MPCX_ERR_NULL;
    if (kind != MPCX_PLANT_DC_MOTOR && kind != MPCX_PLANT_TWO_MASS) return fail(ctx, MPCX_ERR_DIMS, "unknown plant kind %d", kind);
    mpcx_dims d{};
    d.n = kind == MPCX_PLANT_DC_MOTOR ? 2 : 4;
    d.m = d.r = d.f = d.v = 1;
    d.batch = batch;
    d.interleave = interleave;
    long long W = 1;
    int rc = check_common(ctx, &d, &W);
    if (rc != MPCX_OK) return rc;
    if (batch == 0) return MPCX_OK;
    if (!params || !A || !B) return fail(ctx, MPCX_ERR_NULL, "params, A and B are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rc = launch_plants(kind, params, dt, A, B, C, info, batch, W, (hipStream_t)stream);
    if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "plants launch");
    return MPCX_OK;
}


int mpcx_c2d_backward_euler_f64(mpcx_ctx *ctx, const mpcx_dims *d, const double *Ac, const double *Bc, double dt, double *A,
                                double *B, int32_t *info) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->batch == 0) return MPCX_OK;
    if (!Ac || !Bc || !A || !B) return fail(ctx, MPCX_ERR_NULL, "Ac, Bc, A and B are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long long ea = mpcx_array_elems(d->batch, d->interleave, (int64_t)d->n * d->n);
    const long long eb = mpcx_array_elems(d->batch, d->interleave, (int64_t)d->n * d->m);
    HostIo io{};
    io.ni = 2; io.hin[0] = Ac; io.nin[0] = ea; io.hin[1] = Bc; io.nin[1] = eb;
    io.no = 2; io.hout[0] = A; io.nout[0] = ea; io.hout[1] = B; io.nout[1] = eb;
    io.hinfo = info; io.ninfo = d->batch;
    rc = host_stage(ctx, io);
    if (rc != MPCX_OK) return rc;
    rc = mpcx_c2d_backward_euler_f64_dev(ctx, d, io.din[0], io.din[1], dt, io.dout[0], io.dout[1], io.dinfo, (void *)ctx->stream);
    return host_collect(ctx, io, rc);
}

int mpcx_plants_f64(mpcx_ctx *ctx, int kind, int64_t batch, int64_t interleave, const double *params, double dt, double *A,
                    double *B, double *C, int32_t *info) {
    if (!ctx) return MPCX_ERR_NULL;
    if (kind != MPCX_PLANT_DC_MOTOR && kind != MPCX_PLANT_TWO_MASS) return fail(ctx, MPCX_ERR_DIMS, "unknown plant kind %d", kind);
    if (batch < 0) return fail(ctx, MPCX_ERR_DIMS, "batch must be >= 0");
    if (batch == 0) return MPCX_OK;
    if (!params || !A || !B) return fail(ctx, MPCX_ERR_NULL, "params, A and B are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int n = kind == MPCX_PLANT_DC_MOTOR ? 2 : 4, np = kind == MPCX_PLANT_DC_MOTOR ? 5 : 6;
    HostIo io{};
    io.ni = 1; io.hin[0] = params; io.nin[0] = mpcx_array_elems(batch, interleave, np);
    io.no = 3;
    io.hout[0] = A; io.nout[0] = mpcx_array_elems(batch, interleave, n * n);
    io.hout[1] = B; io.nout[1] = mpcx_array_elems(batch, interleave, n);
    io.hout[2] = C; io.nout[2] = mpcx_array_elems(batch, interleave, n);
    io.hinfo = info; io.ninfo = batch;
    int rc = host_stage(ctx, io);
    if (rc != MPCX_OK) return rc;
    rc = mpcx_plants_f64_dev(ctx, kind, batch, interleave, io.din[0], dt, io.dout[0], io.dout[1], io.dout[2], io.dinfo, (void *)ctx->stream);
    return host_collect(ctx, io, rc);
}

int mpcx_closed_loop_f64_dev(mpcx_ctx *ctx, const mpcx_dims *d, const double *A, const double *B, const double *C,
                             const double *x0, const double *Phi, const double *gain, const double *ref, double *X,
                             double *U, double *Y, void *stream) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->T < 0) return fail(ctx, MPCX_ERR_DIMS, "T must be >= 0");
    if (d->f < 1 || d->v < 1 || d->v > d->f) return fail(ctx, MPCX_ERR_DIMS, "need f >= 1 and 1 <= v <= f");
    if (d->batch == 0) return MPCX_OK;
    if (!A || !B || !C || !x0 || !Phi || !gain || !ref) return fail(ctx, MPCX_ERR_NULL, "A, B, C, x0, Phi, gain and ref are required");
    if (!X && !U && !Y) return MPCX_OK;
    LoopArgs a;
    a.A = A; a.B = B; a.C = C; a.x0 = x0; a.Phi = Phi; a.gain = gain; a.ref = ref; a.X = X; a.U = U; a.Y = Y;
    a.batch = d->batch; a.W = W; a.T = d->T;
    a.n = d->n; a.m = d->m; a.r = d->r; a.f = d->f; a.v = d->v; a.ref_shared = d->u_shared;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rc = (W == 1) ? MPCX_ERR_UNSUPPORTED : launch_closed_loop_lane(a, s);
    if (rc == MPCX_ERR_UNSUPPORTED) rc = launch_closed_loop_group(a, s);
    if (rc == MPCX_ERR_UNSUPPORTED && W == 1) rc = launch_closed_loop_lane(a, s);
    if (rc == MPCX_ERR_UNSUPPORTED)
        return fail(ctx, rc, "closed loop: n=%d m=%d r=%d f=%d does not fit on chip", d->n, d->m, d->r, d->f);
    if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "closed loop launch");
    return MPCX_OK;
}

int mpcx_closed_loop_f64(mpcx_ctx *ctx, const mpcx_dims *d, const double *A, const double *B, const double *C,
                         const double *x0, const double *Phi, const double *gain, const double *ref, double *X, double *U,
                         double *Y) {
    long long W = 1;
    int rc = check_common(ctx, d, &W);
    if (rc != MPCX_OK) return rc;
    if (d->T < 0) return fail(ctx, MPCX_ERR_DIMS, "T must be >= 0");
    if (d->batch == 0) return MPCX_OK;
    if (!A || !B || !C || !x0 || !Phi || !gain || !ref) return fail(ctx, MPCX_ERR_NULL, "A, B, C, x0, Phi, gain and ref are required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long long n = d->n, m = d->m, r = d->r, f = d->f, v = d->v, bt = d->batch, il = d->interleave, T = d->T;
    constexpr int NI = 7, NO = 3;
    const double *hin[NI] = {A, B, C, x0, Phi, gain, ref};
    const long long nin[NI] = {mpcx_array_elems(bt, il, n * n), mpcx_array_elems(bt, il, n * m), mpcx_array_elems(bt, il, r * n),
                               mpcx_array_elems(bt, il, n), mpcx_array_elems(bt, il, f * r * n),
                               mpcx_array_elems(bt, il, v * m * f * r),
                               d->u_shared ? r * (T + f) : mpcx_array_elems(bt, il, r * (T + f))};
    double *hout[NO] = {X, U, Y};
    const long long nout[NO] = {mpcx_array_elems(bt, il, n * (T + 1)), mpcx_array_elems(bt, il, m * T), mpcx_array_elems(bt, il, r * T)};
    size_t off_in[NI], off_out[NO], total = 0;
    for (int i = 0; i < NI; ++i) { off_in[i] = total; total += align256((size_t)nin[i] * sizeof(double)); }
    for (int i = 0; i < NO; ++i) { off_out[i] = total; if (hout[i]) total += align256((size_t)nout[i] * sizeof(double)); }
    rc = ensure_work(ctx, total ? total : 256);
    if (rc != MPCX_OK) return rc;
    char *base = (char *)ctx->work;
    const double *din[NI];
    double *dout[NO];
    for (int i = 0; i < NI; ++i) {
        din[i] = (const double *)(base + off_in[i]);
        if (nin[i] > 0)
            HIP_TRY(ctx, hipMemcpyAsync((void *)din[i], hin[i], (size_t)nin[i] * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    for (int i = 0; i < NO; ++i) dout[i] = hout[i] ? (double *)(base + off_out[i]) : nullptr;
    rc = mpcx_closed_loop_f64_dev(ctx, d, din[0], din[1], din[2], din[3], din[4], din[5], din[6], dout[0], dout[1], dout[2],
                                  (void *)ctx->stream);
    if (rc != MPCX_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    for (int i = 0; i < NO; ++i)
        if (hout[i] && nout[i] > 0)
            HIP_TRY(ctx, hipMemcpyAsync(hout[i], dout[i], (size_t)nout[i] * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MPCX_OK;
}

int mpcx_stats_f64_dev(mpcx_ctx *ctx, const double *buf, int64_t count, mpcx_stats *out_dev, void *stream) {
    if (!ctx) return MPCX_ERR_NULL;
    if (!out_dev || (count > 0 && !buf)) return fail(ctx, MPCX_ERR_NULL, "buf / out_dev is NULL");
    if (count < 0) return fail(ctx, MPCX_ERR_DIMS, "count must be >= 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = launch_stats(buf, count, out_dev, ctx->stat_scratch, stats_scratch_bytes(), (hipStream_t)stream);
    if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "stats launch");
    return MPCX_OK;
}

int mpcx_stats_f64(mpcx_ctx *ctx, const double *buf_dev, int64_t count, mpcx_stats *out_host) {
    if (!ctx) return MPCX_ERR_NULL;
    if (!out_host) return fail(ctx, MPCX_ERR_NULL, "out_host is NULL");
    mpcx_stats *rec = (mpcx_stats *)((char *)ctx->stat_scratch + stats_scratch_bytes());
    int rc = mpcx_stats_f64_dev(ctx, buf_dev, count, rec, (void *)ctx->stream);
    if (rc != MPCX_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out_host, rec, sizeof(mpcx_stats), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MPCX_OK;
}

int mpcx_relayout_f64_dev(mpcx_ctx *ctx, int64_t batch, int64_t elems_per_instance, int64_t src_interleave,
                          const double *src, int64_t dst_interleave, double *dst, void *stream) {
    if (!ctx) return MPCX_ERR_NULL;
    if (batch < 0 || elems_per_instance < 0) return fail(ctx, MPCX_ERR_DIMS, "batch and elems_per_instance must be >= 0");
    if (batch == 0 || elems_per_instance == 0) return MPCX_OK;
    if (!src || !dst) return fail(ctx, MPCX_ERR_NULL, "src / dst is NULL");
    const long long Ws = src_interleave == 0 ? batch : src_interleave, Wd = dst_interleave == 0 ? batch : dst_interleave;
    if (Ws < 1 || Wd < 1) return fail(ctx, MPCX_ERR_LAYOUT, "interleave must be >= 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = launch_relayout(batch, elems_per_instance, Ws, src, Wd, dst, (hipStream_t)stream);
    if (rc == MPCX_ERR_UNSUPPORTED) return fail(ctx, rc, "relayout: array too large for one launch");
    if (rc != 0) return hip_fail(ctx, (hipError_t)rc, "relayout launch");
    return MPCX_OK;
}

}  // extern "C"
