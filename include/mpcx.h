/*
 * mpcx.h -- C ABI of the MI355X (gfx950) batched MPC hot path.
 *
 * The reference (rdesarz/mpc-rs, crate `control_sys_rs`) has no FFI of its own: its boundary is the public Rust
 * API in src/control/.  The entry points below are what a `-sys` crate for that API binds; each one names the
 * reference function it replaces.  Everything is f64.  The library computes on the GPU only: there is no CPU
 * path behind these calls, and a call fails (it never falls back) when no HIP device can be used.
 *
 * Conventions
 *   - Return value: 0 = MPCX_OK; negative = argument error (MPCX_ERR_*); positive = hipError_t from the runtime.
 *     Nothing unwinds or aborts across this boundary.  mpcx_last_error() gives the text of the last failure
 *     of a context.
 *   - A context is bound to one device and owns one stream plus staging buffers.  It is not thread-safe: use
 *     one context per host thread (the reference's Controller holds an Rc and is !Send, controller.rs:3,8).
 *     One process per GPU is the intended deployment; batches are sharded by instance index across processes.
 *   - Host-pointer calls (no suffix) copy inputs to the device, run, copy the outputs back and return when the
 *     outputs are in place.  `_dev` calls take device pointers and a hipStream_t (as void*), enqueue the work
 *     and return without synchronising -- except that a call which has to grow the context's device scratch
 *     allocates, which synchronises the device (see mpcx_reserve).  Callers own every buffer; the library keeps
 *     no pointer after return.
 *   - Matrices of one instance are column-major, as nalgebra's DMatrix::as_slice() yields them: element (i, j)
 *     of a rows x cols matrix has element index e = j*rows + i.
 *   - Batch layout (`interleave`, W): element e of instance b of an array with E elements per instance lives at
 *         (b / W) * (E * W)  +  e * W  +  (b % W).
 *     W = 1 is instance-major (each instance's matrix contiguous: what a caller holding one DMatrix per plant
 *     has).  W = 0 means W = batch (fully batch-interleaved, instance index fastest).  Any other W (a multiple
 *     of 64) tiles the batch into groups of W instances; the last group is padded, so buffers hold
 *     ceil(batch / W) * W * E elements.  One W applies to every per-instance array of a call.
 *     Weights (W3, W4) are shared by the whole batch and are plain column-major matrices.
 */
#ifndef MPCX_H
#define MPCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCX_VERSION 100 /* 0.1.0 */

#define MPCX_OK 0
#define MPCX_ERR_NULL (-1)        /* a required pointer is NULL */
#define MPCX_ERR_DIMS (-2)        /* a dimension is out of range (n, m, r, f >= 1; 1 <= v <= f; batch, T >= 0) */
#define MPCX_ERR_LAYOUT (-3)      /* interleave is not 0, 1 or a multiple of 64 */
#define MPCX_ERR_UNSUPPORTED (-4) /* combination of dimensions / modes exceeds what the kernels hold on chip */
#define MPCX_ERR_NO_DEVICE (-5)   /* no usable HIP device */
#define MPCX_ERR_SINGULAR (-6)    /* reported per instance through `info`, see mpcx_qp_gain_* */
#define MPCX_ERR_COMM (-7)        /* RCCL is missing or a communicator call failed, see mpcx_comm_last_error */

#define MPCX_INTERLEAVE_FULL 0 /* W = batch */
#define MPCX_INTERLEAVE_AOS 1  /* W = 1, instance-major */

#define MPCX_WEIGHTS_BLOCK 0 /* W4 = one r x r block, repeated on the diagonal of the (f r) x (f r) weight */
#define MPCX_WEIGHTS_DENSE 1 /* W4 = dense (f r) x (f r), the argument of form_lifted_matrices (controller.rs:23) */

#define MPCX_HESSIAN_AUTO 0       /* library picks */
#define MPCX_HESSIAN_CONTRACT 1   /* H = Gamma^T (W4 Gamma) + W3 as a dense contraction (FP64 MFMA) */
#define MPCX_HESSIAN_STRUCTURED 2 /* block-Toeplitz recurrence; needs MPCX_WEIGHTS_BLOCK */

typedef struct mpcx_ctx mpcx_ctx;

typedef struct mpcx_dims {
    int32_t n; /* states:  A is n x n            (controller.rs:28) */
    int32_t m; /* inputs:  B is n x m            (controller.rs:30) */
    int32_t r; /* outputs: C is r x n            (controller.rs:29) */
    int32_t f; /* prediction horizon             (controller.rs:20) */
    int32_t v; /* control horizon, 1 <= v <= f   (controller.rs:21) */
    int32_t weights_mode; /* MPCX_WEIGHTS_* */
    int32_t hessian_algo; /* MPCX_HESSIAN_* */
    int32_t u_shared;     /* rollout: 0 = U holds m x T per instance, 1 = one m x T shared by all instances;
                             closed loop: the same for the reference trajectory */
    int64_t batch;        /* number of independent instances */
    int64_t T;            /* rollout: number of input columns (simulator.rs:10) */
    int64_t interleave;   /* W, see above */
} mpcx_dims;

/* Fixed-size record of what a shard produced; ranks exchange these (not the results) after a sharded run. */
typedef struct mpcx_stats {
    int64_t count;      /* elements examined */
    int64_t non_finite; /* NaN or Inf among them */
    double sum;         /* sum of the elements */
    double sum_sq;      /* sum of squares (squared Frobenius norm) */
    double max_abs;     /* largest magnitude */
    uint64_t checksum;  /* order-independent 64-bit checksum of the bit patterns (sum of mixed words) */
} mpcx_stats;

int mpcx_version(void);

/* Number of f64 elements a per-instance array of E elements occupies for this batch and interleave. */
int64_t mpcx_array_elems(int64_t batch, int64_t interleave, int64_t elems_per_instance);

/* Creates a context on HIP device `device_id` (its own stream, staging buffers grown on demand). */
int mpcx_create(mpcx_ctx **out, int device_id);
void mpcx_destroy(mpcx_ctx *ctx);
const char *mpcx_last_error(const mpcx_ctx *ctx);

/*
 * Device scratch.  Besides the caller's arrays some calls work in device scratch that the context owns and grows on
 * demand: the blocked solver's workspace (solve / gain / c2d beyond what fits on chip), I - Ac dt of the general
 * discretisation, the instance-major copy of an interleaved batch on the workgroup kernels, and the host-pointer
 * calls' staging.  Growing a buffer is hipFree + hipMalloc, which synchronise the device: the FIRST `_dev` call of a
 * given size therefore may block, later ones only enqueue.  mpcx_reserve grows every buffer that the `_dev` calls of
 * `dims` (build, solve, gain, c2d; its batch, interleave, n, m, r, f, v, weights_mode, hessian_algo) would need, so that
 * after it those calls never allocate.  All calls of a context share its scratch: run one call per context at a time
 * (build, solve, gain and c2d alike), or use one context per stream.
 */
int mpcx_reserve(mpcx_ctx *ctx, const mpcx_dims *dims);

/* Context options.  MPCX_OPT_RELAY_CAP_BYTES: upper bound of the instance-major copy through which an interleaved batch
 * reaches the workgroup kernels (default 2 GiB; the batch goes through in chunks of that size). */
#define MPCX_OPT_RELAY_CAP_BYTES 1
/* MPCX_OPT_QP_STREAMING: 1 sends every instance-major build with block weights through the streaming form (Markov vectors
 * in a device workspace, see mpcx_qp_build) whatever its size; 0 (default) only the shapes beyond the on-chip kernels.
 * For testing and for comparing the two forms; results agree to rounding. */
#define MPCX_OPT_QP_STREAMING 2
int mpcx_set_option(mpcx_ctx *ctx, int option, int64_t value);

/*
 * Condensed QP construction, batched.  Replaces Controller::form_lifted_matrices up to the Hessian
 * (controller.rs:16-93) and the `O x - ref` / `M^T W4` pieces of compute_control_input and of the gain
 * (controller.rs:125-127, :96), for `batch` independent plants:
 *     Phi   (f r) x n        block i   = C A^(i+1)                                   [mat_o, controller.rs:32-44]
 *     Gamma (f r) x (v m)    block i,k = C A^(i-k) B (k <= i); for rows i >= v the last block column holds
 *                            C (sum_{s=0}^{i-v+1} A^s) B                              [mat_m, controller.rs:46-90]
 *     H     (v m) x (v m)    = Gamma^T W4 Gamma + W3                                 [controller.rs:92-93]
 *     fvec  (v m)            = Gamma^T W4 (Phi x0 - ref)
 * Inputs per instance: A, B, C, x0 (n), ref (f r).  Shared: W3 (v m x v m dense), W4 (r x r block or dense
 * (f r) x (f r), per dims->weights_mode).  Any output pointer may be NULL: that output is neither computed
 * for its own sake nor stored.  x0 / ref may be NULL when fvec is NULL; W3 / W4 when H and fvec are NULL.
 * Layouts: plants of up to 6 states and 3 inputs / outputs are native in the interleaved layouts (one lane per
 * instance); larger ones, MPCX_WEIGHTS_DENSE and MPCX_HESSIAN_CONTRACT are native in interleave = 1, and an interleaved
 * batch then goes through an instance-major copy in device scratch that the context owns (at most 2 GiB at a time; every
 * output is moved three times).
 * Sizes: with MPCX_WEIGHTS_BLOCK and MPCX_HESSIAN_AUTO / STRUCTURED there is no limit on f, v or the product sizes beyond
 * memory.  Three kernel families serve them: one lane per instance while (f r m + max(f r, v m m)) * 512 bytes fit 160 KB
 * of LDS; one workgroup per instance while the instance's working set fits LDS -- about
 * 2 m r (2 f - v + 1) + max(n n + n m + r n + 2 n + r r + 2 f r + power tables, 2 (2 v + 1) m m) + (f > v ? 2 (v + 1) m m : 0)
 * <= 20 480 doubles (exactly: n=12 m=4 r=12 up to f = v = 127; n=8 m=2 r=2 up to f = v = 852); and beyond that a streaming form that keeps the
 * Markov vectors in a device workspace the context owns (4 m f r + 2 f r doubles per instance in flight, at most 8 GiB at a
 * time, see mpcx_reserve), limited only by 2 n (m + r) + n + r m <= 20 480 doubles of recurrence state.  MPCX_WEIGHTS_DENSE
 * and MPCX_HESSIAN_CONTRACT keep on-chip limits ((f r)(m + 1) resp. the Markov vectors in LDS) and return
 * MPCX_ERR_UNSUPPORTED beyond them.  H is never formed as the reference's dense triple product: its blocks are summed
 * along block diagonals, and with held inputs (v < f) the one-lane kernels evaluate the blocks of the last two block rows
 * and columns through G_(k+d) = (C A^k) A^d B with the sums over k formed once; every form agrees with the dense product
 * to rounding (the tests hold 1e-10, relative, Frobenius, per instance).  Exact zeros of W3 are skipped: the
 * band structure of W3 is measured on the device by each call (W3 as Controller::new builds it is block tridiagonal,
 * controller.rs:149-177).  The context's scratch is shared by all its calls: do not run two calls of one context
 * concurrently on different streams.
 */
int mpcx_qp_build_f64(mpcx_ctx *ctx, const mpcx_dims *dims, const double *A, const double *B, const double *C,
                      const double *W3, const double *W4, const double *x0, const double *ref, double *Phi,
                      double *Gamma, double *H, double *fvec);
int mpcx_qp_build_f64_dev(mpcx_ctx *ctx, const mpcx_dims *dims, const double *A, const double *B, const double *C,
                          const double *W3, const double *W4, const double *x0, const double *ref, double *Phi,
                          double *Gamma, double *H, double *fvec, void *stream);

/*
 * Open-loop rollout, batched.  Replaces simulator::system_simulate (simulator.rs:5-35; `step` and
 * `compute_system_response`, simulator.rs:37-63, are thin wrappers over it) and, with T = 1,
 * Controller::propagate_dynamics (controller.rs:101-115):
 *     X n x (T+1): X[:,0] = x0, X[:,i+1] = A X[:,i] + B U[:,i];   Y r x T: Y[:,i] = C X[:,i].
 * U is m x T (time along columns).  X or Y may be NULL.
 */
int mpcx_rollout_f64(mpcx_ctx *ctx, const mpcx_dims *dims, const double *A, const double *B, const double *C,
                     const double *x0, const double *U, double *X, double *Y);
int mpcx_rollout_f64_dev(mpcx_ctx *ctx, const mpcx_dims *dims, const double *A, const double *B, const double *C,
                         const double *x0, const double *U, double *X, double *Y, void *stream);

/*
 * Batched solve with the condensed Hessian, following nalgebra's try_inverse (controller.rs:93-95, model.rs:56) in both
 * of its regimes.  Systems of 1..4 unknowns (v m <= 4; n <= 4 in the discretisation): the same closed-form (cofactor)
 * inverses, evaluated with multiply and add kept separate, failing exactly where try_inverse returns None, on
 * `determinant == 0.0`; the one deviation is that a non-finite determinant is flagged as well.  Above 4 unknowns:
 * Gaussian elimination with partial pivoting per instance, as nalgebra's LU; an exactly zero or non-finite pivot fails.
 * Up to 64 unknowns (with right-hand sides that fit 64 KB of LDS beside the matrix: every solve, first-move rows and
 * discretisation of that size, gains of up to about 60 outputs x horizon) the elimination goes column by column like the
 * reference's, so a matrix that is singular only
 * through exact cancellation (two equal rows, say) meets its exactly zero pivot here as it does there.  Larger systems
 * are eliminated in panels of 16 columns on the matrix cores with the same pivoting rule: zero rows or columns and
 * non-finite entries are flagged at every size, but a cancellation that relies on two rows seeing identical rounding
 * may leave a tiny non-zero pivot instead (info 0, a meaningless finite result, as for any numerically singular H).
 * Results agree with the reference's to the 1e-10 bar scaled by the condition number.
 *   mpcx_qp_solve:  u_seq (v m per instance) = -H^-1 fvec: the optimal input sequence; its first m entries are the
 *                   move the reference applies (controller.rs:130-134 with gain s = -H^-1 fvec).
 *   mpcx_qp_gain:   gain ((v m) x (f r) per instance) = H^-1 Gamma^T W4, the reference's gain_matrix
 *                   (controller.rs:93-96); W4 is the r x r block or the dense (f r) x (f r) matrix, per dims->weights_mode.
 * `info` (int32 per instance, plain array of `batch` entries, may be NULL) is 0 on success and 1 where the reference
 * returns Err("Matrix inversion failed") (controller.rs:93-95): an exactly zero or non-finite pivot; the outputs of
 * such an instance are NaN.  The call itself still returns MPCX_OK.  H, fvec, Gamma, u_seq, gain use dims->interleave.
 * Limits: v m <= 1232 unknowns (a block of 16 columns of all rows has to fit the 160 KB of LDS); systems beyond 64
 * unknowns (or with more right-hand sides than a wavefront's LDS share holds beside them) go through a device workspace
 * the context owns ((v m)^2 doubles per instance in flight; it grows on demand, see mpcx_reserve); there the r x r weight
 * block of mpcx_qp_gain is served up to r = 64 on the matrix cores (wider blocks, for up to about 140 unknowns, by a plain
 * and much slower kernel).  Beyond these: MPCX_ERR_UNSUPPORTED.
 */
int mpcx_qp_solve_f64(mpcx_ctx *ctx, const mpcx_dims *dims, const double *H, const double *fvec, double *u_seq,
                      int32_t *info);
int mpcx_qp_solve_f64_dev(mpcx_ctx *ctx, const mpcx_dims *dims, const double *H, const double *fvec, double *u_seq,
                          int32_t *info, void *stream);
int mpcx_qp_gain_f64(mpcx_ctx *ctx, const mpcx_dims *dims, const double *H, const double *Gamma, const double *W4,
                     double *gain, int32_t *info);
int mpcx_qp_gain_f64_dev(mpcx_ctx *ctx, const mpcx_dims *dims, const double *H, const double *Gamma, const double *W4,
                         double *gain, int32_t *info, void *stream);

/*
 * First-move rows of the gain, batched: the first m rows of gain_matrix = H^-1 Gamma^T W4 (controller.rs:93-96), which is
 * all that compute_control_input applies (controller.rs:130-132 takes u_seq[0] of gain s; for m inputs its first m
 * entries).  Computed as m solves with H^T (right-hand sides e_0 .. e_(m-1)) and one product with Gamma^T W4 instead of
 * f r solves, so the cost is that of the factorisation.  gain_rows is m x (f r) per instance, column-major, in
 * dims->interleave: exactly what mpcx_closed_loop_* takes as `gain` when its dims->v is set to 1 (a (1 m) x (f r) gain).
 * Same weights_mode, pivoting, `info` semantics and limits as mpcx_qp_gain; failed instances hold NaN.  Additive: the
 * reference always forms the whole gain.
 */
int mpcx_qp_gain_rows_f64(mpcx_ctx *ctx, const mpcx_dims *dims, const double *H, const double *Gamma, const double *W4,
                          double *gain_rows, int32_t *info);
int mpcx_qp_gain_rows_f64_dev(mpcx_ctx *ctx, const mpcx_dims *dims, const double *H, const double *Gamma, const double *W4,
                              double *gain_rows, int32_t *info, void *stream);

/*
 * Backward-Euler discretisation, batched: A = (I - Ac dt)^-1, B = A (Bc dt), the last three lines of both
 * Model::new (model.rs:55-57, :138-140).  Ac is n x n, Bc n x m per instance (dims->n, m, batch, interleave).
 * `info` as for mpcx_qp_solve: 1 marks an instance whose matrix could not be inverted (the reference unwraps the
 * failed try_inverse and panics, model.rs:56); its A and B are NaN.
 */
int mpcx_c2d_backward_euler_f64(mpcx_ctx *ctx, const mpcx_dims *dims, const double *Ac, const double *Bc, double dt,
                                double *A, double *B, int32_t *info);
int mpcx_c2d_backward_euler_f64_dev(mpcx_ctx *ctx, const mpcx_dims *dims, const double *Ac, const double *Bc, double dt,
                                    double *A, double *B, int32_t *info, void *stream);

/*
 * mpcx_c2d_backward_euler_f64_dev for callers that cannot allocate device memory while they run: code under HIP
 * graph capture, tight serving loops that re-discretise every tick, harnesses that forbid hipMalloc.  It has no
 * counterpart in the reference (the computation is model.rs:55-57, :138-140 as above).  Same arguments, same kernels and
 * bit-identical results; it never calls hipMalloc / hipFree or any allocating routine.  Its scratch -- I - Ac dt of
 * every instance and, for systems beyond the one-wavefront solver kernels (n > 60 or so), the blocked solver's workspace
 * behind it -- lives in a static device array of MPCX_C2D_NOALLOC_MAX_DOUBLES doubles (16 MiB on each GPU the
 * library runs on, held until the process ends), whose address on the context's device mpcx_create resolves.
 * Limits, reported as MPCX_ERR_UNSUPPORTED with a message: mpcx_array_elems(batch, interleave, n n), plus about
 * (n + 15)^2 + 35 n doubles of solver workspace for one instance when n > 60, must not exceed
 * MPCX_C2D_NOALLOC_MAX_DOUBLES (plants the one-lane kernels take -- n <= 6 with m <= 2 -- need no scratch and have no
 * such limit).  What is left behind the matrices decides how many instances one solver launch takes: a batch that
 * nearly fills the array is solved in many small launches.  All calls on one GPU share that GPU's
 * array: two such calls that overlap in time on one device (different streams or contexts) race.
 */
#define MPCX_C2D_NOALLOC_MAX_DOUBLES ((int64_t)1 << 21)
int mpcx_c2d_backward_euler_f64_noalloc_dev(mpcx_ctx *ctx, const mpcx_dims *dims, const double *Ac, const double *Bc,
                                            double dt, double *A, double *B, int32_t *info, void *stream);

/*
 * The reference's two plants from per-instance physical parameters, discretised: dc_motor::Model::new
 * (model.rs:45-66; parameters b, j, k, l, r in that order; n = 2) and mpc::Model::new (model.rs:126-149; m1, m2, k1,
 * k2, d1, d2; n = 4).  m = r = 1.  `params` holds 5 resp. 6 values per instance in the batch layout; A (n x n),
 * B (n x 1) and C (1 x n, may be NULL) come out in the same layout.  The reference keeps the parameter fields private
 * (model.rs:16-22, :95-102) and only offers the defaults; this is the constructor a parameter sweep needs.
 */
#define MPCX_PLANT_DC_MOTOR 0
#define MPCX_PLANT_TWO_MASS 1
int mpcx_plants_f64(mpcx_ctx *ctx, int kind, int64_t batch, int64_t interleave, const double *params, double dt, double *A,
                    double *B, double *C, int32_t *info);
int mpcx_plants_f64_dev(mpcx_ctx *ctx, int kind, int64_t batch, int64_t interleave, const double *params, double dt,
                        double *A, double *B, double *C, int32_t *info, void *stream);

/*
 * Receding-horizon closed loop, batched: the loop of src/main.rs:51-71 for `batch` plants at once.  For t = 0 .. T-1:
 *     s = ref[:, t .. t+f) - Phi x_t; u_t = first m entries of gain s   [compute_control_input, controller.rs:117-135]
 *     y_t = C x_t; x_{t+1} = A x_t + B u_t                              [propagate_dynamics, controller.rs:101-115]
 * Phi ((f r) x n) and gain ((v m) x (f r)) are per instance, as mpcx_qp_build / mpcx_qp_gain leave them.  ref is
 * r x (T + f), time along columns: for r = 1 the reference's desired_ctrl_traj_total with time_steps = T + f
 * (main.rs:18-24, :51); per instance, or one trajectory for the whole batch when dims->u_shared = 1.
 * Outputs (any may be NULL): X n x (T+1) with X[:,0] = x0; U m x T; Y r x T.  The reference's `outputs` matrix
 * (main.rs:40-41, :63) is [C x0, Y[:,0], Y[:,1], ...].  The reference handles m = r = 1 only; larger m, r follow the
 * same formulas with the window stacked as ref[a, t+i] at row i r + a.
 * Instance-major batches of plants with m <= 4, n <= 12, r <= 16 evaluate u_t as K ref_window - (K Phi) x_t (K the first
 * m gain rows; the products with the reference windows are formed ahead of the serial steps): the same trajectories up
 * to the rounding of the regrouped product, which the tests hold to 1e-10 (relative, Frobenius) of the reference's order.
 */
int mpcx_closed_loop_f64(mpcx_ctx *ctx, const mpcx_dims *dims, const double *A, const double *B, const double *C,
                         const double *x0, const double *Phi, const double *gain, const double *ref, double *X, double *U,
                         double *Y);
int mpcx_closed_loop_f64_dev(mpcx_ctx *ctx, const mpcx_dims *dims, const double *A, const double *B, const double *C,
                             const double *x0, const double *Phi, const double *gain, const double *ref, double *X,
                             double *U, double *Y, void *stream);

/*
 * Reduces `count` f64 values at device pointer `buf` to one mpcx_stats record written to device memory
 * `out_dev` (no counterpart in the reference: it is what sharded runs gather instead of the results).
 */
int mpcx_stats_f64_dev(mpcx_ctx *ctx, const double *buf, int64_t count, mpcx_stats *out_dev, void *stream);
/* Same for a host-visible result: the record is copied to `out_host` and the call synchronises. */
int mpcx_stats_f64(mpcx_ctx *ctx, const double *buf_dev, int64_t count, mpcx_stats *out_host);

/*
 * Multi-GPU exchange (no counterpart in the reference, which is single-threaded and device-less).  A batch is sharded
 * by instance index, one rank per GPU -- a process per GPU, or a host thread per GPU with its own context -- and
 * nothing is exchanged while the shards compute.  Afterwards the ranks exchange one mpcx_stats record each and, on
 * request, gather a result array to one rank.  Both run over RCCL (xGMI inside a node) on a communicator that the
 * library owns; RCCL (librccl.so.1) is loaded when the first of these functions is called, never before.
 *
 *   mpcx_comm_get_id    one rank (any) obtains MPCX_COMM_ID_BYTES opaque bytes and hands them to all other ranks
 *                       through whatever channel the host has (a file, a socket, MPI, a parent process).
 *   mpcx_comm_create    every rank, with its own context (the communicator uses that context's device), the same id,
 *                       the number of ranks and its rank.  Collective: returns when all `world` ranks have called it.
 *   mpcx_gather_stats   reduces `count` doubles at `buf_dev` (ordered after the work already queued on `stream`) to this
 *                       rank's record and all-gathers the records: out_host receives `world` records in rank order, on
 *                       every rank.  Collective; returns when the records are in out_host.
 *   mpcx_gather_f64_dev gathers `count` doubles from every rank to dst_dev on `root` (world * count doubles, rank
 *                       order; ignored on the other ranks).  Each rank sends straight to the root (point-to-point, one
 *                       xGMI link per source), ordered after the work queued on `stream`; work queued on `stream`
 *                       afterwards waits for the transfer.  Collective, asynchronous: mpcx_comm_synchronize waits.
 * Errors: MPCX_ERR_COMM (text in mpcx_comm_last_error; with a NULL communicator it reports why RCCL could not be loaded).
 * A communicator is used by one host thread at a time, like a context.
 */
#define MPCX_COMM_ID_BYTES 128
typedef struct mpcx_comm mpcx_comm;
int mpcx_comm_get_id(void *id_out);
int mpcx_comm_create(mpcx_comm **out, mpcx_ctx *ctx, const void *id, int world, int rank);
void mpcx_comm_destroy(mpcx_comm *comm);
const char *mpcx_comm_last_error(const mpcx_comm *comm);
int mpcx_comm_rank(const mpcx_comm *comm);
int mpcx_comm_world(const mpcx_comm *comm);
int mpcx_gather_stats(mpcx_comm *comm, const double *buf_dev, int64_t count, void *stream, mpcx_stats *out_host);
int mpcx_gather_f64_dev(mpcx_comm *comm, const double *src_dev, int64_t count, double *dst_dev, int root, void *stream);
int mpcx_comm_synchronize(mpcx_comm *comm);

/*
 * Re-arranges a batched array between two interleave widths on the device (src and dst must not overlap).
 * No counterpart in the reference; callers holding one DMatrix per plant (W = 1) use it to reach the
 * batch-interleaved layout and back.
 */
int mpcx_relayout_f64_dev(mpcx_ctx *ctx, int64_t batch, int64_t elems_per_instance, int64_t src_interleave,
                          const double *src, int64_t dst_interleave, double *dst, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MPCX_H */
