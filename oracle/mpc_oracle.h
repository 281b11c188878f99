/*
 * mpc_oracle.h -- CPU restatement of the reference's MPC hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the checker, never the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The shipped library (libmpcx.so) does not link,
 * include or call anything in this directory.
 *
 * PARITY UNPINNED: the reference (rdesarz/mpc-rs, Rust + nalgebra 0.33.2, Cargo.lock:438-451)
 * holds no tests, golden vectors or fixtures for this path and cannot be compiled in this
 * environment (no rustc/cargo, nalgebra and matrixmultiply 0.3.9 not vendored).  The
 * restatement below follows the reference source line by line; nalgebra's arithmetic is
 * restated from its published algorithm: column-major storage, left-associative products,
 * k-ascending mul-add accumulation, closed-form inverses for dimension <= 4 and LU with
 * partial pivoting above.  Last-bit agreement with a Rust build is not claimed.
 * The only outputs the reference itself holds are its two charts; the curves read out of them
 * (the two _png_trace.json files under tests/golden) pin this restatement at chart resolution (one pixel: about
 * 2e-4 of the step response, 2e-3 of the controlled output) and no further.
 *
 * All matrices are dense f64, column-major (element (i,j) of an R x C matrix at [j*R + i]),
 * which is what nalgebra's DMatrix::as_slice() yields.
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OK 0
#define ORC_ERR_INVERSION 1 /* Err("Matrix inversion failed"), controller.rs:93-95 */
#define ORC_ERR_SHAPE 2     /* nalgebra shape assertion would panic */

/* controller.rs:149-185 (Controller::new): W1 (vm x vm), W2 (vm x vm), W3 = W1^T (W2 W1) (vm x vm),
 * W4 (fr x fr).  Any output may be NULL. */
void orc_controller_weights(int m, int r, int f, int v, double q0, double q_other, double pred_weight,
                            double *W1, double *W2, double *W3, double *W4);

/* controller.rs:16-99 (Controller::form_lifted_matrices).
 * Outputs: O (fr x n), M (fr x vm), H = (M^T W4) M + W3 (vm x vm; the temporary handed to try_inverse at
 * controller.rs:93), gain = ((H^-1 M^T) W4) (vm x fr).  H and gain may be NULL; when gain is NULL the
 * inversion is skipped and the call cannot fail.  Returns ORC_OK or ORC_ERR_INVERSION. */
int orc_form_lifted_matrices(const double *A, const double *B, const double *C, int n, int m, int r, int f, int v,
                             const double *W3, const double *W4, double *O, double *M, double *H, double *gain);

/* Gradient of the condensed QP, fvec = (M^T W4) (O x0 - ref)  (vm).  The reference never materialises it:
 * controller.rs:125-127 forms s = ref - O x and controller.rs:96,130 applies (H^-1 M^T) W4 to it, so
 * u_seq = -H^-1 fvec. */
void orc_gradient(const double *O, const double *M, const double *W4, const double *x0, const double *ref, int n, int m,
                  int r, int f, int v, double *fvec);

/* controller.rs:117-135 (Controller::compute_control_input): s = traj[t..t+f] - O x; u_seq = gain s.
 * traj_total is the (T x 1) trajectory column.  Writes the full sequence (vm) to u_seq when not NULL and returns
 * u_seq[0], the one value the reference hands back. */
double orc_compute_control_input(const double *O, const double *gain, const double *traj_total, int timestep,
                                 const double *x, int n, int m, int r, int f, int v, double *u_seq);

/* controller.rs:101-115 (Controller::propagate_dynamics): x_next = A x + B u, y = C x. */
void orc_propagate_dynamics(const double *A, const double *B, const double *C, int n, int m, int r, const double *u,
                            const double *x, double *x_next, double *y);

/* simulator.rs:5-35 (system_simulate): U is m x T (time along columns), X is n x (T+1), Y is r x T. */
void orc_system_simulate(const double *A, const double *B, const double *C, int n, int m, int r, const double *U, int T,
                         const double *x0, double *Y, double *X);

/* simulator.rs:50-63 (step): number of samples floor(duration / dt). */
int orc_step_n_samples(double duration, double dt);

/* model.rs:55-58 / 138-141: A = (I - Ac dt)^-1, B = A (Bc dt).  Returns ORC_ERR_INVERSION where the reference
 * would panic on unwrap(). */
int orc_c2d_backward_euler(const double *Ac, const double *Bc, int n, int m, double dt, double *A, double *B);

/* model.rs:47-52 (dc_motor): fills Ac (2x2), Bc (2x1), Cc (1x2) from (b, j, k, l, r). */
void orc_dc_motor_continuous(double b, double j, double k, double l, double r, double *Ac, double *Bc, double *Cc);
/* model.rs:128-135 (mpc two-mass): fills Ac (4x4), Bc (4x1), Cc (1x4) from (m1, m2, k1, k2, d1, d2). */
void orc_two_mass_continuous(double m1, double m2, double k1, double k2, double d1, double d2, double *Ac, double *Bc,
                             double *Cc);

/* trajectory.rs:3-11 (generate_pulse_trajectory).  Returns ORC_ERR_SHAPE where copy_from would panic
 * (time_steps - 2*time_steps/3 != time_steps/3). */
int orc_pulse_trajectory(int time_steps, double *traj);

/* main.rs:51-71: closed loop of compute_control_input + propagate_dynamics for `steps` iterations from x0.
 * states is n x (steps+1), outputs is r x (steps+1) (column 0 = C x0, main.rs:40-41), inputs has `steps` values. */
void orc_closed_loop(const double *A, const double *B, const double *C, int n, int m, int r, int f, int v,
                     const double *O, const double *gain, const double *traj_total, int steps, const double *x0,
                     double *states, double *outputs, double *inputs);
/* The loop for any m, r (window stacked time-major; the reference itself is SISO): ref r x (steps + f) column-major. */
void orc_closed_loop_mimo(const double *A, const double *B, const double *C, int n, int m, int r, int f, int v,
                          const double *O, const double *gain, const double *ref, int steps, const double *x0, double *X,
                          double *U, double *Y);

/* try_inverse as nalgebra does it (closed forms up to 4x4, LU with partial pivoting above).
 * a is n x n, overwritten with its inverse; returns 0 on failure. */
int orc_try_inverse(double *a, int n);

/* Batch drivers over instance-major (AoS) buffers: instance b's matrices start at b * (rows*cols).
 * W3 / W4 are shared.  Each instance runs the functions above unchanged; `threads` > 1 uses OpenMP over
 * instances.  Any of Phi/Gamma/H/fvec may be NULL.  Returns the number of threads actually used. */
int orc_qp_build_batch(int64_t batch, int n, int m, int r, int f, int v, const double *A, const double *B,
                       const double *C, const double *W3, const double *W4, const double *x0, const double *ref,
                       double *Phi, double *Gamma, double *H, double *fvec, int threads);
int orc_rollout_batch(int64_t batch, int n, int m, int r, int64_t T, const double *A, const double *B, const double *C,
                      const double *x0, const double *U, int u_shared, double *X, double *Y, int threads);

#ifdef __cplusplus
}
#endif
/* A CPU implementation written for speed (O(f) recurrences, block-diagonal W4 given as its r x r block Q, one scratch
 * buffer per thread) with the same outputs as orc_qp_build_batch: the second, "optimised" CPU baseline of bench.py.
 * It does not keep the reference's loop structure and is therefore not the parity oracle. */
int orc_qp_build_batch_fast(int64_t batch, int n, int m, int r, int f, int v, const double *A, const double *B,
                            const double *C, const double *W3, const double *Q, const double *x0, const double *ref,
                            double *Phi, double *Gamma, double *H, double *fvec, int threads);

#endif
