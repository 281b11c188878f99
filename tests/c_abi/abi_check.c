/* Plain C99 caller of libmpcx.so through include/mpcx.h: what a -sys crate's bindgen output, cgo or any other FFI sees.
 * Builds the reference scenario (src/main.rs:14-71: DC motor, dt = 0.05, f = v = 20, 280 steps) entirely through the
 * C ABI with host pointers and prints a few values; tests/test_gpu_c_abi.py compares them with the oracle.
 * Without a GPU it checks that mpcx_create reports MPCX_ERR_NO_DEVICE (or a HIP error) and exits 3. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpcx.h"

#define F 20
#define STEPS 280

int main(void) {
    mpcx_ctx *ctx = NULL;
    int rc = mpcx_create(&ctx, 0);
    if (rc != MPCX_OK) {
        printf("create rc=%d\n", rc);
        return 3;
    }
    /* dc_motor::Parameters::default(), model.rs:24-33, in the order b, j, k, l, r */
    const double prm[5] = {0.1, 0.01, 0.01, 0.5, 1.0};
    double A[4], B[2], C[2];
    int32_t info = -1;
    rc = mpcx_plants_f64(ctx, MPCX_PLANT_DC_MOTOR, 1, 1, prm, 0.05, A, B, C, &info);
    if (rc != MPCX_OK || info != 0) { printf("plants rc=%d info=%d %s\n", rc, info, mpcx_last_error(ctx)); return 1; }

    /* weights of Controller::new (controller.rs:149-185) for m = r = 1: W3 tridiagonal, W4 block = pred_weight */
    const double q0 = 0.0000000011, qo = 0.0001, pw = 10.0;
    static double W3[F * F];
    memset(W3, 0, sizeof W3);
    for (int i = 0; i < F; ++i) {
        W3[i * F + i] = (i == 0) ? q0 + qo : (i == F - 1 ? qo : 2.0 * qo);
        if (i + 1 < F) { W3[i * F + i + 1] = -qo; W3[(i + 1) * F + i] = -qo; }
    }
    mpcx_dims d;
    memset(&d, 0, sizeof d);
    d.n = 2; d.m = 1; d.r = 1; d.f = F; d.v = F; d.batch = 1; d.interleave = MPCX_INTERLEAVE_AOS;
    d.weights_mode = MPCX_WEIGHTS_BLOCK;
    static double Phi[F * 2], Gamma[F * F], H[F * F], gain[F * F];
    rc = mpcx_qp_build_f64(ctx, &d, A, B, C, W3, &pw, NULL, NULL, Phi, Gamma, H, NULL);
    if (rc != MPCX_OK) { printf("build rc=%d %s\n", rc, mpcx_last_error(ctx)); return 1; }
    rc = mpcx_qp_gain_f64(ctx, &d, H, Gamma, &pw, gain, &info);
    if (rc != MPCX_OK || info != 0) { printf("gain rc=%d info=%d %s\n", rc, info, mpcx_last_error(ctx)); return 1; }

    /* trajectory.rs:3-11 with time_steps = 300 */
    static double traj[STEPS + F];
    for (int i = 0; i < STEPS + F; ++i) traj[i] = (i < 100 || i >= 200) ? 1.0 : 0.0;
    const double x0[2] = {0.0, 0.0};
    static double X[2 * (STEPS + 1)], U[STEPS], Y[STEPS];
    d.T = STEPS;
    d.u_shared = 1;
    rc = mpcx_closed_loop_f64(ctx, &d, A, B, C, x0, Phi, gain, traj, X, U, Y);
    if (rc != MPCX_OK) { printf("loop rc=%d %s\n", rc, mpcx_last_error(ctx)); return 1; }

    /* argument errors come back as codes, never as aborts */
    d.n = 0;
    rc = mpcx_qp_build_f64(ctx, &d, A, B, C, W3, &pw, NULL, NULL, Phi, Gamma, H, NULL);
    printf("bad_dims rc=%d\n", rc);
    printf("version %d\n", mpcx_version());
    printf("A %.17g %.17g %.17g %.17g\n", A[0], A[1], A[2], A[3]);
    printf("H00 %.17g\n", H[0]);
    printf("gain00 %.17g\n", gain[0]);
    printf("y50 %.17g y150 %.17g y279 %.17g\n", Y[50], Y[150], Y[279]);
    printf("u0 %.17g u279 %.17g\n", U[0], U[279]);
    mpcx_destroy(ctx);
    return 0;
}
