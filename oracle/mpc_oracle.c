/*
 * mpc_oracle.c -- CPU restatement of the reference's MPC hot path.  TEST INFRASTRUCTURE ONLY; see the
 * header for who may use it and for the "parity unpinned" statement.
 *
 * The functions keep the reference's algorithmic structure on purpose (powers of A re-multiplied from the
 * identity for every block row, dense W4, one heap-allocated temporary per matrix product) because the same
 * code is the timed CPU baseline.  Every function cites the reference lines it follows
 * (paths relative to the reference checkout, src/control/...).
 */
#include "mpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------------------------------
 * A minimal heap-backed column-major matrix, standing in for nalgebra::DMatrix<f64> (VecStorage).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int rows, cols;
    double *d;
} mat;

static mat mat_zeros(int rows, int cols) {
    mat x;
    x.rows = rows;
    x.cols = cols;
    x.d = (double *)calloc((size_t)rows * (size_t)cols > 0 ? (size_t)rows * (size_t)cols : 1, sizeof(double));
    return x;
}
static void mat_free(mat *x) {
    free(x->d);
    x->d = NULL;
}
static mat mat_from(const double *src, int rows, int cols) {
    mat x = mat_zeros(rows, cols);
    memcpy(x.d, src, sizeof(double) * (size_t)rows * (size_t)cols);
    return x;
}
static mat mat_identity(int n) {
    mat x = mat_zeros(n, n);
    for (int i = 0; i < n; ++i) x.d[(size_t)i * n + i] = 1.0;
    return x;
}
#define AT(x, i, j) ((x).d[(size_t)(j) * (size_t)(x).rows + (size_t)(i)])

/* nalgebra `&a * &b` for dynamic matrices: a fresh matrix, each output column accumulated as
 * sum over k (ascending) of a[:,k] * b[k,j] (the gemv/axpy path nalgebra takes for small operands). */
static mat mat_mul(const mat *a, const mat *b) {
    mat c = mat_zeros(a->rows, b->cols);
    for (int j = 0; j < b->cols; ++j) {
        for (int k = 0; k < a->cols; ++k) {
            const double bkj = AT(*b, k, j);
            const double *ak = a->d + (size_t)k * a->rows;
            double *cj = c.d + (size_t)j * c.rows;
            for (int i = 0; i < a->rows; ++i) cj[i] += ak[i] * bkj;
        }
    }
    return c;
}
static mat mat_transpose(const mat *a) {
    mat t = mat_zeros(a->cols, a->rows);
    for (int j = 0; j < a->cols; ++j)
        for (int i = 0; i < a->rows; ++i) AT(t, j, i) = AT(*a, i, j);
    return t;
}
/* view_mut((r0,c0),(rows,cols)).copy_from(src) */
static void mat_set_block(mat *dst, int r0, int c0, const mat *src) {
    for (int j = 0; j < src->cols; ++j)
        for (int i = 0; i < src->rows; ++i) AT(*dst, r0 + i, c0 + j) = AT(*src, i, j);
}

/* ---------------------------------------------------------------------------------------------------
 * try_inverse (nalgebra linalg/inverse.rs, restated from its published algorithm): closed forms for
 * dimensions 1..4, LU with partial pivoting above; fails only on an exactly zero determinant / pivot.
 * ------------------------------------------------------------------------------------------------- */
static int inverse4(const double *m, double *out) {
    out[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
             m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    out[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
             m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    out[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
             m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    out[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
             m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    out[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
             m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    out[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
             m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    out[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
             m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    out[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
             m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    out[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
             m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    out[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
             m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    out[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
              m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    out[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
              m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    out[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
              m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    out[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
              m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    out[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
              m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    out[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
              m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * out[0] + m[1] * out[4] + m[2] * out[8] + m[3] * out[12];
    if (det == 0.0) return 0;
    const double inv_det = 1.0 / det;
    for (int i = 0; i < 16; ++i) out[i] *= inv_det;
    return 1;
}

static int inverse_lu(double *a, int n) {
    /* LU with partial (row) pivoting, then solve for the identity. */
    int *perm = (int *)malloc(sizeof(int) * (size_t)n);
    double *lu = (double *)malloc(sizeof(double) * (size_t)n * n);
    memcpy(lu, a, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(lu[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            const double cand = fabs(lu[(size_t)k * n + i]);
            if (cand > best) {
                best = cand;
                piv = i;
            }
        }
        if (lu[(size_t)k * n + piv] == 0.0) {
            free(perm);
            free(lu);
            return 0;
        }
        if (piv != k) {
            for (int j = 0; j < n; ++j) {
                const double t = lu[(size_t)j * n + k];
                lu[(size_t)j * n + k] = lu[(size_t)j * n + piv];
                lu[(size_t)j * n + piv] = t;
            }
            const int t = perm[k];
            perm[k] = perm[piv];
            perm[piv] = t;
        }
        const double inv_p = 1.0 / lu[(size_t)k * n + k];
        for (int i = k + 1; i < n; ++i) lu[(size_t)k * n + i] *= inv_p;
        for (int j = k + 1; j < n; ++j) {
            const double ukj = lu[(size_t)j * n + k];
            for (int i = k + 1; i < n; ++i) lu[(size_t)j * n + i] -= lu[(size_t)k * n + i] * ukj;
        }
    }
    for (int c = 0; c < n; ++c) {
        double *x = a + (size_t)c * n;
        for (int i = 0; i < n; ++i) x[i] = (perm[i] == c) ? 1.0 : 0.0;
        for (int k = 0; k < n; ++k) /* forward, unit lower */
            for (int i = k + 1; i < n; ++i) x[i] -= lu[(size_t)k * n + i] * x[k];
        for (int k = n - 1; k >= 0; --k) { /* backward */
            x[k] /= lu[(size_t)k * n + k];
            for (int i = 0; i < k; ++i) x[i] -= lu[(size_t)k * n + i] * x[k];
        }
    }
    free(perm);
    free(lu);
    return 1;
}

int orc_try_inverse(double *a, int n) {
    if (n == 0) return 1;
    if (n == 1) {
        if (a[0] == 0.0) return 0;
        a[0] = 1.0 / a[0];
        return 1;
    }
    if (n == 2) {
        const double m11 = a[0], m21 = a[1], m12 = a[2], m22 = a[3];
        const double det = m11 * m22 - m21 * m12;
        if (det == 0.0) return 0;
        a[0] = m22 / det;
        a[1] = -m21 / det;
        a[2] = -m12 / det;
        a[3] = m11 / det;
        return 1;
    }
    if (n == 3) {
        const double m11 = a[0], m21 = a[1], m31 = a[2], m12 = a[3], m22 = a[4], m32 = a[5], m13 = a[6], m23 = a[7],
                     m33 = a[8];
        const double minor_m12_m23 = m22 * m33 - m32 * m23;
        const double minor_m11_m23 = m21 * m33 - m31 * m23;
        const double minor_m11_m22 = m21 * m32 - m31 * m22;
        const double det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
        if (det == 0.0) return 0;
        a[0] = minor_m12_m23 / det;
        a[3] = (m13 * m32 - m33 * m12) / det;
        a[6] = (m12 * m23 - m22 * m13) / det;
        a[1] = -minor_m11_m23 / det;
        a[4] = (m11 * m33 - m31 * m13) / det;
        a[7] = (m13 * m21 - m23 * m11) / det;
        a[2] = minor_m11_m22 / det;
        a[5] = (m12 * m31 - m32 * m11) / det;
        a[8] = (m11 * m22 - m21 * m12) / det;
        return 1;
    }
    if (n == 4) {
        double out[16];
        if (!inverse4(a, out)) return 0;
        memcpy(a, out, sizeof(out));
        return 1;
    }
    return inverse_lu(a, n);
}

/* ---------------------------------------------------------------------------------------------------
 * controller.rs:149-185 -- the weight matrices Controller::new builds.
 * ------------------------------------------------------------------------------------------------- */
void orc_controller_weights(int m, int r, int f, int v, double q0, double q_other, double pred_weight, double *W1,
                            double *W2, double *W3, double *W4) {
    const int vm = v * m, fr = f * r;
    /* :149-161  W1 = identity blocks on the diagonal, -identity on the block sub-diagonal */
    mat w1 = mat_zeros(vm, vm);
    for (int i = 0; i < v; ++i) {
        for (int a = 0; a < m; ++a) {
            AT(w1, i * m + a, i * m + a) = 1.0;
            if (i >= 1) AT(w1, i * m + a, (i - 1) * m + a) = -1.0;
        }
    }
    /* :164-175  W2 = block diagonal of from_element(m, m, q) -- every element of the block, not q*I */
    mat w2 = mat_zeros(vm, vm);
    for (int i = 0; i < v; ++i)
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < m; ++b) AT(w2, i * m + a, i * m + b) = (i == 0) ? q0 : q_other;
    /* :177  W3 = W1^T * (W2 * W1) */
    mat w2w1 = mat_mul(&w2, &w1);
    mat w1t = mat_transpose(&w1);
    mat w3 = mat_mul(&w1t, &w2w1);
    if (W1) memcpy(W1, w1.d, sizeof(double) * (size_t)vm * vm);
    if (W2) memcpy(W2, w2.d, sizeof(double) * (size_t)vm * vm);
    if (W3) memcpy(W3, w3.d, sizeof(double) * (size_t)vm * vm);
    /* :179-185  W4 = block diagonal of from_element(r, r, pred_weight) */
    if (W4) {
        memset(W4, 0, sizeof(double) * (size_t)fr * fr);
        for (int i = 0; i < f; ++i)
            for (int a = 0; a < r; ++a)
                for (int b = 0; b < r; ++b) W4[(size_t)(i * r + b) * fr + (i * r + a)] = pred_weight;
    }
    mat_free(&w1);
    mat_free(&w2);
    mat_free(&w2w1);
    mat_free(&w1t);
    mat_free(&w3);
}

/* ---------------------------------------------------------------------------------------------------
 * controller.rs:16-99 -- Controller::form_lifted_matrices.
 * ------------------------------------------------------------------------------------------------- */
int orc_form_lifted_matrices(const double *A, const double *B, const double *C, int n, int m, int r, int f, int v,
                             const double *W3, const double *W4, double *O, double *M, double *H, double *gain) {
    mat mat_a = mat_from(A, n, n), mat_b = mat_from(B, n, m), mat_c = mat_from(C, r, n);
    int status = ORC_OK;

    /* :32-44  lifted matrix O: block i = C * A^(i+1), pow_a accumulated as pow_a = pow_a * A */
    mat mat_o = mat_zeros(f * r, n);
    mat pow_a = mat_from(A, n, n);
    for (int i = 0; i < f; ++i) {
        if (i != 0) {
            mat next = mat_mul(&pow_a, &mat_a); /* :38 */
            mat_free(&pow_a);
            pow_a = next;
        }
        mat blk = mat_mul(&mat_c, &pow_a); /* :43 */
        mat_set_block(&mat_o, i * r, 0, &blk);
        mat_free(&blk);
    }

    /* :46-90  lifted matrix M */
    mat mat_m = mat_zeros(f * r, v * m);
    for (int i = 0; i < f; ++i) {
        if (i < v) { /* :51-62  until the control horizon */
            for (int j = 0; j <= i; ++j) {
                if (j == 0) {
                    mat_free(&pow_a);
                    pow_a = mat_identity(n); /* :54 */
                } else {
                    mat next = mat_mul(&pow_a, &mat_a); /* :56 */
                    mat_free(&pow_a);
                    pow_a = next;
                }
                mat cp = mat_mul(&mat_c, &pow_a); /* :61  (C * pow_a) * B, left-associative */
                mat blk = mat_mul(&cp, &mat_b);
                mat_set_block(&mat_m, i * r, (i - j) * m, &blk);
                mat_free(&cp);
                mat_free(&blk);
            }
        } else { /* :63-88  past the control horizon: the last input is held */
            for (int j = 0; j < v; ++j) {
                if (j == 0) {
                    mat sum_last = mat_zeros(n, n); /* :67 */
                    for (int s = 0; s < i - v + 2; ++s) {
                        if (s == 0) {
                            mat_free(&pow_a);
                            pow_a = mat_identity(n); /* :70 */
                        } else {
                            mat next = mat_mul(&pow_a, &mat_a); /* :72 */
                            mat_free(&pow_a);
                            pow_a = next;
                        }
                        for (int e = 0; e < n * n; ++e) sum_last.d[e] += pow_a.d[e]; /* :75 */
                    }
                    mat cs = mat_mul(&mat_c, &sum_last); /* :80 */
                    mat blk = mat_mul(&cs, &mat_b);
                    mat_set_block(&mat_m, i * r, (v - 1) * m, &blk);
                    mat_free(&cs);
                    mat_free(&blk);
                    mat_free(&sum_last);
                } else {
                    mat next = mat_mul(&pow_a, &mat_a); /* :82 */
                    mat_free(&pow_a);
                    pow_a = next;
                    mat cp = mat_mul(&mat_c, &pow_a); /* :86 */
                    mat blk = mat_mul(&cp, &mat_b);
                    mat_set_block(&mat_m, i * r, (v - 1 - j) * m, &blk);
                    mat_free(&cp);
                    mat_free(&blk);
                }
            }
        }
    }

    if (O) memcpy(O, mat_o.d, sizeof(double) * (size_t)f * r * n);
    if (M) memcpy(M, mat_m.d, sizeof(double) * (size_t)f * r * v * m);

    if (H || gain) {
        /* :92  tmp1 = M^T * W4 * M, evaluated left to right */
        mat w4 = mat_from(W4, f * r, f * r), w3 = mat_from(W3, v * m, v * m);
        mat mt = mat_transpose(&mat_m);
        mat mtw4 = mat_mul(&mt, &w4);
        mat tmp1 = mat_mul(&mtw4, &mat_m);
        for (int e = 0; e < v * m * v * m; ++e) tmp1.d[e] += w3.d[e]; /* :93  tmp1 + W3 */
        if (H) memcpy(H, tmp1.d, sizeof(double) * (size_t)v * m * v * m);
        if (gain) {
            if (!orc_try_inverse(tmp1.d, v * m)) { /* :93-95 */
                status = ORC_ERR_INVERSION;
            } else {
                mat t2mt = mat_mul(&tmp1, &mt); /* :96  (tmp2 * M^T) * W4 */
                mat g = mat_mul(&t2mt, &w4);
                memcpy(gain, g.d, sizeof(double) * (size_t)v * m * f * r);
                mat_free(&t2mt);
                mat_free(&g);
            }
        }
        mat_free(&w4);
        mat_free(&w3);
        mat_free(&mt);
        mat_free(&mtw4);
        mat_free(&tmp1);
    }
    mat_free(&mat_a);
    mat_free(&mat_b);
    mat_free(&mat_c);
    mat_free(&mat_o);
    mat_free(&mat_m);
    mat_free(&pow_a);
    return status;
}

/* controller.rs:125-127 (s = ref - O x) and the M^T W4 factor of controller.rs:96:
 * fvec = (M^T W4) (O x0 - ref). */
void orc_gradient(const double *O, const double *M, const double *W4, const double *x0, const double *ref, int n, int m,
                  int r, int f, int v, double *fvec) {
    mat mat_o = mat_from(O, f * r, n), mat_m = mat_from(M, f * r, v * m), w4 = mat_from(W4, f * r, f * r);
    mat x = mat_from(x0, n, 1);
    mat ox = mat_mul(&mat_o, &x);
    for (int i = 0; i < f * r; ++i) ox.d[i] = ox.d[i] - ref[i];
    mat mt = mat_transpose(&mat_m);
    mat mtw4 = mat_mul(&mt, &w4);
    mat g = mat_mul(&mtw4, &ox);
    memcpy(fvec, g.d, sizeof(double) * (size_t)v * m);
    mat_free(&mat_o);
    mat_free(&mat_m);
    mat_free(&w4);
    mat_free(&x);
    mat_free(&ox);
    mat_free(&mt);
    mat_free(&mtw4);
    mat_free(&g);
}

/* controller.rs:117-135 */
double orc_compute_control_input(const double *O, const double *gain, const double *traj_total, int timestep,
                                 const double *x, int n, int m, int r, int f, int v, double *u_seq) {
    mat mat_o = mat_from(O, f * r, n), g = mat_from(gain, v * m, f * r), xs = mat_from(x, n, 1);
    mat ox = mat_mul(&mat_o, &xs);                                      /* :126 */
    mat vec_s = mat_zeros(f * r, 1);                                    /* the slice :119-122 has f rows */
    for (int i = 0; i < f * r; ++i) vec_s.d[i] = traj_total[timestep + i] - ox.d[i]; /* :125-127 */
    mat seq = mat_mul(&g, &vec_s);                                      /* :130 */
    const double first = seq.d[0];                                      /* :132 */
    if (u_seq) memcpy(u_seq, seq.d, sizeof(double) * (size_t)v * m);
    mat_free(&mat_o);
    mat_free(&g);
    mat_free(&xs);
    mat_free(&ox);
    mat_free(&vec_s);
    mat_free(&seq);
    return first;
}

/* controller.rs:101-115 */
void orc_propagate_dynamics(const double *A, const double *B, const double *C, int n, int m, int r, const double *u,
                            const double *x, double *x_next, double *y) {
    mat a = mat_from(A, n, n), b = mat_from(B, n, m), c = mat_from(C, r, n);
    mat xs = mat_from(x, n, 1), us = mat_from(u, m, 1);
    mat ax = mat_mul(&a, &xs), bu = mat_mul(&b, &us), cx = mat_mul(&c, &xs);
    for (int i = 0; i < n; ++i) x_next[i] = ax.d[i] + bu.d[i]; /* :110 */
    for (int i = 0; i < r; ++i) y[i] = cx.d[i];                /* :112 */
    mat_free(&a);
    mat_free(&b);
    mat_free(&c);
    mat_free(&xs);
    mat_free(&us);
    mat_free(&ax);
    mat_free(&bu);
    mat_free(&cx);
}

/* simulator.rs:5-35 */
void orc_system_simulate(const double *A, const double *B, const double *C, int n, int m, int r, const double *U, int T,
                         const double *x0, double *Y, double *X) {
    mat a = mat_from(A, n, n), b = mat_from(B, n, m), c = mat_from(C, r, n);
    memset(X, 0, sizeof(double) * (size_t)n * ((size_t)T + 1)); /* :13 */
    memset(Y, 0, sizeof(double) * (size_t)r * (size_t)T);       /* :14 */
    for (int i = 0; i < T; ++i) {
        if (i == 0) memcpy(X, x0, sizeof(double) * (size_t)n); /* :17 */
        mat xi = mat_from(X + (size_t)i * n, n, 1);            /* :27 */
        mat ui = mat_from(U + (size_t)i * m, m, 1);
        mat cx = mat_mul(&c, &xi); /* :18 / :25 */
        mat ax = mat_mul(&a, &xi); /* :21 / :29 */
        mat bu = mat_mul(&b, &ui);
        for (int k = 0; k < r; ++k) Y[(size_t)i * r + k] = cx.d[k];
        for (int k = 0; k < n; ++k) X[(size_t)(i + 1) * n + k] = ax.d[k] + bu.d[k];
        mat_free(&xi);
        mat_free(&ui);
        mat_free(&cx);
        mat_free(&ax);
        mat_free(&bu);
    }
    mat_free(&a);
    mat_free(&b);
    mat_free(&c);
}

/* simulator.rs:57 */
int orc_step_n_samples(double duration, double dt) { return (int)floor(duration / dt); }

/* model.rs:55-58 and :138-141 */
int orc_c2d_backward_euler(const double *Ac, const double *Bc, int n, int m, double dt, double *A, double *B) {
    mat t = mat_zeros(n, n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) AT(t, i, j) = ((i == j) ? 1.0 : 0.0) - Ac[(size_t)j * n + i] * dt; /* I - Ac.scale(dt) */
    if (!orc_try_inverse(t.d, n)) {
        mat_free(&t);
        return ORC_ERR_INVERSION;
    }
    mat bs = mat_zeros(n, m);
    for (int e = 0; e < n * m; ++e) bs.d[e] = Bc[e] * dt; /* Bc.scale(dt) */
    mat bd = mat_mul(&t, &bs);
    memcpy(A, t.d, sizeof(double) * (size_t)n * n);
    memcpy(B, bd.d, sizeof(double) * (size_t)n * m);
    mat_free(&t);
    mat_free(&bs);
    mat_free(&bd);
    return ORC_OK;
}

/* model.rs:47-52 */
void orc_dc_motor_continuous(double b, double j, double k, double l, double r, double *Ac, double *Bc, double *Cc) {
    /* dmatrix![-b/j, k/j; -k/l, -r/l] written row by row, stored column-major */
    Ac[0] = -b / j;
    Ac[2] = k / j;
    Ac[1] = -k / l;
    Ac[3] = -r / l;
    Bc[0] = 0.0;
    Bc[1] = 1.0 / l;
    Cc[0] = 1.0;
    Cc[1] = 0.0;
}

/* model.rs:128-135 */
void orc_two_mass_continuous(double m1, double m2, double k1, double k2, double d1, double d2, double *Ac, double *Bc,
                             double *Cc) {
    const double rows[4][4] = {{0.0, 1.0, 0.0, 0.0},
                               {-(k1 + k2) / m1, -(d1 + d2) / m1, k2 / m1, d2 / m1},
                               {0.0, 0.0, 0.0, 1.0},
                               {k2 / m2, d2 / m2, -k2 / m2, -d2 / m2}};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) Ac[j * 4 + i] = rows[i][j];
    Bc[0] = 0.0;
    Bc[1] = 0.0;
    Bc[2] = 0.0;
    Bc[3] = 1.0 / m2;
    Cc[0] = 1.0;
    Cc[1] = 0.0;
    Cc[2] = 0.0;
    Cc[3] = 0.0;
}

/* trajectory.rs:3-11 */
int orc_pulse_trajectory(int time_steps, double *traj) {
    const int third = time_steps / 3;
    if (time_steps - 2 * time_steps / 3 != third) return ORC_ERR_SHAPE; /* copy_from shape mismatch panics, :7-8 */
    for (int i = 0; i < time_steps; ++i) traj[i] = 0.0;
    for (int i = 0; i < third; ++i) traj[i] = 1.0;
    for (int i = 2 * time_steps / 3; i < time_steps; ++i) traj[i] = 1.0;
    return ORC_OK;
}

/* main.rs:51-71 */
void orc_closed_loop(const double *A, const double *B, const double *C, int n, int m, int r, int f, int v,
                     const double *O, const double *gain, const double *traj_total, int steps, const double *x0,
                     double *states, double *outputs, double *inputs) {
    memcpy(states, x0, sizeof(double) * (size_t)n); /* main.rs:33-34 */
    for (int a = 0; a < r; ++a) {                   /* main.rs:40-41  outputs row 0 = C x0 */
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc += C[(size_t)k * r + a] * x0[k];
        outputs[a] = acc;
    }
    double *u = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
    for (int t = 0; t < steps; ++t) {
        const double *x = states + (size_t)t * n;
        u[0] = orc_compute_control_input(O, gain, traj_total, t, x, n, m, r, f, v, NULL); /* main.rs:53 */
        orc_propagate_dynamics(A, B, C, n, m, r, u, x, states + (size_t)(t + 1) * n,
                               outputs + (size_t)(t + 1) * r); /* main.rs:56-63 */
        inputs[t] = u[0];
    }
    free(u);
}

/* The same loop for any m, r (the reference handles m = r = 1 only: controller.rs:119-122 slices a T x 1 trajectory and
 * :131-132 takes element (0,0)): the window is stacked time-major, vec_s[i r + a] = ref[a][t + i] - (O x)[i r + a]
 * (:125-127), u = the first m entries of gain * vec_s (:130), then propagate_dynamics (:106-112).  Same products, same
 * order of operations as orc_compute_control_input / orc_propagate_dynamics.  ref is r x (steps + f), column-major;
 * X is n x (steps + 1), U m x steps, Y r x steps (Y[:, t] = C x_t). */
void orc_closed_loop_mimo(const double *A, const double *B, const double *C, int n, int m, int r, int f, int v,
                          const double *O, const double *gain, const double *ref, int steps, const double *x0, double *X,
                          double *U, double *Y) {
    mat mat_o = mat_from(O, f * r, n), g = mat_from(gain, v * m, f * r);
    memcpy(X, x0, sizeof(double) * (size_t)n);
    for (int t = 0; t < steps; ++t) {
        const double *x = X + (size_t)t * n;
        mat xs = mat_from(x, n, 1);
        mat ox = mat_mul(&mat_o, &xs);
        mat vec_s = mat_zeros(f * r, 1);
        for (int i = 0; i < f; ++i)
            for (int a = 0; a < r; ++a) vec_s.d[i * r + a] = ref[(size_t)(t + i) * r + a] - ox.d[i * r + a];
        mat seq = mat_mul(&g, &vec_s);
        for (int k = 0; k < m; ++k) U[(size_t)t * m + k] = seq.d[k];
        orc_propagate_dynamics(A, B, C, n, m, r, U + (size_t)t * m, x, X + (size_t)(t + 1) * n, Y + (size_t)t * r);
        mat_free(&xs);
        mat_free(&ox);
        mat_free(&vec_s);
        mat_free(&seq);
    }
    mat_free(&mat_o);
    mat_free(&g);
}

/* ---------------------------------------------------------------------------------------------------
 * Batch drivers (instance-major buffers): the per-instance functions above, unchanged, in a loop.
 * ------------------------------------------------------------------------------------------------- */
int orc_qp_build_batch(int64_t batch, int n, int m, int r, int f, int v, const double *A, const double *B,
                       const double *C, const double *W3, const double *W4, const double *x0, const double *ref,
                       double *Phi, double *Gamma, double *H, double *fvec, int threads) {
    const size_t eA = (size_t)n * n, eB = (size_t)n * m, eC = (size_t)r * n;
    const size_t ePhi = (size_t)f * r * n, eG = (size_t)f * r * v * m, eH = (size_t)v * m * v * m;
    int used = 1;
#ifdef _OPENMP
    if (threads > 1) used = threads;
#pragma omp parallel for schedule(static) num_threads(used)
#endif
    for (int64_t b = 0; b < batch; ++b) {
        double *o = (double *)malloc(sizeof(double) * ePhi);
        double *g = (double *)malloc(sizeof(double) * eG);
        double *h = H ? H + (size_t)b * eH : NULL;
        orc_form_lifted_matrices(A + (size_t)b * eA, B + (size_t)b * eB, C + (size_t)b * eC, n, m, r, f, v, W3, W4, o, g,
                                 h, NULL);
        if (fvec) orc_gradient(o, g, W4, x0 + (size_t)b * n, ref + (size_t)b * f * r, n, m, r, f, v, fvec + (size_t)b * v * m);
        if (Phi) memcpy(Phi + (size_t)b * ePhi, o, sizeof(double) * ePhi);
        if (Gamma) memcpy(Gamma + (size_t)b * eG, g, sizeof(double) * eG);
        free(o);
        free(g);
    }
    (void)threads;
    return used;
}

int orc_rollout_batch(int64_t batch, int n, int m, int r, int64_t T, const double *A, const double *B, const double *C,
                      const double *x0, const double *U, int u_shared, double *X, double *Y, int threads) {
    int used = 1;
#ifdef _OPENMP
    if (threads > 1) used = threads;
#pragma omp parallel for schedule(static) num_threads(used)
#endif
    for (int64_t b = 0; b < batch; ++b) {
        const double *u = u_shared ? U : U + (size_t)b * m * (size_t)T;
        orc_system_simulate(A + (size_t)b * n * n, B + (size_t)b * n * m, C + (size_t)b * r * n, n, m, r, u, (int)T,
                            x0 + (size_t)b * n, Y + (size_t)b * r * (size_t)T, X + (size_t)b * n * ((size_t)T + 1));
    }
    (void)threads;
    return used;
}

/* ---------------------------------------------------------------------------------------------------
 * A CPU implementation one would write for speed, for bench.py's second baseline line: same outputs as
 * orc_qp_build_batch with W4 = blockdiag(Q), but Markov blocks and Phi from O(f) recurrences (w <- A w,
 * P <- P A) instead of the reference's re-multiplied powers, H through the block-Toeplitz recurrence
 * H[a][b] = H[a+1][b+1] + G_(f-1-a)^T Q G_(f-1-b) (f = v) or plain Gram sums (f > v), one scratch
 * buffer per thread.  Not a restatement of the reference's loop structure, hence not the parity oracle:
 * tests compare it against orc_qp_build_batch.
 * ------------------------------------------------------------------------------------------------- */
int orc_qp_build_batch_fast(int64_t batch, int n, int m, int r, int f, int v, const double *A, const double *B,
                            const double *C, const double *W3, const double *Q, const double *x0, const double *ref,
                            double *Phi, double *Gamma, double *H, double *fvec, int threads) {
    const size_t eA = (size_t)n * n, eB = (size_t)n * m, eC = (size_t)r * n;
    const int fr = f * r, vm = v * m, nl = f - v + 1;
    const size_t ePhi = (size_t)fr * n, eG = (size_t)fr * vm, eH = (size_t)vm * vm;
    int used = 1;
#ifdef _OPENMP
    if (threads > 1) used = threads;
#pragma omp parallel num_threads(used)
#endif
    {
        /* per thread: G (fr x m), QG (fr x m), S (nl*r x m), QS, W (n x m) x2, P (r x n) x2, s (fr), z (fr), gam (fr x vm) */
        double *buf = (double *)malloc(sizeof(double) * ((size_t)4 * fr * m + 2 * n * m + 2 * r * n + 2 * fr + eG + 16));
        double *G = buf, *QG = G + (size_t)fr * m, *S = QG + (size_t)fr * m, *QS = S + (size_t)fr * m;
        double *W0 = QS + (size_t)fr * m, *W1 = W0 + n * m, *P0 = W1 + n * m, *P1 = P0 + r * n;
        double *sv = P1 + r * n, *z = sv + fr, *gam = z + fr;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t b = 0; b < batch; ++b) {
            const double *a = A + (size_t)b * eA, *bm = B + (size_t)b * eB, *c = C + (size_t)b * eC;
            double *phi = Phi ? Phi + (size_t)b * ePhi : NULL;
            double *gout = Gamma ? Gamma + (size_t)b * eG : gam;
            /* recurrences: G_k = C W_k (W_0 = B), Phi_k = P_k A (P_0 = C) */
            memcpy(W0, bm, sizeof(double) * n * m);
            memcpy(P0, c, sizeof(double) * r * n);
            double *w = W0, *wn = W1, *pk = P0, *pn = P1;
            for (int k = 0; k < f; ++k) {
                for (int j = 0; j < m; ++j)
                    for (int i = 0; i < r; ++i) {
                        double acc = 0.0;
                        for (int q = 0; q < n; ++q) acc += c[(size_t)q * r + i] * w[(size_t)j * n + q];
                        G[(size_t)j * fr + k * r + i] = acc;
                    }
                for (int j = 0; j < m; ++j)
                    for (int i = 0; i < n; ++i) {
                        double acc = 0.0;
                        for (int q = 0; q < n; ++q) acc += a[(size_t)q * n + i] * w[(size_t)j * n + q];
                        wn[(size_t)j * n + i] = acc;
                    }
                for (int j = 0; j < n; ++j)
                    for (int i = 0; i < r; ++i) {
                        double acc = 0.0;
                        for (int q = 0; q < n; ++q) acc += pk[(size_t)q * r + i] * a[(size_t)j * n + q];
                        pn[(size_t)j * r + i] = acc;
                        if (phi) phi[(size_t)j * fr + k * r + i] = acc;
                    }
                if (fvec) {
                    for (int i = 0; i < r; ++i) {
                        double acc = 0.0;
                        for (int q = 0; q < n; ++q) acc += pn[(size_t)q * r + i] * x0[(size_t)b * n + q];
                        sv[k * r + i] = acc - ref[(size_t)b * fr + k * r + i];
                    }
                }
                double *t = w; w = wn; wn = t;
                t = pk; pk = pn; pn = t;
            }
            /* S_j, Q G, Q S, z = Q s */
            for (int j = 0; j < m; ++j)
                for (int i = 0; i < r; ++i) {
                    double run = 0.0;
                    for (int k = 0; k < nl; ++k) { run += G[(size_t)j * fr + k * r + i]; S[(size_t)j * fr + k * r + i] = run; }
                }
            for (int j = 0; j < m; ++j)
                for (int k = 0; k < f; ++k)
                    for (int i = 0; i < r; ++i) {
                        double acc = 0.0, acs = 0.0;
                        for (int q = 0; q < r; ++q) {
                            acc += Q[(size_t)q * r + i] * G[(size_t)j * fr + k * r + q];
                            if (k < nl) acs += Q[(size_t)q * r + i] * S[(size_t)j * fr + k * r + q];
                        }
                        QG[(size_t)j * fr + k * r + i] = acc;
                        if (k < nl) QS[(size_t)j * fr + k * r + i] = acs;
                    }
            if (fvec)
                for (int k = 0; k < f; ++k)
                    for (int i = 0; i < r; ++i) {
                        double acc = 0.0;
                        for (int q = 0; q < r; ++q) acc += Q[(size_t)q * r + i] * sv[k * r + q];
                        z[k * r + i] = acc;
                    }
            /* Gamma columns: shifted Markov vectors; the last block column holds S when f > v */
            for (int kb = 0; kb < v; ++kb)
                for (int j = 0; j < m; ++j) {
                    double *col = gout + (size_t)(kb * m + j) * fr;
                    const double *src = (kb == v - 1) ? S + (size_t)j * fr : G + (size_t)j * fr;
                    const int shift = kb * r;
                    memset(col, 0, sizeof(double) * shift);
                    memcpy(col + shift, src, sizeof(double) * (fr - shift));
                }
            if (fvec)
                for (int col = 0; col < vm; ++col) {
                    const double *gc = gout + (size_t)col * fr;
                    const int shift = (col / m) * r;
                    double acc = 0.0;
                    for (int t = shift; t < fr; ++t) acc += gc[t] * z[t];
                    fvec[(size_t)b * vm + col] = acc;
                }
            if (H) {
                double *h = H + (size_t)b * eH;
                /* column blocks from right to left; Toeplitz recurrence away from the last two block rows / columns */
                for (int bb = v - 1; bb >= 0; --bb)
                    for (int j = 0; j < m; ++j)
                        for (int aa = v - 1; aa >= 0; --aa)
                            for (int i = 0; i < m; ++i) {
                                double acc;
                                const int rec = (nl == 1) ? (aa < v - 1 && bb < v - 1) : (aa < v - 2 && bb < v - 2);
                                if (rec) {
                                    acc = h[(size_t)((bb + 1) * m + j) * vm + (aa + 1) * m + i] -
                                          W3[(size_t)((bb + 1) * m + j) * vm + (aa + 1) * m + i];
                                    const double *ga = G + (size_t)i * fr + (size_t)(f - 1 - aa) * r;
                                    const double *qb = QG + (size_t)j * fr + (size_t)(f - 1 - bb) * r;
                                    for (int q = 0; q < r; ++q) acc += ga[q] * qb[q];
                                } else {
                                    acc = 0.0;
                                    const int i0 = aa > bb ? aa : bb;
                                    for (int i2 = i0; i2 < f; ++i2) {
                                        const double *ga = (aa == v - 1) ? S + (size_t)i * fr + (size_t)(i2 - (v - 1)) * r
                                                                         : G + (size_t)i * fr + (size_t)(i2 - aa) * r;
                                        const double *qb = (bb == v - 1) ? QS + (size_t)j * fr + (size_t)(i2 - (v - 1)) * r
                                                                         : QG + (size_t)j * fr + (size_t)(i2 - bb) * r;
                                        for (int q = 0; q < r; ++q) acc += ga[q] * qb[q];
                                    }
                                }
                                h[(size_t)(bb * m + j) * vm + aa * m + i] = acc + W3[(size_t)(bb * m + j) * vm + aa * m + i];
                            }
            }
        }
        free(buf);
    }
    (void)threads;
    return used;
}
