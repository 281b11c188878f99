// Closed-form inverses of 1x1 .. 4x4 matrices in registers.
//
// nalgebra's try_inverse (the call behind model.rs:56, :139 and controller.rs:93-95) does not eliminate for these
// sizes: it evaluates the determinant and the adjugate directly and fails exactly when `determinant == 0.0`.  The
// functions here evaluate the same expressions in the same order, with multiply and add kept separate (Rust does not
// contract them into fused multiply-adds), so that a matrix whose determinant cancels to exactly zero -- two equal
// columns, say -- is reported singular here as it is there.  Above 4x4 nalgebra eliminates with partial pivoting and
// so do the solver kernels.
#pragma once

namespace mpcx {

// a and inv are [row][col].  Returns false (inv untouched) when the determinant is exactly zero or not finite.
template <int N>
__device__ __forceinline__ bool inverse_closed_form(const double (&a)[N][N], double (&inv)[N][N]) {
#pragma clang fp contract(off)
    static_assert(N >= 1 && N <= 4, "closed forms exist for 1..4 only");
    if constexpr (N == 1) {
        const double d = a[0][0];
        if (d == 0.0 || !isfinite(d)) return false;
        inv[0][0] = 1.0 / d;
        return true;
    } else if constexpr (N == 2) {
        const double m11 = a[0][0], m12 = a[0][1], m21 = a[1][0], m22 = a[1][1];
        const double det = m11 * m22 - m21 * m12;
        if (det == 0.0 || !isfinite(det)) return false;
        inv[0][0] = m22 / det;
        inv[0][1] = -m12 / det;
        inv[1][0] = -m21 / det;
        inv[1][1] = m11 / det;
        return true;
    } else if constexpr (N == 3) {
        const double m11 = a[0][0], m12 = a[0][1], m13 = a[0][2];
        const double m21 = a[1][0], m22 = a[1][1], m23 = a[1][2];
        const double m31 = a[2][0], m32 = a[2][1], m33 = a[2][2];
        const double c11 = m22 * m33 - m32 * m23;  // minors along the first row
        const double c12 = m21 * m33 - m31 * m23;
        const double c13 = m21 * m32 - m31 * m22;
        const double det = m11 * c11 - m12 * c12 + m13 * c13;
        if (det == 0.0 || !isfinite(det)) return false;
        inv[0][0] = c11 / det;
        inv[0][1] = (m13 * m32 - m33 * m12) / det;
        inv[0][2] = (m12 * m23 - m22 * m13) / det;
        inv[1][0] = -c12 / det;
        inv[1][1] = (m11 * m33 - m31 * m13) / det;
        inv[1][2] = (m13 * m21 - m23 * m11) / det;
        inv[2][0] = c13 / det;
        inv[2][1] = (m12 * m31 - m32 * m11) / det;
        inv[2][2] = (m11 * m22 - m21 * m12) / det;
        return true;
    } else {
        // Adjugate by six-term cofactors.  e(k) is entry k of the column-major flattening (k = col * 4 + row); entry k
        // of the adjugate, flattened the same way, is a signed sum of triple products of the other entries.
#define E(k) a[(k) & 3][(k) >> 2]
#define T3(x, y, z) (E(x) * E(y) * E(z))
        double o[16];
        o[0] = T3(5, 10, 15) - T3(5, 11, 14) - T3(9, 6, 15) + T3(9, 7, 14) + T3(13, 6, 11) - T3(13, 7, 10);
        o[1] = -T3(1, 10, 15) + T3(1, 11, 14) + T3(9, 2, 15) - T3(9, 3, 14) - T3(13, 2, 11) + T3(13, 3, 10);
        o[2] = T3(1, 6, 15) - T3(1, 7, 14) - T3(5, 2, 15) + T3(5, 3, 14) + T3(13, 2, 7) - T3(13, 3, 6);
        o[3] = -T3(1, 6, 11) + T3(1, 7, 10) + T3(5, 2, 11) - T3(5, 3, 10) - T3(9, 2, 7) + T3(9, 3, 6);
        o[4] = -T3(4, 10, 15) + T3(4, 11, 14) + T3(8, 6, 15) - T3(8, 7, 14) - T3(12, 6, 11) + T3(12, 7, 10);
        o[5] = T3(0, 10, 15) - T3(0, 11, 14) - T3(8, 2, 15) + T3(8, 3, 14) + T3(12, 2, 11) - T3(12, 3, 10);
        o[6] = -T3(0, 6, 15) + T3(0, 7, 14) + T3(4, 2, 15) - T3(4, 3, 14) - T3(12, 2, 7) + T3(12, 3, 6);
        o[7] = T3(0, 6, 11) - T3(0, 7, 10) - T3(4, 2, 11) + T3(4, 3, 10) + T3(8, 2, 7) - T3(8, 3, 6);
        o[8] = T3(4, 9, 15) - T3(4, 11, 13) - T3(8, 5, 15) + T3(8, 7, 13) + T3(12, 5, 11) - T3(12, 7, 9);
        o[9] = -T3(0, 9, 15) + T3(0, 11, 13) + T3(8, 1, 15) - T3(8, 3, 13) - T3(12, 1, 11) + T3(12, 3, 9);
        o[10] = T3(0, 5, 15) - T3(0, 7, 13) - T3(4, 1, 15) + T3(4, 3, 13) + T3(12, 1, 7) - T3(12, 3, 5);
        o[11] = -T3(0, 5, 11) + T3(0, 7, 9) + T3(4, 1, 11) - T3(4, 3, 9) - T3(8, 1, 7) + T3(8, 3, 5);
        o[12] = -T3(4, 9, 14) + T3(4, 10, 13) + T3(8, 5, 14) - T3(8, 6, 13) - T3(12, 5, 10) + T3(12, 6, 9);
        o[13] = T3(0, 9, 14) - T3(0, 10, 13) - T3(8, 1, 14) + T3(8, 2, 13) + T3(12, 1, 10) - T3(12, 2, 9);
        o[14] = -T3(0, 5, 14) + T3(0, 6, 13) + T3(4, 1, 14) - T3(4, 2, 13) - T3(12, 1, 6) + T3(12, 2, 5);
        o[15] = T3(0, 5, 10) - T3(0, 6, 9) - T3(4, 1, 10) + T3(4, 2, 9) + T3(8, 1, 6) - T3(8, 2, 5);
        const double det = E(0) * o[0] + E(1) * o[4] + E(2) * o[8] + E(3) * o[12];
#undef T3
#undef E
        if (det == 0.0 || !isfinite(det)) return false;
        const double inv_det = 1.0 / det;
#pragma unroll
        for (int k = 0; k < 16; ++k) inv[k & 3][k >> 2] = o[k] * inv_det;
        return true;
    }
}

}  // namespace mpcx
