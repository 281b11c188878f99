//! Additive API (no counterpart in the reference): many independent plants per call -- where the GPU pays.
//! Slices hold one column-major matrix per plant back to back (`DMatrix::as_slice()` order), i.e. the library's
//! interleave = 1 layout; see `include/mpcx.h` for the batch-interleaved layouts the lane kernels are native in.
extern crate nalgebra as na;

use super::device::{dims, with_ctx};
use mpcx_sys as sys;

pub struct QpBatch {
    pub batch: usize,
    pub n: usize,
    pub m: usize,
    pub r: usize,
    pub f: usize,
    pub v: usize,
    pub phi: Vec<f64>,   // batch x (f r x n)
    pub gamma: Vec<f64>, // batch x (f r x v m)
    pub h: Vec<f64>,     // batch x (v m x v m)
    pub fvec: Vec<f64>,  // batch x (v m)
}

/// Phi, Gamma, H = Gamma^T blockdiag(Q) Gamma + W3 and fvec = Gamma^T blockdiag(Q) (Phi x0 - ref) for `batch` plants
/// (controller.rs:16-93, :96, :125-127).  `w3` is (v m) x (v m), `q` the r x r block of W4; both shared by the batch.
#[allow(clippy::too_many_arguments)]
pub fn build_qp_batched(batch: usize, n: usize, m: usize, r: usize, f: usize, v: usize, a: &[f64], b: &[f64], c: &[f64],
                        w3: &na::DMatrix<f64>, q: &na::DMatrix<f64>, x0: &[f64], reference: &[f64]) -> QpBatch {
    assert!(a.len() == batch * n * n && b.len() == batch * n * m && c.len() == batch * r * n, "A, B, C: one matrix per plant");
    assert!(x0.len() == batch * n && reference.len() == batch * f * r, "x0, ref: one vector per plant");
    assert!(w3.nrows() == v * m && w3.ncols() == v * m && q.nrows() == r && q.ncols() == r, "weight shapes");
    let mut out = QpBatch {
        batch, n, m, r, f, v,
        phi: vec![0.0; batch * f * r * n],
        gamma: vec![0.0; batch * f * r * v * m],
        h: vec![0.0; batch * v * m * v * m],
        fvec: vec![0.0; batch * v * m],
    };
    let d = dims(n, m, r, f, v, batch as i64, 0);
    with_ctx("mpcx_qp_build_f64", |ctx| unsafe {
        sys::mpcx_qp_build_f64(ctx, &d, a.as_ptr(), b.as_ptr(), c.as_ptr(), w3.as_ptr(), q.as_ptr(), x0.as_ptr(),
                               reference.as_ptr(), out.phi.as_mut_ptr(), out.gamma.as_mut_ptr(), out.h.as_mut_ptr(),
                               out.fvec.as_mut_ptr())
    });
    out
}

/// X (n x (T+1)) and Y (r x T) per plant for inputs U (m x T per plant): simulator.rs:5-35 for `batch` plants.
#[allow(clippy::too_many_arguments)]
pub fn simulate_batched(batch: usize, n: usize, m: usize, r: usize, t: usize, a: &[f64], b: &[f64], c: &[f64], x0: &[f64],
                        u: &[f64]) -> (Vec<f64>, Vec<f64>) {
    assert!(a.len() == batch * n * n && b.len() == batch * n * m && c.len() == batch * r * n, "A, B, C: one matrix per plant");
    assert!(x0.len() == batch * n && u.len() == batch * m * t, "x0, U: one per plant");
    let mut x = vec![0.0; batch * n * (t + 1)];
    let mut y = vec![0.0; batch * r * t];
    let d = dims(n, m, r, 1, 1, batch as i64, t as i64);
    with_ctx("mpcx_rollout_f64", |ctx| unsafe {
        sys::mpcx_rollout_f64(ctx, &d, a.as_ptr(), b.as_ptr(), c.as_ptr(), x0.as_ptr(), u.as_ptr(), x.as_mut_ptr(), y.as_mut_ptr())
    });
    (x, y)
}

/// gain = H^-1 Gamma^T blockdiag(Q) per plant (controller.rs:93-96) and `info` (1 = "Matrix inversion failed").
pub fn gain_batched(qp: &QpBatch, q: &na::DMatrix<f64>) -> (Vec<f64>, Vec<i32>) {
    let mut gain = vec![0.0; qp.batch * qp.v * qp.m * qp.f * qp.r];
    let mut info = vec![-1i32; qp.batch];
    let d = dims(qp.n, qp.m, qp.r, qp.f, qp.v, qp.batch as i64, 0);
    with_ctx("mpcx_qp_gain_f64", |ctx| unsafe {
        sys::mpcx_qp_gain_f64(ctx, &d, qp.h.as_ptr(), qp.gamma.as_ptr(), q.as_ptr(), gain.as_mut_ptr(), info.as_mut_ptr())
    });
    (gain, info)
}

/// The first m rows of every plant's gain (m x (f r) per plant, column-major): what the control move applies
/// (controller.rs:130-132), for the cost of the factorisation alone (`mpcx_qp_gain_rows_f64`).  Same `info`.
pub fn gain_rows_batched(qp: &QpBatch, q: &na::DMatrix<f64>) -> (Vec<f64>, Vec<i32>) {
    let mut rows = vec![0.0; qp.batch * qp.m * qp.f * qp.r];
    let mut info = vec![-1i32; qp.batch];
    let d = dims(qp.n, qp.m, qp.r, qp.f, qp.v, qp.batch as i64, 0);
    with_ctx("mpcx_qp_gain_rows_f64", |ctx| unsafe {
        sys::mpcx_qp_gain_rows_f64(ctx, &d, qp.h.as_ptr(), qp.gamma.as_ptr(), q.as_ptr(), rows.as_mut_ptr(), info.as_mut_ptr())
    });
    (rows, info)
}
