//! src/control/simulator.rs of the reference; the rollout is one `mpcx_rollout_f64` call.
extern crate nalgebra as na;

use super::device::{dims, with_ctx};
use super::model::DiscreteStateSpaceModel;
use mpcx_sys as sys;

/// simulator.rs:5-35 (private there as here): returns (mat_y r x T, mat_x n x (T+1)), in that order.
fn system_simulate(model: &impl DiscreteStateSpaceModel, mat_u: &na::DMatrix<f64>, x0: &na::DVector<f64>)
                   -> (na::DMatrix<f64>, na::DMatrix<f64>) {
    let (a, b, c) = (model.get_mat_a(), model.get_mat_b(), model.get_mat_c());
    let (n, m, r, t) = (a.nrows(), b.ncols(), c.nrows(), mat_u.ncols());
    let mut mat_x = na::DMatrix::<f64>::zeros(n, t + 1);
    let mut mat_y = na::DMatrix::<f64>::zeros(r, t);
    if t == 0 {
        return (mat_y, mat_x); // the loop body never runs: both stay zero (simulator.rs:13-15)
    }
    assert!(mat_u.nrows() == m && x0.nrows() == n && a.ncols() == n && b.nrows() == n && c.ncols() == n,
            "Matrix multiplication dimensions mismatch.");
    let d = dims(n, m, r, 1, 1, 1, t as i64);
    with_ctx("mpcx_rollout_f64", |ctx| unsafe {
        sys::mpcx_rollout_f64(ctx, &d, a.as_ptr(), b.as_ptr(), c.as_ptr(), x0.as_ptr(), mat_u.as_ptr(),
                              mat_x.as_mut_ptr(), mat_y.as_mut_ptr())
    });
    (mat_y, mat_x)
}

/// simulator.rs:37-48: row 0 of the outputs.
pub fn compute_system_response(model: &impl DiscreteStateSpaceModel, input: &na::DMatrix<f64>,
                               initial_state: &na::DVector<f64>) -> Vec<f64> {
    let (response, _) = system_simulate(model, input, initial_state);
    response.row(0).iter().cloned().collect()
}

/// simulator.rs:50-63: zero initial state, floor(duration / dt) samples of a unit input; (response, input, states).
pub fn step(model: &impl DiscreteStateSpaceModel, duration: f64) -> (na::DMatrix<f64>, na::DMatrix<f64>, na::DMatrix<f64>) {
    let initial_state = na::DVector::<f64>::zeros(model.get_mat_a().nrows());
    let n_samples = (duration / model.get_sampling_dt()).floor() as usize;
    let input = na::DMatrix::<f64>::from_element(1, n_samples, 1.0);
    let (response, states) = system_simulate(model, &input, &initial_state);
    (response, input, states)
}
