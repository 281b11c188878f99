//! The reference's `control::simulator` (src/control/simulator.rs): step response and system response of a plant.
//! Every trajectory is one `mpcx_rollout_f64` call on the device; nothing is propagated on the host.
extern crate nalgebra as na;

use super::device::{dims, with_ctx};
use super::model::DiscreteStateSpaceModel;
use mpcx_sys as sys;

type Mat = na::DMatrix<f64>;

/// States (n x (T+1)) and outputs (r x T) of the plant driven by the T columns of `inputs` from `start`
/// (simulator.rs:5-35, private there as well; returned as (outputs, states), the reference's order).
fn rollout_on_device(plant: &impl DiscreteStateSpaceModel, inputs: &Mat, start: &na::DVector<f64>) -> (Mat, Mat) {
    let (n, m, r) = (plant.get_mat_a().nrows(), plant.get_mat_b().ncols(), plant.get_mat_c().nrows());
    let horizon = inputs.ncols();
    let mut states = Mat::zeros(n, horizon + 1);
    let mut outputs = Mat::zeros(r, horizon);
    if horizon > 0 {
        // what nalgebra's product assertions would reject at simulator.rs:18-21
        let shapes_ok = inputs.nrows() == m && start.nrows() == n && plant.get_mat_a().ncols() == n
            && plant.get_mat_b().nrows() == n && plant.get_mat_c().ncols() == n;
        assert!(shapes_ok, "Matrix multiplication dimensions mismatch.");
        let d = dims(n, m, r, 1, 1, 1, horizon as i64);
        with_ctx("mpcx_rollout_f64", |ctx| unsafe {
            sys::mpcx_rollout_f64(ctx, &d, plant.get_mat_a().as_ptr(), plant.get_mat_b().as_ptr(), plant.get_mat_c().as_ptr(),
                                  start.as_ptr(), inputs.as_ptr(), states.as_mut_ptr(), outputs.as_mut_ptr())
        });
    } // else: no step runs and both stay zero (simulator.rs:13-15)
    (outputs, states)
}

/// simulator.rs:50-63: unit input for floor(duration / dt) samples from the zero state; (response, input, states).
pub fn step(model: &impl DiscreteStateSpaceModel, duration: f64) -> (Mat, Mat, Mat) {
    let samples = (duration / model.get_sampling_dt()).floor() as usize;
    let unit = Mat::from_element(1, samples, 1.0);
    let origin = na::DVector::<f64>::zeros(model.get_mat_a().nrows());
    let (response, states) = rollout_on_device(model, &unit, &origin);
    (response, unit, states)
}

/// simulator.rs:37-48: the first output row of the response to `input` from `initial_state`.
pub fn compute_system_response(model: &impl DiscreteStateSpaceModel, input: &na::DMatrix<f64>,
                               initial_state: &na::DVector<f64>) -> Vec<f64> {
    rollout_on_device(model, input, initial_state).0.row(0).iter().copied().collect()
}
