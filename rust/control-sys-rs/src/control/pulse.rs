//! src/control/trajectory.rs of the reference: a one-shot host helper, not on the device path.
extern crate nalgebra as na;

/// trajectory.rs:3-11: ones on [0, T/3) and [2T/3, T), zeros between; a T x 1 matrix.
pub fn generate_pulse_trajectory(time_steps: usize) -> na::DMatrix<f64> {
    let mut traj = na::DMatrix::<f64>::zeros(time_steps, 1);
    let (third, start) = (time_steps / 3, 2 * time_steps / 3);
    // the reference copies a (T/3)-long block of ones into rows 2T/3..T and panics when the lengths differ
    assert!(time_steps - start == third, "Unable to copy from a matrix with incompatible shape.");
    for i in (0..third).chain(start..time_steps) {
        traj[(i, 0)] = 1.0;
    }
    traj
}
