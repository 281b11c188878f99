//! src/control/controller.rs of the reference.  Signatures are the reference's; the bodies are device calls.
pub mod mpc {
    extern crate nalgebra as na;

    use crate::control::device::{dims, with_ctx};
    use crate::control::model::DiscreteStateSpaceModel;
    use mpcx_sys as sys;
    use std::error::Error;
    use std::rc::Rc;

    pub struct Controller {
        model: Rc<dyn DiscreteStateSpaceModel>,
        f: usize,
        v: usize,
        mat_o: na::DMatrix<f64>,
        gain_matrix: na::DMatrix<f64>,
        pub desired_ctrl_traj_total: na::DMatrix<f64>,
    }

    /// W4 as `Controller::new` builds it is block diagonal with one repeated r x r block (controller.rs:179-185):
    /// true when `w4` has that form, in which case the library takes the block alone.
    fn is_repeated_block(w4: &na::DMatrix<f64>, f: usize, r: usize) -> bool {
        for j in 0..f * r {
            for i in 0..f * r {
                let want = if i / r == j / r { w4[(i % r, j % r)] } else { 0.0 };
                if !(w4[(i, j)] == want) {
                    return false;
                }
            }
        }
        true
    }

    /// Phi, Gamma and H = M^T W4 M + W3 (controller.rs:28-93) through `mpcx_qp_build_f64`.
    fn lifted_and_hessian(mat_a: &na::DMatrix<f64>, mat_b: &na::DMatrix<f64>, mat_c: &na::DMatrix<f64>, f: usize, v: usize,
                          mat_w3: &na::DMatrix<f64>, mat_w4: &na::DMatrix<f64>)
                          -> (na::DMatrix<f64>, na::DMatrix<f64>, na::DMatrix<f64>, sys::mpcx_dims) {
        let (n, r, m) = (mat_a.nrows(), mat_c.nrows(), mat_b.ncols()); // controller.rs:28-30
        // the shape assertions nalgebra raises at :38, :43, :61, :92, :93
        assert!(mat_a.ncols() == n && mat_b.nrows() == n && mat_c.ncols() == n, "Matrix multiplication dimensions mismatch.");
        assert!(mat_w4.nrows() == f * r && mat_w4.ncols() == f * r, "Matrix multiplication dimensions mismatch.");
        assert!(mat_w3.nrows() == v * m && mat_w3.ncols() == v * m, "Matrix addition/subtraction dimensions mismatch.");
        let mut d = dims(n, m, r, f, v, 1, 0);
        let block = is_repeated_block(mat_w4, f, r);
        d.weights_mode = if block { sys::MPCX_WEIGHTS_BLOCK } else { sys::MPCX_WEIGHTS_DENSE };
        let q = mat_w4.view((0, 0), (r, r)).into_owned();
        let w4 = if block { q.as_ptr() } else { mat_w4.as_ptr() };
        let mut mat_o = na::DMatrix::<f64>::zeros(f * r, n);
        let mut mat_m = na::DMatrix::<f64>::zeros(f * r, v * m);
        let mut h = na::DMatrix::<f64>::zeros(v * m, v * m);
        with_ctx("mpcx_qp_build_f64", |ctx| unsafe {
            sys::mpcx_qp_build_f64(ctx, &d, mat_a.as_ptr(), mat_b.as_ptr(), mat_c.as_ptr(), mat_w3.as_ptr(), w4,
                                   std::ptr::null(), std::ptr::null(), mat_o.as_mut_ptr(), mat_m.as_mut_ptr(),
                                   h.as_mut_ptr(), std::ptr::null_mut())
        });
        (mat_o, mat_m, h, d)
    }

    impl Controller {
        /// controller.rs:16-99: (mat_o, mat_m, gain_matrix).  The inversion and H^-1 M^T W4 are `mpcx_qp_gain_f64`
        /// (LU with partial pivoting, status per instance); a failed inversion is the reference's `Err`.
        pub fn form_lifted_matrices(mat_a: &na::DMatrix<f64>, mat_b: &na::DMatrix<f64>, mat_c: &na::DMatrix<f64>, f: usize,
                                    v: usize, mat_w3: &na::DMatrix<f64>, mat_w4: &na::DMatrix<f64>)
                                    -> Result<(na::DMatrix<f64>, na::DMatrix<f64>, na::DMatrix<f64>), Box<dyn Error>> {
            let (mat_o, mat_m, h, d) = lifted_and_hessian(mat_a, mat_b, mat_c, f, v, mat_w3, mat_w4);
            let (r, m) = (mat_c.nrows(), mat_b.ncols());
            let q = mat_w4.view((0, 0), (r, r)).into_owned();
            let w4 = if d.weights_mode == sys::MPCX_WEIGHTS_BLOCK { q.as_ptr() } else { mat_w4.as_ptr() };
            let mut gain_matrix = na::DMatrix::<f64>::zeros(v * m, f * r);
            let mut info: i32 = -1;
            with_ctx("mpcx_qp_gain_f64", |ctx| unsafe {
                sys::mpcx_qp_gain_f64(ctx, &d, h.as_ptr(), mat_m.as_ptr(), w4, gain_matrix.as_mut_ptr(), &mut info)
            });
            if info != 0 {
                return Err("Matrix inversion failed".into()); // controller.rs:93-95
            }
            Ok((mat_o, mat_m, gain_matrix))
        }

        /// Addition: (mat_o, rows) with rows = the first m rows of gain_matrix (m x f r), all that the control move uses
        /// (controller.rs:130-132): m solves with H^T instead of f r solves (`mpcx_qp_gain_rows_f64`).
        pub fn form_first_move_rows(mat_a: &na::DMatrix<f64>, mat_b: &na::DMatrix<f64>, mat_c: &na::DMatrix<f64>, f: usize,
                                    v: usize, mat_w3: &na::DMatrix<f64>, mat_w4: &na::DMatrix<f64>)
                                    -> Result<(na::DMatrix<f64>, na::DMatrix<f64>), Box<dyn Error>> {
            let (mat_o, mat_m, h, d) = lifted_and_hessian(mat_a, mat_b, mat_c, f, v, mat_w3, mat_w4);
            let (r, m) = (mat_c.nrows(), mat_b.ncols());
            let q = mat_w4.view((0, 0), (r, r)).into_owned();
            let w4 = if d.weights_mode == sys::MPCX_WEIGHTS_BLOCK { q.as_ptr() } else { mat_w4.as_ptr() };
            let mut rows = na::DMatrix::<f64>::zeros(m, f * r);
            let mut info: i32 = -1;
            with_ctx("mpcx_qp_gain_rows_f64", |ctx| unsafe {
                sys::mpcx_qp_gain_rows_f64(ctx, &d, h.as_ptr(), mat_m.as_ptr(), w4, rows.as_mut_ptr(), &mut info)
            });
            if info != 0 {
                return Err("Matrix inversion failed".into()); // controller.rs:93-95
            }
            Ok((mat_o, rows))
        }

        /// controller.rs:101-115: (x_kp1, y_k) with y_k = C x_k.  One rollout step on the device.
        pub fn propagate_dynamics(&self, control_input: &na::DVector<f64>, state: &na::DVector<f64>)
                                  -> (na::DVector<f64>, na::DVector<f64>) {
            let (a, b, c) = (self.model.get_mat_a(), self.model.get_mat_b(), self.model.get_mat_c());
            let (n, m, r) = (a.nrows(), b.ncols(), c.nrows());
            assert!(state.nrows() == n && control_input.nrows() == m, "Matrix multiplication dimensions mismatch.");
            let mut x = na::DMatrix::<f64>::zeros(n, 2);
            let mut y_k = na::DVector::<f64>::zeros(r);
            let d = dims(n, m, r, 1, 1, 1, 1);
            with_ctx("mpcx_rollout_f64", |ctx| unsafe {
                sys::mpcx_rollout_f64(ctx, &d, a.as_ptr(), b.as_ptr(), c.as_ptr(), state.as_ptr(), control_input.as_ptr(),
                                      x.as_mut_ptr(), y_k.as_mut_ptr())
            });
            (x.column(1).into_owned(), y_k)
        }

        /// gain (window - O x)[0 .. m) on the device: one step of the receding-horizon loop started at `state`
        /// (`mpcx_closed_loop_f64` with T = 1 keeps the reference's order: s first, then gain s).
        fn first_move(&self, timestep: usize, state: &na::DVector<f64>) -> na::DVector<f64> {
            let (a, b, c) = (self.model.get_mat_a(), self.model.get_mat_b(), self.model.get_mat_c());
            let (n, m, r) = (a.nrows(), b.ncols(), c.nrows());
            let traj = &self.desired_ctrl_traj_total;
            assert!(timestep + self.f <= traj.nrows(), "Matrix slicing out of bounds."); // view_range, :121
            assert!(traj.ncols() == r, "Matrix addition/subtraction dimensions mismatch."); // :125
            assert!(state.nrows() == n, "Matrix multiplication dimensions mismatch."); // :126
            // the device takes r x (T + f) with time along the columns; T = 1, the last column is never read
            let mut reference = na::DMatrix::<f64>::zeros(r, self.f + 1);
            for i in 0..self.f {
                for q in 0..r {
                    reference[(q, i)] = traj[(timestep + i, q)];
                }
            }
            let mut u = na::DVector::<f64>::zeros(m);
            let mut d = dims(n, m, r, self.f, self.v, 1, 1);
            d.u_shared = 1;
            with_ctx("mpcx_closed_loop_f64", |ctx| unsafe {
                sys::mpcx_closed_loop_f64(ctx, &d, a.as_ptr(), b.as_ptr(), c.as_ptr(), state.as_ptr(), self.mat_o.as_ptr(),
                                          self.gain_matrix.as_ptr(), reference.as_ptr(), std::ptr::null_mut(),
                                          u.as_mut_ptr(), std::ptr::null_mut())
            });
            u
        }

        /// controller.rs:117-135: the first element of gain (traj[t .. t+f] - O x) as a 1-vector (one input, one
        /// output, as the reference: it takes element (0, 0) and slices a T x 1 trajectory).
        pub fn compute_control_input(&mut self, timestep: usize, state: &na::DVector<f64>) -> na::DVector<f64> {
            assert!(self.desired_ctrl_traj_total.ncols() == 1 && self.mat_o.nrows() == self.f,
                    "Matrix addition/subtraction dimensions mismatch.");
            let u = self.first_move(timestep, state);
            na::DVector::<f64>::from_element(1, u[0])
        }

        /// Addition: the move for m inputs and r outputs -- the first m entries of gain (window - O x), the window
        /// being rows timestep .. timestep + f of a (time_steps x r) trajectory.
        pub fn compute_control_input_mimo(&mut self, timestep: usize, state: &na::DVector<f64>) -> na::DVector<f64> {
            self.first_move(timestep, state)
        }

        /// Addition: the loop of src/main.rs:51-71 in one device call: (states n x (steps+1), inputs m x steps,
        /// outputs r x steps) with outputs[:, t] = C x_t.
        pub fn closed_loop(&self, steps: usize, x0: &na::DVector<f64>) -> (na::DMatrix<f64>, na::DMatrix<f64>, na::DMatrix<f64>) {
            let (a, b, c) = (self.model.get_mat_a(), self.model.get_mat_b(), self.model.get_mat_c());
            let (n, m, r) = (a.nrows(), b.ncols(), c.nrows());
            let traj = &self.desired_ctrl_traj_total;
            assert!(x0.nrows() == n, "Matrix multiplication dimensions mismatch.");
            assert!(steps + self.f <= traj.nrows() && traj.ncols() == r, "Matrix slicing out of bounds.");
            let reference = traj.rows(0, steps + self.f).transpose(); // r x (T + f)
            let mut states = na::DMatrix::<f64>::zeros(n, steps + 1);
            let mut inputs = na::DMatrix::<f64>::zeros(m, steps);
            let mut outputs = na::DMatrix::<f64>::zeros(r, steps);
            let mut d = dims(n, m, r, self.f, self.v, 1, steps as i64);
            d.u_shared = 1;
            with_ctx("mpcx_closed_loop_f64", |ctx| unsafe {
                sys::mpcx_closed_loop_f64(ctx, &d, a.as_ptr(), b.as_ptr(), c.as_ptr(), x0.as_ptr(), self.mat_o.as_ptr(),
                                          self.gain_matrix.as_ptr(), reference.as_ptr(), states.as_mut_ptr(),
                                          inputs.as_mut_ptr(), outputs.as_mut_ptr())
            });
            (states, inputs, outputs)
        }

        /// controller.rs:137-207.  The weights are built on the host as the reference builds them -- every element of
        /// a weight block carries the weight (:164-165, :184) -- and are shared by every plant of a batch.
        pub fn new(model: Rc<dyn DiscreteStateSpaceModel>, f: usize, v: usize, q0: f64, q_other: f64, pred_weight: f64,
                   desired_ctrl_traj: &na::DMatrix<f64>) -> Result<Controller, Box<dyn Error>> {
            let (m, r) = (model.get_mat_b().ncols(), model.get_mat_c().nrows());
            let vm = v * m;
            let mut mat_w1 = na::DMatrix::<f64>::zeros(vm, vm); // :149-161
            let mut mat_w2 = na::DMatrix::<f64>::zeros(vm, vm); // :164-175
            for i in 0..v {
                mat_w1.view_mut((i * m, i * m), (m, m)).fill_with_identity();
                if i >= 1 {
                    mat_w1.view_mut((i * m, (i - 1) * m), (m, m)).fill_diagonal(-1.0);
                }
                mat_w2.view_mut((i * m, i * m), (m, m)).fill(if i == 0 { q0 } else { q_other });
            }
            let mat_w3 = mat_w1.transpose() * (&mat_w2 * &mat_w1); // :177, host side: shared by the batch
            let mut mat_w4 = na::DMatrix::<f64>::zeros(f * r, f * r); // :179-185
            for i in 0..f {
                mat_w4.view_mut((i * r, i * r), (r, r)).fill(pred_weight);
            }
            let (mat_o, _mat_m, gain_matrix) = Controller::form_lifted_matrices(
                model.get_mat_a(), model.get_mat_b(), model.get_mat_c(), f, v, &mat_w3, &mat_w4)?;
            Ok(Controller { model, f, v, mat_o, gain_matrix, desired_ctrl_traj_total: desired_ctrl_traj.clone() })
        }
    }
}
