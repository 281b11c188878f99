//! src/control/model.rs of the reference: the plant trait and the two parametrised plants.  `Model::new` runs on the
//! device (`mpcx_plants_f64`: continuous matrices from the parameters, then backward Euler, model.rs:45-66, :126-149).
extern crate nalgebra as na;

use super::device::{dims, with_ctx};
use mpcx_sys as sys;

pub trait DiscreteStateSpaceModel {
    fn get_mat_a(&self) -> &na::DMatrix<f64>;
    fn get_mat_b(&self) -> &na::DMatrix<f64>;
    fn get_mat_c(&self) -> &na::DMatrix<f64>;
    fn get_sampling_dt(&self) -> f64;
}

fn plant(kind: i32, n: usize, params: &[f64], sampling_dt: f64) -> (na::DMatrix<f64>, na::DMatrix<f64>, na::DMatrix<f64>) {
    let (mut a, mut b, mut c) = (na::DMatrix::<f64>::zeros(n, n), na::DMatrix::<f64>::zeros(n, 1), na::DMatrix::<f64>::zeros(1, n));
    let mut info: i32 = -1;
    with_ctx("mpcx_plants_f64", |ctx| unsafe {
        sys::mpcx_plants_f64(ctx, kind, 1, sys::MPCX_INTERLEAVE_AOS, params.as_ptr(), sampling_dt, a.as_mut_ptr(),
                             b.as_mut_ptr(), c.as_mut_ptr(), &mut info)
    });
    // model.rs:56, :139 unwrap the failed inversion of I - Ac dt: keep the panic
    assert!(info == 0, "called `Option::unwrap()` on a `None` value");
    (a, b, c)
}

/// Backward Euler on the device for any continuous plant: A = (I - Ac dt)^-1, B = A (Bc dt) (model.rs:55-57).
/// An addition: the reference only discretises its two own plants.
pub fn c2d_backward_euler(mat_ac: &na::DMatrix<f64>, mat_bc: &na::DMatrix<f64>, dt: f64) -> (na::DMatrix<f64>, na::DMatrix<f64>) {
    let (n, m) = (mat_ac.nrows(), mat_bc.ncols());
    assert!(mat_ac.ncols() == n, "Unable to invert a non-square matrix.");
    assert!(mat_bc.nrows() == n, "Matrix multiplication dimensions mismatch.");
    let (mut a, mut b) = (na::DMatrix::<f64>::zeros(n, n), na::DMatrix::<f64>::zeros(n, m));
    let mut info: i32 = -1;
    let d = dims(n, m, 1, 1, 1, 1, 0);
    with_ctx("mpcx_c2d_backward_euler_f64", |ctx| unsafe {
        sys::mpcx_c2d_backward_euler_f64(ctx, &d, mat_ac.as_ptr(), mat_bc.as_ptr(), dt, a.as_mut_ptr(), b.as_mut_ptr(), &mut info)
    });
    assert!(info == 0, "called `Option::unwrap()` on a `None` value");
    (a, b)
}

macro_rules! plant_model {
    () => {
        #[derive(Clone)]
        pub struct Model {
            mat_a: na::DMatrix<f64>,
            mat_b: na::DMatrix<f64>,
            mat_c: na::DMatrix<f64>,
            sampling_dt: f64,
        }

        impl DiscreteStateSpaceModel for Model {
            fn get_mat_a(&self) -> &na::DMatrix<f64> {
                &self.mat_a
            }
            fn get_mat_b(&self) -> &na::DMatrix<f64> {
                &self.mat_b
            }
            fn get_mat_c(&self) -> &na::DMatrix<f64> {
                &self.mat_c
            }
            fn get_sampling_dt(&self) -> f64 {
                self.sampling_dt
            }
        }
    };
}

pub mod dc_motor {
    extern crate nalgebra as na;
    use super::{plant, DiscreteStateSpaceModel};
    use mpcx_sys as sys;

    /// model.rs:16-34.  The fields stay private as in the reference; `with` is an addition (parameter sweeps).
    pub struct Parameters {
        b: f64,
        j: f64,
        k: f64,
        l: f64,
        r: f64,
    }

    impl Default for Parameters {
        fn default() -> Parameters {
            Parameters { b: 0.1, j: 0.01, k: 0.01, l: 0.5, r: 1.0 }
        }
    }

    impl Parameters {
        pub fn with(b: f64, j: f64, k: f64, l: f64, r: f64) -> Parameters {
            Parameters { b, j, k, l, r }
        }
    }

    plant_model!();

    impl Model {
        pub fn new(params: Parameters, sampling_dt: f64) -> Model {
            let (mat_a, mat_b, mat_c) =
                plant(sys::MPCX_PLANT_DC_MOTOR, 2, &[params.b, params.j, params.k, params.l, params.r], sampling_dt);
            Model { mat_a, mat_b, mat_c, sampling_dt }
        }
    }
}

pub mod mpc {
    extern crate nalgebra as na;
    use super::{plant, DiscreteStateSpaceModel};
    use mpcx_sys as sys;

    /// model.rs:95-115
    pub struct Parameters {
        m1: f64,
        m2: f64,
        k1: f64,
        k2: f64,
        d1: f64,
        d2: f64,
    }

    impl Default for Parameters {
        fn default() -> Parameters {
            Parameters { m1: 2.0, m2: 2.0, k1: 100.0, k2: 200.0, d1: 1.0, d2: 5.0 }
        }
    }

    impl Parameters {
        pub fn with(m1: f64, m2: f64, k1: f64, k2: f64, d1: f64, d2: f64) -> Parameters {
            Parameters { m1, m2, k1, k2, d1, d2 }
        }
    }

    plant_model!();

    impl Model {
        pub fn new(params: Parameters, sampling_dt: f64) -> Model {
            let p = [params.m1, params.m2, params.k1, params.k2, params.d1, params.d2];
            let (mat_a, mat_b, mat_c) = plant(sys::MPCX_PLANT_TWO_MASS, 4, &p, sampling_dt);
            Model { mat_a, mat_b, mat_c, sampling_dt }
        }
    }
}
