//! `control_sys_rs` with the reference's public surface (src/lib.rs:1, src/control.rs:1-4) and every computation of
//! the hot path -- plant discretisation, lifted matrices, Hessian, inversion and gain, control move, propagation,
//! rollout -- executed by libmpcx on the GPU through `mpcx-sys`.  There is no host arithmetic fallback: without a
//! usable device the first call panics with the library's error.
//!
//! Written against `include/mpcx.h`; this image has no Rust toolchain, so the crate has not been compiled.  The
//! same calls, in the same order, are what `mpc-rs_amd/host/control_sys_rs.hpp` (C++, compiled and tested on the
//! GPU) makes.
/// The reference's `control` module tree (src/control.rs:1-4) plus the additive `batch`.
pub mod control {
    pub mod batch;
    pub mod controller;
    mod device;
    pub mod model;
    #[path = "rollout_host.rs"]
    pub mod simulator;
    #[path = "pulse.rs"]
    pub mod trajectory;
}
