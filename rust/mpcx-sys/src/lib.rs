//! Raw declarations of `include/mpcx.h`, one per entry point.  See that header for the contract of each call
//! (layouts, which pointers may be null, per-instance `info`).  Nothing here computes: the library is the
//! hand-written HIP implementation and `mpcx_create` fails with `MPCX_ERR_NO_DEVICE` without a GPU.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const MPCX_OK: c_int = 0;
pub const MPCX_ERR_NULL: c_int = -1;
pub const MPCX_ERR_DIMS: c_int = -2;
pub const MPCX_ERR_LAYOUT: c_int = -3;
pub const MPCX_ERR_UNSUPPORTED: c_int = -4;
pub const MPCX_ERR_NO_DEVICE: c_int = -5;
pub const MPCX_ERR_SINGULAR: c_int = -6;
pub const MPCX_ERR_COMM: c_int = -7;
pub const MPCX_COMM_ID_BYTES: usize = 128;

pub const MPCX_INTERLEAVE_FULL: i64 = 0;
pub const MPCX_INTERLEAVE_AOS: i64 = 1;
pub const MPCX_WEIGHTS_BLOCK: i32 = 0;
pub const MPCX_WEIGHTS_DENSE: i32 = 1;
pub const MPCX_HESSIAN_AUTO: i32 = 0;
pub const MPCX_HESSIAN_CONTRACT: i32 = 1;
pub const MPCX_HESSIAN_STRUCTURED: i32 = 2;
pub const MPCX_PLANT_DC_MOTOR: c_int = 0;
pub const MPCX_PLANT_TWO_MASS: c_int = 1;
pub const MPCX_OPT_RELAY_CAP_BYTES: c_int = 1;
pub const MPCX_OPT_QP_STREAMING: c_int = 2;

#[repr(C)]
pub struct mpcx_ctx {
    _private: [u8; 0],
}

#[repr(C)]
pub struct mpcx_comm {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct mpcx_dims {
    pub n: i32,
    pub m: i32,
    pub r: i32,
    pub f: i32,
    pub v: i32,
    pub weights_mode: i32,
    pub hessian_algo: i32,
    pub u_shared: i32,
    pub batch: i64,
    pub t: i64,
    pub interleave: i64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct mpcx_stats {
    pub count: i64,
    pub non_finite: i64,
    pub sum: f64,
    pub sum_sq: f64,
    pub max_abs: f64,
    pub checksum: u64,
}

extern "C" {
    pub fn mpcx_version() -> c_int;
    pub fn mpcx_array_elems(batch: i64, interleave: i64, elems_per_instance: i64) -> i64;
    pub fn mpcx_create(out: *mut *mut mpcx_ctx, device_id: c_int) -> c_int;
    pub fn mpcx_destroy(ctx: *mut mpcx_ctx);
    pub fn mpcx_last_error(ctx: *const mpcx_ctx) -> *const c_char;
    pub fn mpcx_reserve(ctx: *mut mpcx_ctx, dims: *const mpcx_dims) -> c_int;
    pub fn mpcx_set_option(ctx: *mut mpcx_ctx, option: c_int, value: i64) -> c_int;

    pub fn mpcx_qp_build_f64(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, a: *const f64, b: *const f64, c: *const f64,
        w3: *const f64, w4: *const f64, x0: *const f64, reference: *const f64, phi: *mut f64, gamma: *mut f64,
        h: *mut f64, fvec: *mut f64) -> c_int;
    pub fn mpcx_qp_build_f64_dev(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, a: *const f64, b: *const f64, c: *const f64,
        w3: *const f64, w4: *const f64, x0: *const f64, reference: *const f64, phi: *mut f64, gamma: *mut f64,
        h: *mut f64, fvec: *mut f64, stream: *mut c_void) -> c_int;

    pub fn mpcx_rollout_f64(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, a: *const f64, b: *const f64, c: *const f64,
        x0: *const f64, u: *const f64, x: *mut f64, y: *mut f64) -> c_int;
    pub fn mpcx_rollout_f64_dev(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, a: *const f64, b: *const f64, c: *const f64,
        x0: *const f64, u: *const f64, x: *mut f64, y: *mut f64, stream: *mut c_void) -> c_int;

    pub fn mpcx_qp_solve_f64(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, h: *const f64, fvec: *const f64,
        u_seq: *mut f64, info: *mut i32) -> c_int;
    pub fn mpcx_qp_solve_f64_dev(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, h: *const f64, fvec: *const f64,
        u_seq: *mut f64, info: *mut i32, stream: *mut c_void) -> c_int;
    pub fn mpcx_qp_gain_f64(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, h: *const f64, gamma: *const f64,
        w4: *const f64, gain: *mut f64, info: *mut i32) -> c_int;
    pub fn mpcx_qp_gain_f64_dev(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, h: *const f64, gamma: *const f64,
        w4: *const f64, gain: *mut f64, info: *mut i32, stream: *mut c_void) -> c_int;
    pub fn mpcx_qp_gain_rows_f64(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, h: *const f64, gamma: *const f64,
        w4: *const f64, gain_rows: *mut f64, info: *mut i32) -> c_int;
    pub fn mpcx_qp_gain_rows_f64_dev(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, h: *const f64, gamma: *const f64,
        w4: *const f64, gain_rows: *mut f64, info: *mut i32, stream: *mut c_void) -> c_int;

    pub fn mpcx_closed_loop_f64(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, a: *const f64, b: *const f64,
        c: *const f64, x0: *const f64, phi: *const f64, gain: *const f64, reference: *const f64, x: *mut f64,
        u: *mut f64, y: *mut f64) -> c_int;
    pub fn mpcx_closed_loop_f64_dev(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, a: *const f64, b: *const f64,
        c: *const f64, x0: *const f64, phi: *const f64, gain: *const f64, reference: *const f64, x: *mut f64,
        u: *mut f64, y: *mut f64, stream: *mut c_void) -> c_int;

    pub fn mpcx_c2d_backward_euler_f64(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, ac: *const f64, bc: *const f64,
        dt: f64, a: *mut f64, b: *mut f64, info: *mut i32) -> c_int;
    pub fn mpcx_c2d_backward_euler_f64_dev(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, ac: *const f64,
        bc: *const f64, dt: f64, a: *mut f64, b: *mut f64, info: *mut i32, stream: *mut c_void) -> c_int;
    /// Same as `mpcx_c2d_backward_euler_f64_dev` without any device allocation (static device scratch, bounded batch).
    pub fn mpcx_c2d_backward_euler_f64_noalloc_dev(ctx: *mut mpcx_ctx, dims: *const mpcx_dims, ac: *const f64,
        bc: *const f64, dt: f64, a: *mut f64, b: *mut f64, info: *mut i32, stream: *mut c_void) -> c_int;
    pub fn mpcx_plants_f64(ctx: *mut mpcx_ctx, kind: c_int, batch: i64, interleave: i64, params: *const f64, dt: f64,
        a: *mut f64, b: *mut f64, c: *mut f64, info: *mut i32) -> c_int;
    pub fn mpcx_plants_f64_dev(ctx: *mut mpcx_ctx, kind: c_int, batch: i64, interleave: i64, params: *const f64,
        dt: f64, a: *mut f64, b: *mut f64, c: *mut f64, info: *mut i32, stream: *mut c_void) -> c_int;

    pub fn mpcx_stats_f64_dev(ctx: *mut mpcx_ctx, buf: *const f64, count: i64, out_dev: *mut mpcx_stats,
        stream: *mut c_void) -> c_int;
    pub fn mpcx_stats_f64(ctx: *mut mpcx_ctx, buf_dev: *const f64, count: i64, out_host: *mut mpcx_stats) -> c_int;
    pub fn mpcx_comm_get_id(id_out: *mut c_void) -> c_int;
    pub fn mpcx_comm_create(out: *mut *mut mpcx_comm, ctx: *mut mpcx_ctx, id: *const c_void, world: c_int, rank: c_int) -> c_int;
    pub fn mpcx_comm_destroy(comm: *mut mpcx_comm);
    pub fn mpcx_comm_last_error(comm: *const mpcx_comm) -> *const c_char;
    pub fn mpcx_comm_rank(comm: *const mpcx_comm) -> c_int;
    pub fn mpcx_comm_world(comm: *const mpcx_comm) -> c_int;
    pub fn mpcx_gather_stats(comm: *mut mpcx_comm, buf_dev: *const f64, count: i64, stream: *mut c_void,
        out_host: *mut mpcx_stats) -> c_int;
    pub fn mpcx_gather_f64_dev(comm: *mut mpcx_comm, src_dev: *const f64, count: i64, dst_dev: *mut f64, root: c_int,
        stream: *mut c_void) -> c_int;
    pub fn mpcx_comm_synchronize(comm: *mut mpcx_comm) -> c_int;
    pub fn mpcx_relayout_f64_dev(ctx: *mut mpcx_ctx, batch: i64, elems_per_instance: i64, src_interleave: i64,
        src: *const f64, dst_interleave: i64, dst: *mut f64, stream: *mut c_void) -> c_int;
}
