//! One `mpcx_ctx` per host thread, created on first use on device `MPCX_DEVICE` (default 0).  The reference's
//! `Controller` holds an `Rc` and is `!Send`; a context is likewise bound to its thread.
use mpcx_sys as sys;
use std::ffi::CStr;

pub(crate) struct Device {
    ctx: *mut sys::mpcx_ctx,
}

impl Device {
    fn open() -> Device {
        let id = std::env::var("MPCX_DEVICE").ok().and_then(|s| s.parse::<i32>().ok()).unwrap_or(0);
        let mut ctx: *mut sys::mpcx_ctx = std::ptr::null_mut();
        let rc = unsafe { sys::mpcx_create(&mut ctx, id) };
        if rc != sys::MPCX_OK {
            panic!("mpcx_create(device {id}) failed with code {rc}: this library computes on the GPU only");
        }
        Device { ctx }
    }
}

impl Drop for Device {
    fn drop(&mut self) {
        unsafe { sys::mpcx_destroy(self.ctx) }
    }
}

thread_local! {
    static DEVICE: Device = Device::open();
}

/// Runs `call` with this thread's context; a non-zero status becomes a panic carrying `mpcx_last_error`.
pub(crate) fn with_ctx<F: FnOnce(*mut sys::mpcx_ctx) -> i32>(what: &str, call: F) {
    DEVICE.with(|dev| {
        let rc = call(dev.ctx);
        if rc != sys::MPCX_OK {
            let msg = unsafe { CStr::from_ptr(sys::mpcx_last_error(dev.ctx)) }.to_string_lossy().into_owned();
            panic!("{what}: {msg} (code {rc})");
        }
    })
}

pub(crate) fn dims(n: usize, m: usize, r: usize, f: usize, v: usize, batch: i64, t: i64) -> sys::mpcx_dims {
    sys::mpcx_dims {
        n: n as i32, m: m as i32, r: r as i32, f: f as i32, v: v as i32,
        weights_mode: sys::MPCX_WEIGHTS_BLOCK, hessian_algo: sys::MPCX_HESSIAN_AUTO, u_shared: 0,
        batch, t, interleave: sys::MPCX_INTERLEAVE_AOS,
    }
}
