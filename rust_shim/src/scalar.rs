//! Per-`Scalar` dispatch to the two instantiations of the C ABI (`Scalar` is implemented for f32 and f64
//! only, src/scalar.rs:56-57).  The crate has no dependencies (Cargo.toml:13), so the entry points are passed
//! to the macro by name instead of being pasted together.
use std::os::raw::{c_int, c_void};

use super::ffi;
use crate::scalar::Scalar;

#[allow(clippy::too_many_arguments)]
pub trait HipScalar: Scalar {
    unsafe fn csr_create(d: c_int, nr: u64, nc: u64, rp: &[usize], ci: &[usize], v: &[Self], out: *mut *mut ffi::spal_csr) -> c_int;
    unsafe fn csr_spmv(a: *mut ffi::spal_csr, x: &[Self], y: &mut [Self]) -> c_int;
    unsafe fn csr_spmv_dev(a: *mut ffi::spal_csr, x: *const Self, y: *mut Self, stream: *mut c_void) -> c_int;
    unsafe fn csr_autotune(a: *mut ffi::spal_csr, x: *const Self, y: *mut Self, stream: *mut c_void, iters: c_int) -> c_int;
    unsafe fn csr_download(a: *mut ffi::spal_csr, rp: &mut [usize], ci: &mut [usize], v: &mut [Self]) -> c_int;
    unsafe fn csc_create(d: c_int, nr: u64, nc: u64, cp: &[usize], ri: &[usize], v: &[Self], out: *mut *mut ffi::spal_csc) -> c_int;
    unsafe fn csc_spmv(a: *mut ffi::spal_csc, x: &[Self], y: &mut [Self]) -> c_int;
    unsafe fn csc_spmv_dev(a: *mut ffi::spal_csc, x: *const Self, y: *mut Self, stream: *mut c_void) -> c_int;
    unsafe fn csc_download(a: *mut ffi::spal_csc, cp: &mut [usize], ri: &mut [usize], v: &mut [Self]) -> c_int;
    unsafe fn coo_upload(d: c_int, nr: u64, nc: u64, r: &[usize], c: &[usize], v: &[Self], out: *mut *mut ffi::spal_coo) -> c_int;
    unsafe fn mg_csr_create(ctx: *mut ffi::spal_mg, nr: u64, nc: u64, rp: &[usize], ci: &[usize], v: &[Self], out: *mut *mut ffi::spal_mg_csr) -> c_int;
    unsafe fn mg_csr_spmv(a: *mut ffi::spal_mg_csr, x: &[Self], y: &mut [Self]) -> c_int;
}

macro_rules! impl_hip_scalar {
    ($t:ty, $csr_create:ident, $csr_spmv:ident, $csr_spmv_dev:ident, $csr_autotune:ident, $csr_download:ident,
     $csc_create:ident, $csc_spmv:ident, $csc_spmv_dev:ident, $csc_download:ident, $coo_upload:ident,
     $mg_csr_create:ident, $mg_csr_spmv:ident) => {
        impl HipScalar for $t {
            unsafe fn csr_create(d: c_int, nr: u64, nc: u64, rp: &[usize], ci: &[usize], v: &[Self], out: *mut *mut ffi::spal_csr) -> c_int {
                ffi::$csr_create(d, nr, nc, rp.as_ptr() as *const u64, rp.len() as u64, ci.as_ptr() as *const u64,
                                 ci.len() as u64, v.as_ptr(), v.len() as u64, out)
            }
            unsafe fn csr_spmv(a: *mut ffi::spal_csr, x: &[Self], y: &mut [Self]) -> c_int {
                ffi::$csr_spmv(a, x.as_ptr(), x.len() as u64, y.as_mut_ptr(), y.len() as u64)
            }
            unsafe fn csr_spmv_dev(a: *mut ffi::spal_csr, x: *const Self, y: *mut Self, stream: *mut c_void) -> c_int {
                ffi::$csr_spmv_dev(a, x, y, stream)
            }
            unsafe fn csr_autotune(a: *mut ffi::spal_csr, x: *const Self, y: *mut Self, stream: *mut c_void, iters: c_int) -> c_int {
                ffi::$csr_autotune(a, x, y, stream, iters)
            }
            unsafe fn csr_download(a: *mut ffi::spal_csr, rp: &mut [usize], ci: &mut [usize], v: &mut [Self]) -> c_int {
                ffi::$csr_download(a, rp.as_mut_ptr() as *mut u64, ci.as_mut_ptr() as *mut u64, v.as_mut_ptr())
            }
            unsafe fn csc_create(d: c_int, nr: u64, nc: u64, cp: &[usize], ri: &[usize], v: &[Self], out: *mut *mut ffi::spal_csc) -> c_int {
                ffi::$csc_create(d, nr, nc, cp.as_ptr() as *const u64, cp.len() as u64, ri.as_ptr() as *const u64,
                                 ri.len() as u64, v.as_ptr(), v.len() as u64, out)
            }
            unsafe fn csc_spmv(a: *mut ffi::spal_csc, x: &[Self], y: &mut [Self]) -> c_int {
                ffi::$csc_spmv(a, x.as_ptr(), x.len() as u64, y.as_mut_ptr(), y.len() as u64)
            }
            unsafe fn csc_spmv_dev(a: *mut ffi::spal_csc, x: *const Self, y: *mut Self, stream: *mut c_void) -> c_int {
                ffi::$csc_spmv_dev(a, x, y, stream)
            }
            unsafe fn csc_download(a: *mut ffi::spal_csc, cp: &mut [usize], ri: &mut [usize], v: &mut [Self]) -> c_int {
                ffi::$csc_download(a, cp.as_mut_ptr() as *mut u64, ri.as_mut_ptr() as *mut u64, v.as_mut_ptr())
            }
            unsafe fn coo_upload(d: c_int, nr: u64, nc: u64, r: &[usize], c: &[usize], v: &[Self], out: *mut *mut ffi::spal_coo) -> c_int {
                ffi::$coo_upload(d, nr, nc, v.len() as u64, r.as_ptr() as *const u64, c.as_ptr() as *const u64, v.as_ptr(), out)
            }
            unsafe fn mg_csr_create(ctx: *mut ffi::spal_mg, nr: u64, nc: u64, rp: &[usize], ci: &[usize], v: &[Self], out: *mut *mut ffi::spal_mg_csr) -> c_int {
                ffi::$mg_csr_create(ctx, nr, nc, rp.as_ptr() as *const u64, rp.len() as u64, ci.as_ptr() as *const u64,
                                    ci.len() as u64, v.as_ptr(), v.len() as u64, out)
            }
            unsafe fn mg_csr_spmv(a: *mut ffi::spal_mg_csr, x: &[Self], y: &mut [Self]) -> c_int {
                ffi::$mg_csr_spmv(a, x.as_ptr(), x.len() as u64, y.as_mut_ptr(), y.len() as u64)
            }
        }
    };
}
impl_hip_scalar!(f64, spal_csr_create_f64, spal_csr_spmv_f64, spal_csr_spmv_dev_f64, spal_csr_autotune_f64,
                 spal_csr_download_f64, spal_csc_create_f64, spal_csc_spmv_f64, spal_csc_spmv_dev_f64,
                 spal_csc_download_f64, spal_coo_upload_f64, spal_mg_csr_create_f64, spal_mg_csr_spmv_f64);
impl_hip_scalar!(f32, spal_csr_create_f32, spal_csr_spmv_f32, spal_csr_spmv_dev_f32, spal_csr_autotune_f32,
                 spal_csr_download_f32, spal_csc_create_f32, spal_csc_spmv_f32, spal_csc_spmv_dev_f32,
                 spal_csc_download_f32, spal_coo_upload_f32, spal_mg_csr_create_f32, spal_mg_csr_spmv_f32);
