//! Raw bindings to libspal_hip.so (include/spal.h).
//!
//! NOT COMPILED IN THIS REPOSITORY'S PIPELINE: the build image has no
//! rustc/cargo.  This is the source a spalinalg maintainer adds
//! to the crate as `src/hip/ffi.rs`; see INTEGRATION.md.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct spal_csr { _private: [u8; 0] }
#[repr(C)] pub struct spal_csc { _private: [u8; 0] }

pub const SPAL_OK: c_int = 0;

#[link(name = "spal_hip")]
extern "C" {
    pub fn spal_last_error() -> *const c_char;

    pub fn spal_csr_create_f64(device: c_int, nrows: u64, ncols: u64,
        rowptr: *const u64, rowptr_len: u64, colind: *const u64, colind_len: u64,
        values: *const f64, values_len: u64, out: *mut *mut spal_csr) -> c_int;
    pub fn spal_csr_create_f32(device: c_int, nrows: u64, ncols: u64,
        rowptr: *const u64, rowptr_len: u64, colind: *const u64, colind_len: u64,
        values: *const f32, values_len: u64, out: *mut *mut spal_csr) -> c_int;
    pub fn spal_csr_destroy(a: *mut spal_csr) -> c_int;
    pub fn spal_csr_shape(a: *mut spal_csr, nrows: *mut u64, ncols: *mut u64,
        nnz: *mut u64, elem_size: *mut c_int) -> c_int;
    pub fn spal_csr_spmv_f64(a: *mut spal_csr, x: *const f64, x_len: u64, y: *mut f64, y_len: u64) -> c_int;
    pub fn spal_csr_spmv_f32(a: *mut spal_csr, x: *const f32, x_len: u64, y: *mut f32, y_len: u64) -> c_int;
    pub fn spal_csr_spmv_dev_f64(a: *mut spal_csr, x_dev: *const f64, y_dev: *mut f64, stream: *mut c_void) -> c_int;
    pub fn spal_csr_download_f64(a: *mut spal_csr, rowptr: *mut u64, colind: *mut u64, values: *mut f64) -> c_int;
    pub fn spal_csr_download_f32(a: *mut spal_csr, rowptr: *mut u64, colind: *mut u64, values: *mut f32) -> c_int;

    pub fn spal_csc_create_f64(device: c_int, nrows: u64, ncols: u64,
        colptr: *const u64, colptr_len: u64, rowind: *const u64, rowind_len: u64,
        values: *const f64, values_len: u64, out: *mut *mut spal_csc) -> c_int;
    pub fn spal_csc_create_f32(device: c_int, nrows: u64, ncols: u64,
        colptr: *const u64, colptr_len: u64, rowind: *const u64, rowind_len: u64,
        values: *const f32, values_len: u64, out: *mut *mut spal_csc) -> c_int;
    pub fn spal_csc_destroy(a: *mut spal_csc) -> c_int;
    pub fn spal_csc_spmv_f64(a: *mut spal_csc, x: *const f64, x_len: u64, y: *mut f64, y_len: u64) -> c_int;
    pub fn spal_csc_spmv_f32(a: *mut spal_csc, x: *const f32, x_len: u64, y: *mut f32, y_len: u64) -> c_int;

    pub fn spal_coo_to_csr_f64(device: c_int, nrows: u64, ncols: u64, len: u64,
        rows: *const u64, cols: *const u64, vals: *const f64, out: *mut *mut spal_csr) -> c_int;
    pub fn spal_coo_to_csr_f32(device: c_int, nrows: u64, ncols: u64, len: u64,
        rows: *const u64, cols: *const u64, vals: *const f32, out: *mut *mut spal_csr) -> c_int;
}

/// The reference panics on contract violations (`assert!`, src/csr.rs:144-156);
/// every non-zero status keeps that convention.
pub fn check(status: c_int) {
    if status != SPAL_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(spal_last_error()) }.to_string_lossy().into_owned();
        panic!("spal_hip status {}: {}", status, msg);
    }
}
