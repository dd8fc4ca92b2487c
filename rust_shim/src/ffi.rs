//! Raw bindings to libspal_hip.so -- GENERATED from include/spal.h by tools/gen_rust_ffi.py; do not edit.
//!
//! NOT COMPILED IN THIS REPOSITORY'S PIPELINE: the build image has no rustc/cargo.  This is
//! the source a spalinalg maintainer adds to the crate as `src/hip/ffi.rs`; see INTEGRATION.md.  Every
//! function of the C ABI is declared (tests/test_host_abi.py keeps this file in step with the header).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct spal_csr { _private: [u8; 0] }
#[repr(C)] pub struct spal_csc { _private: [u8; 0] }
#[repr(C)] pub struct spal_coo { _private: [u8; 0] }
#[repr(C)] pub struct spal_mg { _private: [u8; 0] }
#[repr(C)] pub struct spal_mg_csr { _private: [u8; 0] }

pub const SPAL_OK: c_int = 0;
pub const SPAL_ERR_INVALID_ARGUMENT: c_int = 1;
pub const SPAL_ERR_INVARIANT: c_int = 2;
pub const SPAL_ERR_HIP: c_int = 3;
pub const SPAL_ERR_OUT_OF_MEMORY: c_int = 4;
pub const SPAL_ERR_UNSUPPORTED: c_int = 5;
pub const SPAL_ERR_NO_DEVICE: c_int = 6;
pub const SPAL_ERR_INDEX_OUT_OF_BOUNDS: c_int = 7;

#[link(name = "spal_hip")]
extern "C" {
    pub fn spal_last_error() -> *const c_char;
    pub fn spal_version() -> *const c_char;
    pub fn spal_device_count(count: *mut c_int) -> c_int;
    pub fn spal_csr_validate(nrows: u64, ncols: u64, rowptr: *const u64, rowptr_len: u64, colind: *const u64, colind_len: u64, values_len: u64, reason: *mut c_int) -> c_int;
    pub fn spal_csc_validate(nrows: u64, ncols: u64, colptr: *const u64, colptr_len: u64, rowind: *const u64, rowind_len: u64, values_len: u64, reason: *mut c_int) -> c_int;
    pub fn spal_partition_rows(rowptr: *const u64, nrows: u64, nparts: u32, bounds: *mut u64) -> c_int;
    pub fn spal_csr_create_f64(device: c_int, nrows: u64, ncols: u64, rowptr: *const u64, rowptr_len: u64, colind: *const u64, colind_len: u64, values: *const f64, values_len: u64, out: *mut *mut spal_csr) -> c_int;
    pub fn spal_csr_create_f32(device: c_int, nrows: u64, ncols: u64, rowptr: *const u64, rowptr_len: u64, colind: *const u64, colind_len: u64, values: *const f32, values_len: u64, out: *mut *mut spal_csr) -> c_int;
    pub fn spal_csr_destroy(a: *mut spal_csr) -> c_int;
    pub fn spal_csr_shape(a: *mut spal_csr, nrows: *mut u64, ncols: *mut u64, nnz: *mut u64, elem_size: *mut c_int) -> c_int;
    pub fn spal_csr_spmv_f64(a: *mut spal_csr, x: *const f64, x_len: u64, y: *mut f64, y_len: u64) -> c_int;
    pub fn spal_csr_spmv_f32(a: *mut spal_csr, x: *const f32, x_len: u64, y: *mut f32, y_len: u64) -> c_int;
    pub fn spal_csr_spmv_dev_f64(a: *mut spal_csr, x_dev: *const f64, y_dev: *mut f64, stream: *mut c_void) -> c_int;
    pub fn spal_csr_spmv_dev_f32(a: *mut spal_csr, x_dev: *const f32, y_dev: *mut f32, stream: *mut c_void) -> c_int;
    pub fn spal_csr_download_f64(a: *mut spal_csr, rowptr: *mut u64, colind: *mut u64, values: *mut f64) -> c_int;
    pub fn spal_csr_download_f32(a: *mut spal_csr, rowptr: *mut u64, colind: *mut u64, values: *mut f32) -> c_int;
    pub fn spal_csr_set_option(a: *mut spal_csr, key: *const c_char, value: i64) -> c_int;
    pub fn spal_csr_autotune_f64(a: *mut spal_csr, x_dev: *const f64, y_dev: *mut f64, stream: *mut c_void, iters: c_int) -> c_int;
    pub fn spal_csr_autotune_f32(a: *mut spal_csr, x_dev: *const f32, y_dev: *mut f32, stream: *mut c_void, iters: c_int) -> c_int;
    pub fn spal_csr_alloc_vectors(a: *mut spal_csr, x_dev: *mut *mut c_void, y_dev: *mut *mut c_void, stream: *mut c_void) -> c_int;
    pub fn spal_csr_describe(a: *mut spal_csr, buf: *mut c_char, buf_len: usize) -> c_int;
    pub fn spal_csc_create_f64(device: c_int, nrows: u64, ncols: u64, colptr: *const u64, colptr_len: u64, rowind: *const u64, rowind_len: u64, values: *const f64, values_len: u64, out: *mut *mut spal_csc) -> c_int;
    pub fn spal_csc_create_f32(device: c_int, nrows: u64, ncols: u64, colptr: *const u64, colptr_len: u64, rowind: *const u64, rowind_len: u64, values: *const f32, values_len: u64, out: *mut *mut spal_csc) -> c_int;
    pub fn spal_csc_destroy(a: *mut spal_csc) -> c_int;
    pub fn spal_csc_shape(a: *mut spal_csc, nrows: *mut u64, ncols: *mut u64, nnz: *mut u64, elem_size: *mut c_int) -> c_int;
    pub fn spal_csc_spmv_f64(a: *mut spal_csc, x: *const f64, x_len: u64, y: *mut f64, y_len: u64) -> c_int;
    pub fn spal_csc_spmv_f32(a: *mut spal_csc, x: *const f32, x_len: u64, y: *mut f32, y_len: u64) -> c_int;
    pub fn spal_csc_spmv_dev_f64(a: *mut spal_csc, x_dev: *const f64, y_dev: *mut f64, stream: *mut c_void) -> c_int;
    pub fn spal_csc_spmv_dev_f32(a: *mut spal_csc, x_dev: *const f32, y_dev: *mut f32, stream: *mut c_void) -> c_int;
    pub fn spal_csc_download_f64(a: *mut spal_csc, colptr: *mut u64, rowind: *mut u64, values: *mut f64) -> c_int;
    pub fn spal_csc_download_f32(a: *mut spal_csc, colptr: *mut u64, rowind: *mut u64, values: *mut f32) -> c_int;
    pub fn spal_csc_set_option(a: *mut spal_csc, key: *const c_char, value: i64) -> c_int;
    pub fn spal_csc_autotune_f64(a: *mut spal_csc, x_dev: *const f64, y_dev: *mut f64, stream: *mut c_void, iters: c_int) -> c_int;
    pub fn spal_csc_autotune_f32(a: *mut spal_csc, x_dev: *const f32, y_dev: *mut f32, stream: *mut c_void, iters: c_int) -> c_int;
    pub fn spal_csc_describe(a: *mut spal_csc, buf: *mut c_char, buf_len: usize) -> c_int;
    pub fn spal_csc_status(a: *mut spal_csc, invalid_products: *mut c_int) -> c_int;
    pub fn spal_csc_to_csr(a: *mut spal_csc, out: *mut *mut spal_csr) -> c_int;
    pub fn spal_csr_to_csc(a: *mut spal_csr, out: *mut *mut spal_csc) -> c_int;
    pub fn spal_coo_upload_f64(device: c_int, nrows: u64, ncols: u64, len: u64, rows: *const u64, cols: *const u64, vals: *const f64, out: *mut *mut spal_coo) -> c_int;
    pub fn spal_coo_upload_f32(device: c_int, nrows: u64, ncols: u64, len: u64, rows: *const u64, cols: *const u64, vals: *const f32, out: *mut *mut spal_coo) -> c_int;
    pub fn spal_coo_destroy(c: *mut spal_coo) -> c_int;
    pub fn spal_coo_assemble_csr(c: *mut spal_coo, stream: *mut c_void, out: *mut *mut spal_csr) -> c_int;
    pub fn spal_csr_plan(a: *mut spal_csr) -> c_int;
    pub fn spal_coo_assemble_csc(c: *mut spal_coo, stream: *mut c_void, out: *mut *mut spal_csc) -> c_int;
    pub fn spal_coo_describe(c: *mut spal_coo, buf: *mut c_char, buf_len: usize) -> c_int;
    pub fn spal_coo_to_csr_f64(device: c_int, nrows: u64, ncols: u64, len: u64, rows: *const u64, cols: *const u64, vals: *const f64, out: *mut *mut spal_csr) -> c_int;
    pub fn spal_coo_to_csr_f32(device: c_int, nrows: u64, ncols: u64, len: u64, rows: *const u64, cols: *const u64, vals: *const f32, out: *mut *mut spal_csr) -> c_int;
    pub fn spal_coo_to_csc_f64(device: c_int, nrows: u64, ncols: u64, len: u64, rows: *const u64, cols: *const u64, vals: *const f64, out: *mut *mut spal_csc) -> c_int;
    pub fn spal_coo_to_csc_f32(device: c_int, nrows: u64, ncols: u64, len: u64, rows: *const u64, cols: *const u64, vals: *const f32, out: *mut *mut spal_csc) -> c_int;
    pub fn spal_mg_create(ngpus: c_int, devices: *const c_int, out: *mut *mut spal_mg) -> c_int;
    pub fn spal_mg_create_transport(ngpus: c_int, devices: *const c_int, transport: c_int, out: *mut *mut spal_mg) -> c_int;
    pub fn spal_mg_destroy(ctx: *mut spal_mg) -> c_int;
    pub fn spal_mg_device_count(ctx: *mut spal_mg, ngpus: *mut c_int) -> c_int;
    pub fn spal_mg_transport(ctx: *mut spal_mg, transport: *mut c_int) -> c_int;
    pub fn spal_mg_csr_create_f64(ctx: *mut spal_mg, nrows: u64, ncols: u64, rowptr: *const u64, rowptr_len: u64, colind: *const u64, colind_len: u64, values: *const f64, values_len: u64, out: *mut *mut spal_mg_csr) -> c_int;
    pub fn spal_mg_csr_create_f32(ctx: *mut spal_mg, nrows: u64, ncols: u64, rowptr: *const u64, rowptr_len: u64, colind: *const u64, colind_len: u64, values: *const f32, values_len: u64, out: *mut *mut spal_mg_csr) -> c_int;
    pub fn spal_mg_csr_destroy(a: *mut spal_mg_csr) -> c_int;
    pub fn spal_mg_csr_partition(a: *mut spal_mg_csr, bounds: *mut u64) -> c_int;
    pub fn spal_mg_csr_windows(a: *mut spal_mg_csr, need_lo: *mut u64, need_hi: *mut u64) -> c_int;
    pub fn spal_mg_csr_exchange_bytes(a: *mut spal_mg_csr, x_scatter: *mut u64, y_gather: *mut u64, halo: *mut u64) -> c_int;
    pub fn spal_mg_csr_spmv_f64(a: *mut spal_mg_csr, x: *const f64, x_len: u64, y: *mut f64, y_len: u64) -> c_int;
    pub fn spal_mg_csr_spmv_f32(a: *mut spal_mg_csr, x: *const f32, x_len: u64, y: *mut f32, y_len: u64) -> c_int;
    pub fn spal_mg_csr_x_root(a: *mut spal_mg_csr, x_dev: *mut *mut c_void) -> c_int;
    pub fn spal_mg_csr_broadcast_x(a: *mut spal_mg_csr) -> c_int;
    pub fn spal_mg_csr_scatter_x(a: *mut spal_mg_csr) -> c_int;
    pub fn spal_mg_csr_spmv_local(a: *mut spal_mg_csr) -> c_int;
    pub fn spal_mg_csr_gather_y(a: *mut spal_mg_csr) -> c_int;
    pub fn spal_mg_csr_y_gathered(a: *mut spal_mg_csr, y_dev: *mut *mut c_void) -> c_int;
    pub fn spal_mg_csr_spmv_halo(a: *mut spal_mg_csr) -> c_int;
    pub fn spal_mg_csr_spmv_resident(a: *mut spal_mg_csr) -> c_int;
    pub fn spal_mg_csr_y_root(a: *mut spal_mg_csr, y_dev: *mut *mut c_void, slice_stride: *mut u64) -> c_int;
    pub fn spal_mg_csr_synchronize(a: *mut spal_mg_csr) -> c_int;
    pub fn spal_mg_csr_timing(a: *mut spal_mg_csr, ms: *mut f64) -> c_int;
    pub fn spal_dev_malloc(device: c_int, bytes: usize, ptr: *mut *mut c_void) -> c_int;
    pub fn spal_dev_free(device: c_int, ptr: *mut c_void) -> c_int;
    pub fn spal_memcpy_h2d(device: c_int, dst_dev: *mut c_void, src_host: *const c_void, bytes: usize) -> c_int;
    pub fn spal_memcpy_d2h(device: c_int, dst_host: *mut c_void, src_dev: *const c_void, bytes: usize) -> c_int;
    pub fn spal_device_synchronize(device: c_int) -> c_int;
    pub fn spal_cache_trim() -> c_int;
}

/// The reference panics on contract violations (`assert!`, src/csr.rs:144-156; `assert_eq!`,
/// src/csr/ops/mul.rs:9); every non-zero status keeps that convention.
pub fn check(status: c_int) {
    if status != SPAL_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(spal_last_error()) }.to_string_lossy().into_owned();
        panic!("spal_hip status {}: {}", status, msg);
    }
}
