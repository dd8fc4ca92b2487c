//! `impl Mul<&[T]> for &CsrMatrix<T>` / `&CscMatrix<T>` and the device
//! `From<&CooMatrix<T>>` over libspal_hip.so.
//!
//! NOT COMPILED IN THIS REPOSITORY'S PIPELINE (no rustc in the image).  Added to
//! the crate as `src/hip/mulvec.rs`, registered the way `src/csr/ops.rs:1-4`
//! registers the operator modules.  `usize == u64` is assumed
//! (x86-64 Linux): the slices are passed as-is, no copy.
use std::ops::Mul;

use super::ffi;
use crate::{scalar::Scalar, CooMatrix, CscMatrix, CsrMatrix};

/// Per-`Scalar` dispatch to the two instantiations of the C ABI
/// (`Scalar` is implemented for f32 and f64 only, src/scalar.rs:56-57).
pub trait HipScalar: Scalar {
    unsafe fn csr_create(d: i32, nr: u64, nc: u64, rp: &[usize], ci: &[usize], v: &[Self]) -> *mut ffi::spal_csr;
    unsafe fn csr_spmv(a: *mut ffi::spal_csr, x: &[Self], y: &mut [Self]);
    unsafe fn csc_create(d: i32, nr: u64, nc: u64, cp: &[usize], ri: &[usize], v: &[Self]) -> *mut ffi::spal_csc;
    unsafe fn csc_spmv(a: *mut ffi::spal_csc, x: &[Self], y: &mut [Self]);
    unsafe fn coo_to_csr(d: i32, nr: u64, nc: u64, r: &[usize], c: &[usize], v: &[Self]) -> *mut ffi::spal_csr;
    unsafe fn csr_download(a: *mut ffi::spal_csr, rp: &mut [usize], ci: &mut [usize], v: &mut [Self]);
}

// (the crate has no dependencies, Cargo.toml:13, so the entry points are
// passed to the macro by name instead of being pasted together)
macro_rules! impl_hip_scalar {
    ($t:ty, $csr_create:ident, $csr_spmv:ident, $csc_create:ident, $csc_spmv:ident,
     $coo_to_csr:ident, $csr_download:ident) => {
        impl HipScalar for $t {
            unsafe fn csr_create(d: i32, nr: u64, nc: u64, rp: &[usize], ci: &[usize], v: &[Self]) -> *mut ffi::spal_csr {
                let mut h = std::ptr::null_mut();
                ffi::check(ffi::$csr_create(d, nr, nc, rp.as_ptr() as *const u64, rp.len() as u64,
                    ci.as_ptr() as *const u64, ci.len() as u64, v.as_ptr(), v.len() as u64, &mut h));
                h
            }
            unsafe fn csr_spmv(a: *mut ffi::spal_csr, x: &[Self], y: &mut [Self]) {
                ffi::check(ffi::$csr_spmv(a, x.as_ptr(), x.len() as u64, y.as_mut_ptr(), y.len() as u64));
            }
            unsafe fn csc_create(d: i32, nr: u64, nc: u64, cp: &[usize], ri: &[usize], v: &[Self]) -> *mut ffi::spal_csc {
                let mut h = std::ptr::null_mut();
                ffi::check(ffi::$csc_create(d, nr, nc, cp.as_ptr() as *const u64, cp.len() as u64,
                    ri.as_ptr() as *const u64, ri.len() as u64, v.as_ptr(), v.len() as u64, &mut h));
                h
            }
            unsafe fn csc_spmv(a: *mut ffi::spal_csc, x: &[Self], y: &mut [Self]) {
                ffi::check(ffi::$csc_spmv(a, x.as_ptr(), x.len() as u64, y.as_mut_ptr(), y.len() as u64));
            }
            unsafe fn coo_to_csr(d: i32, nr: u64, nc: u64, r: &[usize], c: &[usize], v: &[Self]) -> *mut ffi::spal_csr {
                let mut h = std::ptr::null_mut();
                ffi::check(ffi::$coo_to_csr(d, nr, nc, v.len() as u64, r.as_ptr() as *const u64,
                    c.as_ptr() as *const u64, v.as_ptr(), &mut h));
                h
            }
            unsafe fn csr_download(a: *mut ffi::spal_csr, rp: &mut [usize], ci: &mut [usize], v: &mut [Self]) {
                ffi::check(ffi::$csr_download(a, rp.as_mut_ptr() as *mut u64, ci.as_mut_ptr() as *mut u64, v.as_mut_ptr()));
            }
        }
    };
}
impl_hip_scalar!(f64, spal_csr_create_f64, spal_csr_spmv_f64, spal_csc_create_f64, spal_csc_spmv_f64,
                 spal_coo_to_csr_f64, spal_csr_download_f64);
impl_hip_scalar!(f32, spal_csr_create_f32, spal_csr_spmv_f32, spal_csc_create_f32, spal_csc_spmv_f32,
                 spal_coo_to_csr_f32, spal_csr_download_f32);

/// y = A * x.  Panics when `x.len() != ncols`, like `assert_eq!(self.ncols(),
/// rhs.nrows())` in src/csr/ops/mul.rs:9.
impl<T: HipScalar> Mul<&[T]> for &CsrMatrix<T> {
    type Output = Vec<T>;
    fn mul(self, x: &[T]) -> Vec<T> {
        assert_eq!(self.ncols(), x.len());
        let mut y = vec![T::zero(); self.nrows()];
        unsafe {
            // a production binding caches the handle next to the matrix and frees it in Drop
            let h = T::csr_create(0, self.nrows() as u64, self.ncols() as u64, self.rowptr(), self.colind(), self.values());
            T::csr_spmv(h, x, &mut y);
            ffi::check(ffi::spal_csr_destroy(h));
        }
        y
    }
}

impl<T: HipScalar> Mul<&[T]> for &CscMatrix<T> {
    type Output = Vec<T>;
    fn mul(self, x: &[T]) -> Vec<T> {
        assert_eq!(self.ncols(), x.len());
        let mut y = vec![T::zero(); self.nrows()];
        unsafe {
            let h = T::csc_create(0, self.nrows() as u64, self.ncols() as u64, self.colptr(), self.rowind(), self.values());
            T::csc_spmv(h, x, &mut y);
            ffi::check(ffi::spal_csc_destroy(h));
        }
        y
    }
}

/// Device twin of `impl From<&CooMatrix<T>> for CsrMatrix<T>`
/// (src/csr/conv/coo.rs:3-116); results are bit-identical to it.
pub fn csr_from_coo_hip<T: HipScalar>(coo: &CooMatrix<T>) -> CsrMatrix<T> {
    // Vec<(usize, usize, T)> has no guaranteed layout: unzip `iter()` (src/coo.rs:491)
    let (mut r, mut c, mut v) = (Vec::new(), Vec::new(), Vec::new());
    for (row, col, val) in coo.iter() { r.push(row); c.push(col); v.push(*val); }
    unsafe {
        let h = T::coo_to_csr(0, coo.nrows() as u64, coo.ncols() as u64, &r, &c, &v);
        let (mut nr, mut nc, mut nz, mut es) = (0u64, 0u64, 0u64, 0i32);
        ffi::check(ffi::spal_csr_shape(h, &mut nr, &mut nc, &mut nz, &mut es));
        let mut rowptr = vec![0usize; nr as usize + 1];
        let mut colind = vec![0usize; nz as usize];
        let mut values = vec![T::zero(); nz as usize];
        T::csr_download(h, &mut rowptr, &mut colind, &mut values);
        ffi::check(ffi::spal_csr_destroy(h));
        // the assembly upholds CsrMatrix::new's invariants by construction
        CsrMatrix::new(nr as usize, nc as usize, rowptr, colind, values)
    }
}
