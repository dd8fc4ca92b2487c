//! The operator and conversion impls the reference lacks for dense vectors (its only route to `A * x` is
//! `&a * &x_as_matrix`, src/csr/ops/mul.rs:5-59), over the device handles of `device.rs`.
use std::ops::Mul;

use super::device::{DeviceCoo, DeviceCsc, DeviceCsr};
use super::scalar::HipScalar;
use crate::{CooMatrix, CscMatrix, CsrMatrix};

/// y = A * x on the resident matrix: the form to use when a matrix multiplies more than once.
impl<T: HipScalar> Mul<&[T]> for &DeviceCsr<T> {
    type Output = Vec<T>;
    fn mul(self, x: &[T]) -> Vec<T> { self.mul_vec(x) }
}
impl<T: HipScalar> Mul<&[T]> for &DeviceCsc<T> {
    type Output = Vec<T>;
    fn mul(self, x: &[T]) -> Vec<T> { self.mul_vec(x) }
}

/// One-shot convenience with the reference's operator shape: uploads `self`, multiplies, frees the device copy.
/// The upload dominates (1.7 GB for 140M entries); keep a `DeviceCsr` (`DeviceCsr::from(&a)`) for repeated
/// products.  Panics when `x.len() != ncols` (src/csr/ops/mul.rs:9).
impl<T: HipScalar> Mul<&[T]> for &CsrMatrix<T> {
    type Output = Vec<T>;
    fn mul(self, x: &[T]) -> Vec<T> {
        assert_eq!(self.ncols(), x.len());
        DeviceCsr::new(self, 0).mul_vec(x)
    }
}
impl<T: HipScalar> Mul<&[T]> for &CscMatrix<T> {
    type Output = Vec<T>;
    fn mul(self, x: &[T]) -> Vec<T> {
        assert_eq!(self.ncols(), x.len());
        DeviceCsc::new(self, 0).mul_vec(x)
    }
}

impl<T: HipScalar> From<&CsrMatrix<T>> for DeviceCsr<T> {
    fn from(a: &CsrMatrix<T>) -> Self { DeviceCsr::new(a, 0) }
}
impl<T: HipScalar> From<&CscMatrix<T>> for DeviceCsc<T> {
    fn from(a: &CscMatrix<T>) -> Self { DeviceCsc::new(a, 0) }
}
/// `CsrMatrix::from(&coo)` on the device (src/csr/conv/coo.rs:3): upload, assemble, the COO copy is freed.
impl<T: HipScalar> From<&CooMatrix<T>> for DeviceCsr<T> {
    fn from(coo: &CooMatrix<T>) -> Self { DeviceCoo::new(coo, 0).assemble_csr() }
}
/// `CscMatrix::from(&coo)` on the device (src/csc/conv/coo.rs:3).
impl<T: HipScalar> From<&CooMatrix<T>> for DeviceCsc<T> {
    fn from(coo: &CooMatrix<T>) -> Self { DeviceCoo::new(coo, 0).assemble_csc() }
}
/// `CsrMatrix::from(&csc)` (src/csr/conv/csc.rs:4) and `CscMatrix::from(&csr)` (src/csc/conv/csr.rs:4) between
/// resident matrices: a stable sort by the minor index on the device.
impl<T: HipScalar> From<&DeviceCsc<T>> for DeviceCsr<T> {
    fn from(a: &DeviceCsc<T>) -> Self { a.to_csr() }
}
impl<T: HipScalar> From<&DeviceCsr<T>> for DeviceCsc<T> {
    fn from(a: &DeviceCsr<T>) -> Self { a.to_csc() }
}

/// Host results with the reference's own conversion signatures, computed on the device:
/// `csr_from_coo_hip(&coo)` == `CsrMatrix::from(&coo)` bit for bit.
pub fn csr_from_coo_hip<T: HipScalar>(coo: &CooMatrix<T>) -> CsrMatrix<T> { DeviceCsr::from(coo).download() }
pub fn csc_from_coo_hip<T: HipScalar>(coo: &CooMatrix<T>) -> CscMatrix<T> { DeviceCsc::from(coo).download() }
pub fn csr_from_csc_hip<T: HipScalar>(a: &CscMatrix<T>) -> CsrMatrix<T> { DeviceCsc::new(a, 0).to_csr().download() }
pub fn csc_from_csr_hip<T: HipScalar>(a: &CsrMatrix<T>) -> CscMatrix<T> { DeviceCsr::new(a, 0).to_csc().download() }
