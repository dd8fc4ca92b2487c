//! Owned device handles.  A handle is created ONCE per matrix (validation, index narrowing, upload, kernel
//! plan: 0.3 s for 140M entries) and reused for every product; `Drop` releases the device copy.  The host
//! matrix is only borrowed for the duration of `new` (the C ABI copies what it needs).
use std::marker::PhantomData;
use std::os::raw::{c_int, c_void};

use super::{ffi, scalar::HipScalar};
use crate::{CooMatrix, CscMatrix, CsrMatrix};

/// `CsrMatrix<T>` resident on a GPU (include/spal.h: spal_csr_t).
pub struct DeviceCsr<T: HipScalar> {
    pub(crate) h: *mut ffi::spal_csr,
    _t: PhantomData<T>,
}
/// `CscMatrix<T>` resident on a GPU (spal_csc_t).
pub struct DeviceCsc<T: HipScalar> {
    pub(crate) h: *mut ffi::spal_csc,
    _t: PhantomData<T>,
}
/// `CooMatrix<T>` triplets resident on a GPU (spal_coo_t): the input of the timed assembly path.
pub struct DeviceCoo<T: HipScalar> {
    pub(crate) h: *mut ffi::spal_coo,
    _t: PhantomData<T>,
}

// The C ABI serialises what needs it (per-handle mutex) and `spmv_dev` is read-only on the handle: a handle
// may move between threads and be shared, like the `Vec`-backed reference types (auto Send + Sync).
unsafe impl<T: HipScalar> Send for DeviceCsr<T> {}
unsafe impl<T: HipScalar> Sync for DeviceCsr<T> {}
unsafe impl<T: HipScalar> Send for DeviceCsc<T> {}
unsafe impl<T: HipScalar> Sync for DeviceCsc<T> {}
unsafe impl<T: HipScalar> Send for DeviceCoo<T> {}

impl<T: HipScalar> Drop for DeviceCsr<T> {
    fn drop(&mut self) { unsafe { ffi::spal_csr_destroy(self.h); } }
}
impl<T: HipScalar> Drop for DeviceCsc<T> {
    fn drop(&mut self) { unsafe { ffi::spal_csc_destroy(self.h); } }
}
impl<T: HipScalar> Drop for DeviceCoo<T> {
    fn drop(&mut self) { unsafe { ffi::spal_coo_destroy(self.h); } }
}

impl<T: HipScalar> DeviceCsr<T> {
    /// Uploads `a` to GPU `device`.  Re-checks the invariants of `CsrMatrix::new` (src/csr.rs:144-156): a
    /// matrix built through the public API passes.
    pub fn new(a: &CsrMatrix<T>, device: i32) -> Self {
        let mut h = std::ptr::null_mut();
        unsafe {
            ffi::check(T::csr_create(device as c_int, a.nrows() as u64, a.ncols() as u64, a.rowptr(), a.colind(),
                                     a.values(), &mut h));
        }
        DeviceCsr { h, _t: PhantomData }
    }
    pub(crate) fn from_raw(h: *mut ffi::spal_csr) -> Self { DeviceCsr { h, _t: PhantomData } }

    /// (nrows, ncols, nnz)  (src/csr.rs:200-222, :287-289)
    pub fn shape(&self) -> (usize, usize, usize) {
        let (mut nr, mut nc, mut nz, mut es) = (0u64, 0u64, 0u64, 0 as c_int);
        unsafe { ffi::check(ffi::spal_csr_shape(self.h, &mut nr, &mut nc, &mut nz, &mut es)); }
        (nr as usize, nc as usize, nz as usize)
    }

    /// y = A * x with host vectors (H2D x, kernel, D2H y).  Panics when `x.len() != ncols`, like
    /// `assert_eq!(self.ncols(), rhs.nrows())` in src/csr/ops/mul.rs:9.
    pub fn mul_vec(&self, x: &[T]) -> Vec<T> {
        let (nrows, ncols, _) = self.shape();
        assert_eq!(ncols, x.len());
        let mut y = vec![T::zero(); nrows];
        unsafe { ffi::check(T::csr_spmv(self.h, x, &mut y)); }
        y
    }

    /// The timed path: `x_dev` (ncols) and `y_dev` (nrows) are device pointers on this handle's GPU, the launch is
    /// enqueued on `stream` (a hipStream_t; null = the default stream) and not synchronised.
    ///
    /// # Safety
    /// the pointers must be valid device allocations of those lengths that do not overlap.
    pub unsafe fn mul_dev(&self, x_dev: *const T, y_dev: *mut T, stream: *mut c_void) {
        ffi::check(T::csr_spmv_dev(self.h, x_dev, y_dev, stream));
    }

    /// Setup-time autotune on the caller's device vectors (kernel form, placement of the values array).
    ///
    /// # Safety
    /// as `mul_dev`.
    pub unsafe fn autotune(&self, x_dev: *const T, y_dev: *mut T, stream: *mut c_void, iters: i32) {
        ffi::check(T::csr_autotune(self.h, x_dev, y_dev, stream, iters as c_int));
    }

    /// Device vectors `(x, y)` of `ncols` / `nrows` elements owned by this handle, placed so that the stores of y do
    /// not collide with the matrix stream (include/spal.h: spal_csr_alloc_vectors; a second call returns the same
    /// pointers; freed with the handle).  Setup time: a walk over the device's memory.
    pub fn vectors(&self, stream: *mut c_void) -> (*mut T, *mut T) {
        let (mut x, mut y): (*mut c_void, *mut c_void) = (std::ptr::null_mut(), std::ptr::null_mut());
        unsafe { ffi::check(ffi::spal_csr_alloc_vectors(self.h, &mut x, &mut y, stream)); }
        (x as *mut T, y as *mut T)
    }

    /// Builds the product kernels' plan of a device-assembled handle now (include/spal.h: spal_csr_plan); otherwise its first
    /// product, `set_option`, `autotune` or `vectors` does.  No-op on handles created from host arrays.
    pub fn plan(&self) {
        unsafe { ffi::check(ffi::spal_csr_plan(self.h)); }
    }

    /// Kernel plan knob (include/spal.h: spal_csr_set_option).
    pub fn set_option(&self, key: &str, value: i64) {
        let k = std::ffi::CString::new(key).expect("option key");
        unsafe { ffi::check(ffi::spal_csr_set_option(self.h, k.as_ptr(), value)); }
    }

    /// The device matrix back on the host.  The device upholds `CsrMatrix::new`'s invariants by construction.
    pub fn download(&self) -> CsrMatrix<T> {
        let (nr, nc, nz) = self.shape();
        let (mut rowptr, mut colind, mut values) = (vec![0usize; nr + 1], vec![0usize; nz], vec![T::zero(); nz]);
        unsafe { ffi::check(T::csr_download(self.h, &mut rowptr, &mut colind, &mut values)); }
        CsrMatrix::new(nr, nc, rowptr, colind, values)
    }

    /// Device twin of `impl From<&CsrMatrix<T>> for CscMatrix<T>` (src/csc/conv/csr.rs:4-52): a stable sort of the
    /// entries by column; entries are only moved, so the result equals the reference's exactly.
    pub fn to_csc(&self) -> DeviceCsc<T> {
        let mut out = std::ptr::null_mut();
        unsafe { ffi::check(ffi::spal_csr_to_csc(self.h, &mut out)); }
        DeviceCsc { h: out, _t: PhantomData }
    }
}

impl<T: HipScalar> DeviceCsc<T> {
    pub fn new(a: &CscMatrix<T>, device: i32) -> Self {
        let mut h = std::ptr::null_mut();
        unsafe {
            ffi::check(T::csc_create(device as c_int, a.nrows() as u64, a.ncols() as u64, a.colptr(), a.rowind(),
                                     a.values(), &mut h));
        }
        DeviceCsc { h, _t: PhantomData }
    }

    pub fn shape(&self) -> (usize, usize, usize) {
        let (mut nr, mut nc, mut nz, mut es) = (0u64, 0u64, 0u64, 0 as c_int);
        unsafe { ffi::check(ffi::spal_csc_shape(self.h, &mut nr, &mut nc, &mut nz, &mut es)); }
        (nr as usize, nc as usize, nz as usize)
    }

    /// Device-pointer products of this handle whose neighbour hand-off hit its spin bound (their y was not valid):
    /// include/spal.h: spal_csc_status; call after synchronising the stream.  0 on every other route.
    pub fn invalid_products(&self) -> i32 {
        let mut n: c_int = 0;
        unsafe { ffi::check(ffi::spal_csc_status(self.h, &mut n)); }
        n as i32
    }

    pub fn mul_vec(&self, x: &[T]) -> Vec<T> {
        let (nrows, ncols, _) = self.shape();
        assert_eq!(ncols, x.len());   // src/csc/ops/mul.rs:9
        let mut y = vec![T::zero(); nrows];
        unsafe { ffi::check(T::csc_spmv(self.h, x, &mut y)); }
        y
    }

    /// # Safety
    /// as `DeviceCsr::mul_dev`.
    pub unsafe fn mul_dev(&self, x_dev: *const T, y_dev: *mut T, stream: *mut c_void) {
        ffi::check(T::csc_spmv_dev(self.h, x_dev, y_dev, stream));
    }

    /// "kernel" = 1: atomic scatter, 2 (default): converted to CSR on the device once, deterministic.
    pub fn set_option(&self, key: &str, value: i64) {
        let k = std::ffi::CString::new(key).expect("option key");
        unsafe { ffi::check(ffi::spal_csc_set_option(self.h, k.as_ptr(), value)); }
    }

    pub fn download(&self) -> CscMatrix<T> {
        let (nr, nc, nz) = self.shape();
        let (mut colptr, mut rowind, mut values) = (vec![0usize; nc + 1], vec![0usize; nz], vec![T::zero(); nz]);
        unsafe { ffi::check(T::csc_download(self.h, &mut colptr, &mut rowind, &mut values)); }
        CscMatrix::new(nr, nc, colptr, rowind, values)
    }

    /// Device twin of `impl From<&CscMatrix<T>> for CsrMatrix<T>` (src/csr/conv/csc.rs:4-52).
    pub fn to_csr(&self) -> DeviceCsr<T> {
        let mut out = std::ptr::null_mut();
        unsafe { ffi::check(ffi::spal_csc_to_csr(self.h, &mut out)); }
        DeviceCsr::from_raw(out)
    }
}

impl<T: HipScalar> DeviceCoo<T> {
    /// Uploads the triplets in insertion order.  `Vec<(usize, usize, T)>` has no guaranteed layout, so `iter()`
    /// (src/coo.rs:491) is unzipped into three arrays; bounds are re-checked (src/coo.rs:432-433).
    pub fn new(coo: &CooMatrix<T>, device: i32) -> Self {
        let (mut r, mut c, mut v) = (Vec::new(), Vec::new(), Vec::new());
        for (row, col, val) in coo.iter() {
            r.push(row);
            c.push(col);
            v.push(*val);
        }
        let mut h = std::ptr::null_mut();
        unsafe { ffi::check(T::coo_upload(device as c_int, coo.nrows() as u64, coo.ncols() as u64, &r, &c, &v, &mut h)); }
        DeviceCoo { h, _t: PhantomData }
    }

    /// Device twin of `impl From<&CooMatrix<T>> for CsrMatrix<T>` (src/csr/conv/coo.rs:4-115): stable order by
    /// (row, col), duplicates summed left to right in insertion order, results equal to zero dropped --
    /// rowptr, colind and values bit-identical to the reference's.
    pub fn assemble_csr(&self) -> DeviceCsr<T> {
        let mut out = std::ptr::null_mut();
        unsafe { ffi::check(ffi::spal_coo_assemble_csr(self.h, std::ptr::null_mut(), &mut out)); }
        DeviceCsr::from_raw(out)
    }

    /// ... and of `impl From<&CooMatrix<T>> for CscMatrix<T>` (src/csc/conv/coo.rs:4-115).
    pub fn assemble_csc(&self) -> DeviceCsc<T> {
        let mut out = std::ptr::null_mut();
        unsafe { ffi::check(ffi::spal_coo_assemble_csc(self.h, std::ptr::null_mut(), &mut out)); }
        DeviceCsc { h: out, _t: PhantomData }
    }
}
