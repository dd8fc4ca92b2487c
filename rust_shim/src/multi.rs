//! The row-partitioned product over the GPUs of one node, driven from this one process (include/spal.h:
//! spal_mg_*; SURVEY.md sections 8e / 8f-4): every GPU reads only its window of x, the y slices come back to
//! GPU 0; for iterative use a per-step halo exchange.
use std::marker::PhantomData;
use std::os::raw::{c_int, c_void};

use super::{ffi, scalar::HipScalar};
use crate::CsrMatrix;

/// How bytes move between the GPUs: RCCL (grouped ncclSend / ncclRecv over xGMI) or peer copies.
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum Transport { Auto = -1, Rccl = 0, Copy = 1 }

pub struct MultiGpuCsr<T: HipScalar> {
    ctx: *mut ffi::spal_mg,
    h: *mut ffi::spal_mg_csr,
    nrows: usize,
    ncols: usize,
    ngpus: usize,
    _t: PhantomData<T>,
}
unsafe impl<T: HipScalar> Send for MultiGpuCsr<T> {}

/// HIP-event durations (ms, the longest over the GPUs) of the last calls; `None`: that phase has not run.
#[derive(Clone, Copy, Debug, Default)]
pub struct PhaseTimes { pub x_distribution: Option<f64>, pub compute: Option<f64>, pub halo: Option<f64>, pub y_collection: Option<f64> }

impl<T: HipScalar> MultiGpuCsr<T> {
    /// Partitions `a` by rows (balanced stored entries) over `devices` (None: GPUs 0 .. ngpus - 1).
    pub fn new(a: &CsrMatrix<T>, ngpus: usize, devices: Option<&[i32]>, transport: Transport) -> Self {
        let (mut ctx, mut h) = (std::ptr::null_mut(), std::ptr::null_mut());
        let devs: Option<Vec<c_int>> = devices.map(|d| d.iter().map(|&x| x as c_int).collect());
        unsafe {
            ffi::check(ffi::spal_mg_create_transport(ngpus as c_int, devs.as_ref().map_or(std::ptr::null(), |d| d.as_ptr()),
                                                     transport as c_int, &mut ctx));
            let st = T::mg_csr_create(ctx, a.nrows() as u64, a.ncols() as u64, a.rowptr(), a.colind(), a.values(), &mut h);
            if st != ffi::SPAL_OK {
                ffi::spal_mg_destroy(ctx);
                ffi::check(st);
            }
        }
        MultiGpuCsr { ctx, h, nrows: a.nrows(), ncols: a.ncols(), ngpus, _t: PhantomData }
    }

    /// the ngpus + 1 row boundaries in use
    pub fn partition(&self) -> Vec<usize> {
        let mut b = vec![0u64; self.ngpus + 1];
        unsafe { ffi::check(ffi::spal_mg_csr_partition(self.h, b.as_mut_ptr())); }
        b.into_iter().map(|x| x as usize).collect()
    }

    /// per GPU: its rows store columns in `[lo, hi)` only
    pub fn windows(&self) -> Vec<(usize, usize)> {
        let (mut lo, mut hi) = (vec![0u64; self.ngpus], vec![0u64; self.ngpus]);
        unsafe { ffi::check(ffi::spal_mg_csr_windows(self.h, lo.as_mut_ptr(), hi.as_mut_ptr())); }
        lo.into_iter().zip(hi).map(|(a, b)| (a as usize, b as usize)).collect()
    }

    /// y = A * x with host vectors: x to GPU 0, windows scattered, local kernels, y gathered on GPU 0, back.
    pub fn mul_vec(&self, x: &[T]) -> Vec<T> {
        assert_eq!(self.ncols, x.len());   // src/csr/ops/mul.rs:9
        let mut y = vec![T::zero(); self.nrows];
        unsafe { ffi::check(T::mg_csr_spmv(self.h, x, &mut y)); }
        y
    }

    // ---- the resident path: asynchronous on the context's streams until `synchronize`
    /// GPU 0's x buffer (valid until the next `spmv_halo`): write x there, then `scatter_x` or `broadcast_x`.
    pub fn x_root(&self) -> *mut T {
        let mut p: *mut c_void = std::ptr::null_mut();
        unsafe { ffi::check(ffi::spal_mg_csr_x_root(self.h, &mut p)); }
        p as *mut T
    }
    /// GPU 0's y (nrows elements) after `gather_y`.
    pub fn y_gathered(&self) -> *const T {
        let mut p: *mut c_void = std::ptr::null_mut();
        unsafe { ffi::check(ffi::spal_mg_csr_y_gathered(self.h, &mut p)); }
        p as *const T
    }
    pub fn set_x(&self, x: &[T]) {
        assert_eq!(self.ncols, x.len());
        self.synchronize();
        unsafe { ffi::check(ffi::spal_memcpy_h2d(0, self.x_root() as *mut c_void, x.as_ptr() as *const c_void, std::mem::size_of_val(x))); }
    }
    pub fn broadcast_x(&self) { unsafe { ffi::check(ffi::spal_mg_csr_broadcast_x(self.h)); } }
    pub fn scatter_x(&self) { unsafe { ffi::check(ffi::spal_mg_csr_scatter_x(self.h)); } }
    pub fn spmv_local(&self) { unsafe { ffi::check(ffi::spal_mg_csr_spmv_local(self.h)); } }
    pub fn gather_y(&self) { unsafe { ffi::check(ffi::spal_mg_csr_gather_y(self.h)); } }
    /// square matrices: x <- A * x with the halo exchange (y of one step is the x of the next)
    pub fn spmv_halo(&self) { unsafe { ffi::check(ffi::spal_mg_csr_spmv_halo(self.h)); } }
    /// kernels + all-gather: every GPU ends with all of y
    pub fn spmv_resident(&self) { unsafe { ffi::check(ffi::spal_mg_csr_spmv_resident(self.h)); } }
    pub fn synchronize(&self) { unsafe { ffi::check(ffi::spal_mg_csr_synchronize(self.h)); } }

    pub fn timing(&self) -> PhaseTimes {
        let mut ms = [0f64; 4];
        unsafe { ffi::check(ffi::spal_mg_csr_timing(self.h, ms.as_mut_ptr())); }
        let f = |v: f64| if v < 0.0 { None } else { Some(v) };
        PhaseTimes { x_distribution: f(ms[0]), compute: f(ms[1]), halo: f(ms[2]), y_collection: f(ms[3]) }
    }
}

impl<T: HipScalar> Drop for MultiGpuCsr<T> {
    fn drop(&mut self) {
        unsafe {
            ffi::spal_mg_csr_destroy(self.h);
            ffi::spal_mg_destroy(self.ctx);
        }
    }
}

impl<T: HipScalar> std::ops::Mul<&[T]> for &MultiGpuCsr<T> {
    type Output = Vec<T>;
    fn mul(self, x: &[T]) -> Vec<T> { self.mul_vec(x) }
}
