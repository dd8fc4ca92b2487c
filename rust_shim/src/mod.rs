//! `spalinalg::hip` -- the MI355X path underneath the crate's public types.
//!
//! NOT COMPILED IN THIS REPOSITORY'S PIPELINE (no rustc in the image): source a maintainer
//! adds to the crate as `src/hip/`, registered from `src/lib.rs` as `pub mod hip;` the way `src/csr/ops.rs:1-4`
//! registers the operator modules.  `usize == u64` is assumed (x86-64 Linux): index slices cross the ABI as
//! they are, without a copy.
//!
//! * `ffi`     -- generated `extern "C"` declarations of every function of include/spal.h
//! * `scalar`  -- `HipScalar`: per-`Scalar` dispatch to the `_f32` / `_f64` entry points
//! * `device`  -- `DeviceCsr`, `DeviceCsc`, `DeviceCoo`: owned device handles (freed in `Drop`), created once per
//!                matrix and reused for every product; conversions between them on the device
//! * `ops`     -- `impl Mul<&[T]>` for `&CsrMatrix<T>`, `&CscMatrix<T>` and the device handles;
//!                `impl From<&CooMatrix<T>>` / `From<&CscMatrix<T>>` / `From<&CsrMatrix<T>>` for the device handles
//! * `multi`   -- `MultiGpuCsr`: the row-partitioned product over the GPUs of one node
pub mod device;
pub mod ffi;
pub mod multi;
pub mod ops;
pub mod scalar;

pub use device::{DeviceCoo, DeviceCsc, DeviceCsr};
pub use multi::MultiGpuCsr;
pub use scalar::HipScalar;
