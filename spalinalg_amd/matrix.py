"""Host-side mirror of the reference's matrix types over the C ABI.

Names, argument meaning and error behaviour follow the reference
(lokyhark/spalinalg; paths relative to its root):

* ``CsrMatrix(nrows, ncols, rowptr, colind, values)``  == ``CsrMatrix::new``
  (src/csr.rs:137-164) incl. its panics (-> :class:`Panic`),
  ``nrows()/ncols()/rowptr()/colind()/values()/nnz()`` (src/csr.rs:200-289).
* ``a * x`` / ``a @ x`` with a dense vector: the product the reference only
  reaches through ``&a * &x_as_matrix`` (src/csr/ops/mul.rs:5-59); what a Rust
  binding adds as ``impl Mul<&[T]> for &CsrMatrix<T>``.
* ``CsrMatrix.from_coo(coo)`` == ``CsrMatrix::from(&coo)``
  (src/csr/conv/coo.rs:3-116), assembled on the device.
* ``CscMatrix`` / ``CooMatrix`` likewise (src/csc.rs, src/coo.rs).

Host arrays stay numpy (``uint64`` indices like ``usize``, ``float64`` /
``float32`` values like the two ``Scalar`` impls); a device copy is created
on first use and owned by the :class:`DeviceCsr` / :class:`DeviceCsc` handle.
"""
from __future__ import annotations

import ctypes as C
import json

import numpy as np

from . import _ffi
from ._ffi import Panic, check, f32p, f64p, u64, u64p, vp


def _scalar_dtype(values) -> np.dtype:
    dt = np.asarray(values).dtype
    if dt == np.float32:
        return np.dtype(np.float32)
    if dt == np.float64 or dt.kind in "iub":
        return np.dtype(np.float64)
    raise TypeError("Scalar is implemented for f32 and f64 only (src/scalar.rs:56-57)")


def _sfx(dt: np.dtype) -> str:
    return "f64" if dt == np.float64 else "f32"


def _idx(a) -> np.ndarray:
    arr = np.asarray(a)
    if arr.size and arr.dtype.kind not in "iu":
        raise TypeError("indices must be integers (usize)")
    if arr.size and arr.dtype.kind == "i" and (arr < 0).any():
        raise Panic(_ffi.SPAL_ERR_INVALID_ARGUMENT, "negative index (usize cannot be negative)")
    return np.ascontiguousarray(arr, dtype=np.uint64)


def _p(a: np.ndarray):
    return a.ctypes.data_as({np.dtype(np.uint64): u64p, np.dtype(np.float64): f64p,
                             np.dtype(np.float32): f32p}[a.dtype])


def _stream_ptr(stream) -> vp:
    if stream is None:
        return vp(0)
    if isinstance(stream, int):
        return vp(stream)
    return vp(int(stream.cuda_stream))  # torch.cuda.Stream


# --------------------------------------------------------------------------
# device handles
# --------------------------------------------------------------------------
class _DeviceMatrix:
    _kind = ""  # "csr" | "csc"

    def __init__(self, handle, dtype: np.dtype, device: int):
        self._h = handle
        self.dtype = np.dtype(dtype)
        self.device = device

    def _fn(self, name):
        return getattr(_ffi.lib(), f"spal_{self._kind}_{name}")

    def close(self):
        if self._h is not None:
            self._fn("destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shape(self):
        nr, nc, nz, es = u64(), u64(), u64(), C.c_int()
        check(self._fn("shape")(self._h, C.byref(nr), C.byref(nc), C.byref(nz), C.byref(es)))
        return nr.value, nc.value, nz.value

    def set_option(self, key: str, value: int) -> None:
        check(self._fn("set_option")(self._h, key.encode(), C.c_int64(value)))

    def describe(self) -> dict:
        buf = C.create_string_buffer(16384)
        check(self._fn("describe")(self._h, buf, C.c_size_t(len(buf))))
        return json.loads(buf.value.decode())

    def spmv(self, x) -> np.ndarray:
        """Host vectors in, host vector out (H2D x, kernel, D2H y)."""
        x = np.ascontiguousarray(x, dtype=self.dtype)
        nrows = self.shape()[0]
        y = np.empty(nrows, dtype=self.dtype)
        check(self._fn(f"spmv_{_sfx(self.dtype)}")(self._h, _p(x), u64(x.size), _p(y), u64(y.size)))
        return y

    def spmv_dev(self, x_ptr: int, y_ptr: int, stream=None) -> None:
        """Device pointers; enqueued on `stream`, not synchronised."""
        check(self._fn(f"spmv_dev_{_sfx(self.dtype)}")(self._h, vp(x_ptr), vp(y_ptr), _stream_ptr(stream)))

    def alloc_vectors(self, stream=None):
        """Device pointers (x, y) of vectors owned by this handle and placed so that the stores of y do not collide
        with the matrix stream (spal_csr_alloc_vectors: a walk over the device's memory, setup time).  CSR handles."""
        if self._kind != "csr":   # the entry point reads a spal_csr: never hand it another handle type
            raise TypeError("alloc_vectors() exists for CSR handles only (spal_csr_alloc_vectors)")
        x, y = vp(), vp()
        check(_ffi.lib().spal_csr_alloc_vectors(self._h, C.byref(x), C.byref(y), _stream_ptr(stream)))
        return x.value, y.value

    def vectors_torch(self):
        """alloc_vectors() as torch tensors (zero-copy views of the handle's block; they keep the handle alive)."""
        import torch
        nrows, ncols, _ = self.shape()
        xp, yp = self.alloc_vectors(torch.cuda.current_stream(self.device))

        class _View:
            def __init__(self, owner, ptr, n, dtype):
                self._owner = owner
                self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": np.dtype(dtype).str, "data": (int(ptr), False),
                                                 "version": 2, "strides": None}
        dev = torch.device("cuda", self.device)
        return (torch.as_tensor(_View(self, xp, ncols, self.dtype), device=dev),
                torch.as_tensor(_View(self, yp, nrows, self.dtype), device=dev))

    def autotune(self, x, y, iters: int = 30) -> dict:
        """Times the plan's kernel variants on torch device vectors x, y and keeps
        the fastest (setup-time; synchronises the current stream)."""
        import torch
        st = torch.cuda.current_stream(x.device)
        check(self._fn(f"autotune_{_sfx(self.dtype)}")(self._h, vp(x.data_ptr()), vp(y.data_ptr()),
                                                       _stream_ptr(st), C.c_int(iters)))
        return self.describe()

    def spmv_torch(self, x, out=None):
        """x, out: torch tensors on this handle's device; runs on torch's
        current stream (so torch.cuda.Event brackets it)."""
        import torch
        nrows, ncols, _ = self.shape()
        tdt = torch.float64 if self.dtype == np.float64 else torch.float32
        if x.dtype != tdt or not x.is_cuda or not x.is_contiguous() or x.numel() != ncols:
            raise Panic(_ffi.SPAL_ERR_INVALID_ARGUMENT,
                        f"x must be a contiguous {tdt} device vector of length ncols = {ncols}")
        if x.device.index != self.device:
            raise Panic(_ffi.SPAL_ERR_INVALID_ARGUMENT,
                        f"x lives on cuda:{x.device.index} but the matrix on cuda:{self.device}")
        if out is None:
            out = torch.empty(nrows, dtype=tdt, device=x.device)
        elif out.dtype != tdt or not out.is_cuda or not out.is_contiguous() or out.numel() != nrows:
            raise Panic(_ffi.SPAL_ERR_INVALID_ARGUMENT,
                        f"out must be a contiguous {tdt} device vector of length nrows = {nrows}")
        elif out.device != x.device:
            raise Panic(_ffi.SPAL_ERR_INVALID_ARGUMENT, "out and x live on different devices")
        else:
            # the kernels read x and write y through __restrict__ pointers: the two must not overlap
            es = x.element_size()
            if out.data_ptr() < x.data_ptr() + ncols * es and x.data_ptr() < out.data_ptr() + nrows * es:
                raise Panic(_ffi.SPAL_ERR_INVALID_ARGUMENT, "out overlaps x (y = A*x is not computed in place)")
        self.spmv_dev(x.data_ptr(), out.data_ptr(), torch.cuda.current_stream(x.device))
        return out


class DeviceCsr(_DeviceMatrix):
    _kind = "csr"

    def plan(self) -> None:
        """spal_csr_plan: builds the product kernels' plan of a device-assembled handle now (otherwise its first product,
        set_option, autotune, alloc_vectors or describe does).  No-op on handles created from host arrays."""
        check(_ffi.lib().spal_csr_plan(self._h))

    def download(self):
        nrows, _, nnz = self.shape()
        rp = np.empty(nrows + 1, dtype=np.uint64)
        ci = np.empty(nnz, dtype=np.uint64)
        va = np.empty(nnz, dtype=self.dtype)
        check(self._fn(f"download_{_sfx(self.dtype)}")(self._h, _p(rp), _p(ci), _p(va)))
        return rp, ci, va

    def to_csc(self) -> "DeviceCsc":
        """device CSR -> CSC (stable sort by column; src/csc/conv/csr.rs:4-52)"""
        out = vp()
        check(_ffi.lib().spal_csr_to_csc(self._h, C.byref(out)))
        return DeviceCsc(out, self.dtype, self.device)


class DeviceCsc(_DeviceMatrix):
    _kind = "csc"

    def download(self):
        _, ncols, nnz = self.shape()
        cp = np.empty(ncols + 1, dtype=np.uint64)
        ri = np.empty(nnz, dtype=np.uint64)
        va = np.empty(nnz, dtype=self.dtype)
        check(self._fn(f"download_{_sfx(self.dtype)}")(self._h, _p(cp), _p(ri), _p(va)))
        return cp, ri, va

    def invalid_products(self) -> int:
        """spal_csc_status: device-pointer products of this handle whose hand-off hit its spin bound (their y was not
        valid); call after synchronising the stream."""
        n = C.c_int(-1)
        check(_ffi.lib().spal_csc_status(self._h, C.byref(n)))
        return n.value

    def to_csr(self) -> DeviceCsr:
        """device CSC -> CSR (stable sort by row; src/csr/conv/csc.rs:4-52)"""
        out = vp()
        check(_ffi.lib().spal_csc_to_csr(self._h, C.byref(out)))
        return DeviceCsr(out, self.dtype, self.device)


class DeviceCoo:
    """Device-resident COO triplets (SoA, 32-bit indices): the input of the
    timed assembly path."""

    def __init__(self, handle, dtype, device, nrows, ncols, length):
        self._h, self.dtype, self.device = handle, np.dtype(dtype), device
        self.nrows, self.ncols, self.length = nrows, ncols, length

    def assemble_csr(self, stream=None) -> DeviceCsr:
        """COO -> CSR on the device (one host sync for the output size)."""
        out = vp()
        check(_ffi.lib().spal_coo_assemble_csr(self._h, _stream_ptr(stream), C.byref(out)))
        return DeviceCsr(out, self.dtype, self.device)

    def assemble_csc(self, stream=None) -> DeviceCsc:
        """COO -> CSC on the device (src/csc/conv/coo.rs:4-115)."""
        out = vp()
        check(_ffi.lib().spal_coo_assemble_csc(self._h, _stream_ptr(stream), C.byref(out)))
        return DeviceCsc(out, self.dtype, self.device)

    def describe(self) -> dict:
        """The handle and the route its last assembly took (tile geometry of the local sort)."""
        buf = C.create_string_buffer(1024)
        check(_ffi.lib().spal_coo_describe(self._h, buf, C.c_size_t(len(buf))))
        return json.loads(buf.value.decode())

    def close(self):
        if self._h is not None:
            _ffi.lib().spal_coo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiGpuCsr:
    """A CsrMatrix partitioned by rows over `ngpus` GPUs of this node, driven
    from this one process (C ABI spal_mg_*): every GPU reads
    only its window of x (scatter_x; broadcast_x sends all of it), local
    kernels, y slices gathered on GPU 0 (gather_y) or all-gathered
    (spmv_resident), per-step halo exchange for iterative use (spmv_halo).

    transport: "rccl" (grouped ncclSend/ncclRecv over xGMI), "copy" (peer
    copies ordered by events; also takes a device list with repeats, i.e.
    several shards on one GPU) or None = rccl unless `devices` repeats."""

    PHASES = ("x_distribution", "compute", "halo", "y_collection")

    def __init__(self, csr: "CsrMatrix", ngpus: int, devices=None, transport=None):
        self.dtype = csr.dtype
        self._ctx, self._h = vp(), vp()
        dev = (C.c_int * ngpus)(*devices) if devices is not None else None
        tr = {None: -1, "rccl": 0, "copy": 1}[transport]
        check(_ffi.lib().spal_mg_create_transport(C.c_int(ngpus), dev, C.c_int(tr), C.byref(self._ctx)))
        try:
            check(getattr(_ffi.lib(), f"spal_mg_csr_create_{_sfx(self.dtype)}")(
                self._ctx, u64(csr.nrows()), u64(csr.ncols()), _p(csr.rowptr()), u64(csr.rowptr().size),
                _p(csr.colind()), u64(csr.colind().size), _p(csr.values()), u64(csr.values().size),
                C.byref(self._h)))
        except Exception:
            _ffi.lib().spal_mg_destroy(self._ctx)
            self._ctx = None
            raise
        self.nrows, self.ncols, self.ngpus = csr.nrows(), csr.ncols(), ngpus
        t = C.c_int(0)
        check(_ffi.lib().spal_mg_transport(self._ctx, C.byref(t)))
        self.transport = "rccl" if t.value == 0 else "copy"
        self.root_device = devices[0] if devices is not None else 0
        self.devices = list(devices) if devices is not None else list(range(ngpus))

    def partition(self) -> np.ndarray:
        b = np.empty(self.ngpus + 1, dtype=np.uint64)
        check(_ffi.lib().spal_mg_csr_partition(self._h, _p(b)))
        return b

    def windows(self):
        """per GPU: its rows store columns in [need_lo[g], need_hi[g]) only"""
        lo, hi = np.empty(self.ngpus, dtype=np.uint64), np.empty(self.ngpus, dtype=np.uint64)
        check(_ffi.lib().spal_mg_csr_windows(self._h, _p(lo), _p(hi)))
        return lo, hi

    def exchange_bytes(self) -> dict:
        x, y, h = u64(), u64(), u64()
        check(_ffi.lib().spal_mg_csr_exchange_bytes(self._h, C.byref(x), C.byref(y), C.byref(h)))
        return {"x_scatter": x.value, "y_gather": y.value, "halo": h.value}

    def spmv(self, x) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=self.dtype)
        y = np.empty(self.nrows, dtype=self.dtype)
        check(getattr(_ffi.lib(), f"spal_mg_csr_spmv_{_sfx(self.dtype)}")(self._h, _p(x), u64(x.size), _p(y), u64(y.size)))
        return y

    # ---- resident path (asynchronous on the context's streams until synchronize) ----
    def _call(self, name):
        check(getattr(_ffi.lib(), f"spal_mg_csr_{name}")(self._h))

    def set_x(self, x) -> None:
        """host vector -> GPU 0's x buffer (synchronous copy)"""
        x = np.ascontiguousarray(x, dtype=self.dtype)
        if x.size != self.ncols:
            raise Panic(_ffi.SPAL_ERR_INVALID_ARGUMENT, f"assertion failed: ncols == x.len() (left: {self.ncols}, right: {x.size})")
        ptr = vp()
        self.synchronize()
        check(_ffi.lib().spal_mg_csr_x_root(self._h, C.byref(ptr)))
        check(_ffi.lib().spal_memcpy_h2d(C.c_int(self.root_device), ptr, _p(x), C.c_size_t(x.nbytes)))

    def broadcast_x(self): self._call("broadcast_x")
    def scatter_x(self): self._call("scatter_x")
    def spmv_local(self): self._call("spmv_local")
    def gather_y(self): self._call("gather_y")
    def spmv_halo(self): self._call("spmv_halo")
    def spmv_resident(self): self._call("spmv_resident")
    def synchronize(self): self._call("synchronize")

    def y_gathered(self) -> np.ndarray:
        """GPU 0's y after gather_y (synchronises)"""
        self.synchronize()
        ptr = vp()
        check(_ffi.lib().spal_mg_csr_y_gathered(self._h, C.byref(ptr)))
        y = np.empty(self.nrows, dtype=self.dtype)
        check(_ffi.lib().spal_memcpy_d2h(C.c_int(self.root_device), _p(y), ptr, C.c_size_t(y.nbytes)))
        return y

    def y_allgathered(self) -> np.ndarray:
        """GPU 0's copy of the all-gathered y after spmv_resident (synchronises)"""
        self.synchronize()
        ptr, stride = vp(), u64()
        check(_ffi.lib().spal_mg_csr_y_root(self._h, C.byref(ptr), C.byref(stride)))
        pad = np.empty(self.ngpus * stride.value, dtype=self.dtype)
        check(_ffi.lib().spal_memcpy_d2h(C.c_int(self.root_device), _p(pad), ptr, C.c_size_t(pad.nbytes)))
        b = self.partition().astype(np.int64)
        return np.concatenate([pad[g * stride.value: g * stride.value + (b[g + 1] - b[g])] for g in range(self.ngpus)])

    def timing(self) -> dict:
        """HIP-event durations (ms, longest over the GPUs) of the last x distribution, local kernels,
        halo exchange and y collection; None for a phase that has not run.  Synchronises."""
        ms = (C.c_double * 4)()
        check(_ffi.lib().spal_mg_csr_timing(self._h, ms))
        return {k: (None if ms[i] < 0 else ms[i]) for i, k in enumerate(self.PHASES)}

    def close(self):
        if getattr(self, "_h", None):
            _ffi.lib().spal_mg_csr_destroy(self._h)
            self._h = None
        if getattr(self, "_ctx", None):
            _ffi.lib().spal_mg_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------
# CsrMatrix / CscMatrix
# --------------------------------------------------------------------------
class _Compressed:
    _kind = ""
    _dev_cls = _DeviceMatrix

    def __init__(self, nrows, ncols, ptr, ind, values):
        dt = _scalar_dtype(values)
        self._nrows, self._ncols = int(nrows), int(ncols)
        if self._nrows < 0 or self._ncols < 0:
            raise Panic(_ffi.SPAL_ERR_INVALID_ARGUMENT, "negative dimension")
        self._ptr, self._ind = _idx(ptr), _idx(ind)
        self._values = np.ascontiguousarray(values, dtype=dt)
        # the constructor's assertions, checked by the library's host code
        reason = C.c_int(0)
        check(getattr(_ffi.lib(), f"spal_{self._kind}_validate")(
            u64(self._nrows), u64(self._ncols), _p(self._ptr), u64(self._ptr.size),
            _p(self._ind), u64(self._ind.size), u64(self._values.size), C.byref(reason)))
        self._dev = {}

    @classmethod
    def _trusted(cls, nrows, ncols, ptr, ind, values):
        """struct-literal construction (what the reference's conversions do,
        e.g. src/csr/conv/coo.rs:108-114): no validation."""
        self = cls.__new__(cls)
        self._nrows, self._ncols = int(nrows), int(ncols)
        self._ptr, self._ind, self._values = ptr, ind, values
        self._dev = {}
        return self

    # accessors (src/csr.rs:200-289)
    def nrows(self) -> int:
        return self._nrows

    def ncols(self) -> int:
        return self._ncols

    def values(self) -> np.ndarray:
        return self._values

    def nnz(self) -> int:
        return int(self._ptr[-1])

    @property
    def dtype(self) -> np.dtype:
        return self._values.dtype

    def device(self, device: int = 0):
        """The device-resident copy (created on first use)."""
        h = self._dev.get(device)
        if h is None or h._h is None:
            out = vp()
            check(getattr(_ffi.lib(), f"spal_{self._kind}_create_{_sfx(self.dtype)}")(
                C.c_int(device), u64(self._nrows), u64(self._ncols), _p(self._ptr),
                u64(self._ptr.size), _p(self._ind), u64(self._ind.size), _p(self._values),
                u64(self._values.size), C.byref(out)))
            h = self._dev_cls(out, self.dtype, device)
            self._dev[device] = h
        return h

    def device_copy(self, device: int = 0):
        """ANOTHER device-resident copy with arrays of its own, not cached on the matrix (benchmarks rotate their
        launches over several, so that a matrix smaller than the 256 MB Infinity Cache is not served from it)."""
        out = vp()
        check(getattr(_ffi.lib(), f"spal_{self._kind}_create_{_sfx(self.dtype)}")(
            C.c_int(device), u64(self._nrows), u64(self._ncols), _p(self._ptr),
            u64(self._ptr.size), _p(self._ind), u64(self._ind.size), _p(self._values),
            u64(self._values.size), C.byref(out)))
        return self._dev_cls(out, self.dtype, device)

    def _mul_vec(self, x):
        x = np.asarray(x)
        if x.ndim != 1:
            raise TypeError("right-hand side must be a dense vector")
        if x.shape[0] != self._ncols:
            # assert_eq!(self.ncols(), rhs.nrows())  src/csr/ops/mul.rs:9
            raise Panic(_ffi.SPAL_ERR_INVALID_ARGUMENT,
                        f"assertion failed: ncols == x.len() (left: {self._ncols}, right: {x.shape[0]})")
        return self.device().spmv(x.astype(self.dtype, copy=False))

    def __mul__(self, x):
        return self._mul_vec(x)

    __matmul__ = __mul__


class CsrMatrix(_Compressed):
    """Compressed sparse row matrix (reference src/csr.rs:66-72)."""
    _kind = "csr"
    _dev_cls = DeviceCsr

    @classmethod
    def new(cls, nrows, ncols, rowptr, colind, values):
        return cls(nrows, ncols, rowptr, colind, values)

    @classmethod
    def eye(cls, size: int, dtype=np.float64):
        """src/csr.rs:179-189"""
        if not size > 0:
            raise Panic(_ffi.SPAL_ERR_INVARIANT, "assertion failed: size > 0")
        idx = np.arange(size + 1, dtype=np.uint64)
        return cls._trusted(size, size, idx, idx[:-1].copy(), np.ones(size, dtype=dtype))

    def rowptr(self) -> np.ndarray:
        return self._ptr

    def colind(self) -> np.ndarray:
        return self._ind

    def row_slice(self, row_begin: int, row_end: int) -> "CsrMatrix":
        """Rows [row_begin, row_end) as a CsrMatrix of their own (a row range of
        a valid CSR matrix is a valid CSR matrix after rebasing rowptr):
        the per-GPU shard of the row-partitioned product."""
        a0, a1 = int(self._ptr[row_begin]), int(self._ptr[row_end])
        return CsrMatrix._trusted(row_end - row_begin, self._ncols,
                                  self._ptr[row_begin:row_end + 1] - np.uint64(a0),
                                  self._ind[a0:a1], self._values[a0:a1])

    @classmethod
    def from_coo(cls, coo: "CooMatrix", device: int = 0) -> "CsrMatrix":
        """`CsrMatrix::from(&coo)` (src/csr/conv/coo.rs:3-116) on the device."""
        dev = coo.assemble_csr(device)
        rp, ci, va = dev.download()
        out = cls._trusted(coo.nrows(), coo.ncols(), rp, ci, va)
        out._dev[device] = dev
        return out

    @classmethod
    def from_csc(cls, csc: "CscMatrix", device: int = 0) -> "CsrMatrix":
        """`CsrMatrix::from(&csc)` (src/csr/conv/csc.rs:4-52) on the device."""
        dev = csc.device(device).to_csr()
        rp, ci, va = dev.download()
        out = cls._trusted(csc.nrows(), csc.ncols(), rp, ci, va)
        out._dev[device] = dev
        return out

    @classmethod
    def from_(cls, other, device: int = 0):
        return cls.from_csc(other, device) if isinstance(other, CscMatrix) else cls.from_coo(other, device)


class CscMatrix(_Compressed):
    """Compressed sparse column matrix (reference src/csc.rs:66-72)."""
    _kind = "csc"
    _dev_cls = DeviceCsc

    @classmethod
    def new(cls, nrows, ncols, colptr, rowind, values):
        return cls(nrows, ncols, colptr, rowind, values)

    def colptr(self) -> np.ndarray:
        return self._ptr

    def rowind(self) -> np.ndarray:
        return self._ind

    @classmethod
    def from_coo(cls, coo: "CooMatrix", device: int = 0) -> "CscMatrix":
        """`CscMatrix::from(&coo)` (src/csc/conv/coo.rs:3-116) on the device."""
        d = coo.upload(device)
        dev = d.assemble_csc()
        d.close()
        cp, ri, va = dev.download()
        out = cls._trusted(coo.nrows(), coo.ncols(), cp, ri, va)
        out._dev[device] = dev
        return out

    @classmethod
    def from_csr(cls, csr: "CsrMatrix", device: int = 0) -> "CscMatrix":
        """`CscMatrix::from(&csr)` (src/csc/conv/csr.rs:4-52) on the device."""
        dev = csr.device(device).to_csc()
        cp, ri, va = dev.download()
        out = cls._trusted(csr.nrows(), csr.ncols(), cp, ri, va)
        out._dev[device] = dev
        return out

    @classmethod
    def from_(cls, csr, device: int = 0):
        return cls.from_csr(csr, device)


# --------------------------------------------------------------------------
# CooMatrix (host container; reference src/coo.rs:53-57)
# --------------------------------------------------------------------------
class CooMatrix:
    """Coordinate format: insertion order is significant (duplicates are
    summed in that order by the conversion)."""

    def __init__(self, nrows: int, ncols: int, dtype=np.float64):
        if not nrows > 0:
            raise Panic(_ffi.SPAL_ERR_INVARIANT, "assertion failed: nrows > 0")   # src/coo.rs:105
        if not ncols > 0:
            raise Panic(_ffi.SPAL_ERR_INVARIANT, "assertion failed: ncols > 0")   # src/coo.rs:106
        self._nrows, self._ncols = int(nrows), int(ncols)
        self._dtype = np.dtype(dtype)
        self._rows = np.empty(0, dtype=np.uint64)
        self._cols = np.empty(0, dtype=np.uint64)
        self._vals = np.empty(0, dtype=self._dtype)
        self._pending = []  # pushed one by one since the last flush

    @classmethod
    def new(cls, nrows, ncols, dtype=np.float64):
        return cls(nrows, ncols, dtype)

    @classmethod
    def with_capacity(cls, nrows, ncols, capacity, dtype=np.float64):
        return cls(nrows, ncols, dtype)

    @classmethod
    def with_triplets(cls, nrows, ncols, rowind, colind, values):
        """src/coo.rs:260-288"""
        dt = _scalar_dtype(values)
        self = cls(nrows, ncols, dt)
        rows, cols = _idx(rowind), _idx(colind)
        vals = np.ascontiguousarray(values, dtype=dt)
        if rows.size != vals.size:
            raise Panic(_ffi.SPAL_ERR_INVARIANT, "assertion failed: rowind.len() == values.len()")
        if cols.size != vals.size:
            raise Panic(_ffi.SPAL_ERR_INVARIANT, "assertion failed: colind.len() == values.len()")
        if rows.size and int(rows.max()) >= self._nrows:
            raise Panic(_ffi.SPAL_ERR_INDEX_OUT_OF_BOUNDS, "assertion failed: *row < nrows (row < nrows)")
        if cols.size and int(cols.max()) >= self._ncols:
            raise Panic(_ffi.SPAL_ERR_INDEX_OUT_OF_BOUNDS, "assertion failed: *col < ncols (col < ncols)")
        self._rows, self._cols, self._vals = rows, cols, vals
        return self

    @classmethod
    def with_entries(cls, nrows, ncols, entries, dtype=np.float64):
        """src/coo.rs:204-221"""
        entries = list(entries)
        return cls.with_triplets(nrows, ncols, [e[0] for e in entries], [e[1] for e in entries],
                                 np.array([e[2] for e in entries], dtype=dtype))

    def push(self, row: int, col: int, value) -> None:
        """src/coo.rs:431-435"""
        if not (0 <= row < self._nrows):
            raise Panic(_ffi.SPAL_ERR_INDEX_OUT_OF_BOUNDS, "assertion failed: row < self.nrows")
        if not (0 <= col < self._ncols):
            raise Panic(_ffi.SPAL_ERR_INDEX_OUT_OF_BOUNDS, "assertion failed: col < self.ncols")
        self._pending.append((row, col, value))

    def _flush(self):
        if self._pending:
            r, c, v = zip(*self._pending)
            self._rows = np.concatenate([self._rows, np.array(r, dtype=np.uint64)])
            self._cols = np.concatenate([self._cols, np.array(c, dtype=np.uint64)])
            self._vals = np.concatenate([self._vals, np.array(v, dtype=self._dtype)])
            self._pending = []

    def nrows(self) -> int:
        return self._nrows

    def ncols(self) -> int:
        return self._ncols

    def length(self) -> int:
        return self._rows.size + len(self._pending)

    @property
    def dtype(self):
        return self._dtype

    def triplets(self):
        """(rows, cols, values) in insertion order (what `iter()` yields,
        src/coo.rs:491-495, unzipped)."""
        self._flush()
        return self._rows, self._cols, self._vals

    def iter(self):
        r, c, v = self.triplets()
        return zip(r.tolist(), c.tolist(), v.tolist())

    def upload(self, device: int = 0) -> DeviceCoo:
        r, c, v = self.triplets()
        out = vp()
        check(getattr(_ffi.lib(), f"spal_coo_upload_{_sfx(self._dtype)}")(
            C.c_int(device), u64(self._nrows), u64(self._ncols), u64(v.size), _p(r), _p(c), _p(v),
            C.byref(out)))
        return DeviceCoo(out, self._dtype, device, self._nrows, self._ncols, v.size)

    def assemble_csr(self, device: int = 0) -> DeviceCsr:
        r, c, v = self.triplets()
        out = vp()
        check(getattr(_ffi.lib(), f"spal_coo_to_csr_{_sfx(self._dtype)}")(
            C.c_int(device), u64(self._nrows), u64(self._ncols), u64(v.size), _p(r), _p(c), _p(v),
            C.byref(out)))
        return DeviceCsr(out, self._dtype, device)
