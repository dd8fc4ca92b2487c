"""ctypes front end of the CPU oracle (oracle/spal_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of spal_oracle.c.  Only tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``spalinalg_amd``) never does.

Every wrapper takes / returns numpy arrays with the reference's own types:
indices ``uint64`` (Rust ``usize``), scalars ``float64`` / ``float32``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_U64P = C.POINTER(C.c_uint64)
_U32P = C.POINTER(C.c_uint32)


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (oracle/Makefile).  Returns its path."""
    src = os.path.join(_HERE, "spal_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_compressed_validate.restype = C.c_int
        for name in ("orc_csc_spmv_f64", "orc_csc_spmv_f32",
                     "orc_transpose_f64", "orc_transpose_f32",
                     "orc_from_coo_f64", "orc_from_coo_f32",
                     "orc_spgemm_f64", "orc_spgemm_f32"):
            getattr(_lib, name).restype = C.c_int
        _lib.orc_free.restype = None
    return _lib


def _u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _p(a: np.ndarray, ptype=None):
    if ptype is None:
        ptype = {np.dtype(np.uint64): _U64P, np.dtype(np.uint32): _U32P,
                 np.dtype(np.float64): C.POINTER(C.c_double),
                 np.dtype(np.float32): C.POINTER(C.c_float)}[a.dtype]
    return a.ctypes.data_as(ptype)


def _sfx(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64"
    if dtype == np.float32:
        return "f32"
    raise TypeError("Scalar is implemented for f32 and f64 only (src/scalar.rs:56-57)")


# --------------------------------------------------------------------------
# validation  (CsrMatrix::new / CscMatrix::new, src/csr.rs:144-156)
# --------------------------------------------------------------------------
VALIDATE_REASONS = {
    0: "ok",
    1: "nrows > 0",
    2: "ncols > 0",
    3: "ptr.len() == nmajor + 1",
    4: "ptr[0] == 0",
    5: "ind.len() == ptr[nmajor]",
    6: "values.len() == ptr[nmajor]",
    7: "ptr is non-decreasing",
    8: "every index < nminor",
    9: "indices strictly increasing inside each major slice",
}


def validate(nrows, ncols, ptr, ind, nvalues, *, csr=True) -> int:
    ptr, ind = _u64(ptr), _u64(ind)
    # the C code dereferences ptr only after `ptr_len == nmajor + 1` passed
    return int(lib().orc_compressed_validate(
        C.c_uint64(nrows), C.c_uint64(ncols), C.c_int(1 if csr else 0),
        _p(ptr), C.c_uint64(ptr.size), _p(ind), C.c_uint64(ind.size),
        C.c_uint64(nvalues)))


# --------------------------------------------------------------------------
# SpMV
# --------------------------------------------------------------------------
def csr_spmv(rowptr, colind, values, x) -> np.ndarray:
    values = np.ascontiguousarray(values)
    sfx = _sfx(values.dtype)
    rowptr, colind = _u64(rowptr), _u64(colind)
    x = np.ascontiguousarray(x, dtype=values.dtype)
    nrows = rowptr.size - 1
    y = np.empty(nrows, dtype=values.dtype)
    getattr(lib(), f"orc_csr_spmv_{sfx}")(
        C.c_uint64(nrows), _p(rowptr), _p(colind), _p(values), _p(x), _p(y))
    return y


def csr_spmv_idx32(rowptr32, colind32, values, x, y=None) -> np.ndarray:
    """Same loop on 32-bit indices; bench.py's cpu_baseline leg."""
    values = np.ascontiguousarray(values)
    sfx = _sfx(values.dtype)
    assert rowptr32.dtype == np.uint32 and colind32.dtype == np.uint32
    nrows = rowptr32.size - 1
    if y is None:
        y = np.empty(nrows, dtype=values.dtype)
    getattr(lib(), f"orc_csr_spmv_idx32_{sfx}")(
        C.c_uint64(nrows), _p(rowptr32), _p(colind32), _p(values), _p(x), _p(y))
    return y


def csr_spmv_idx32_threads(rowptr32, colind32, values, x, y, threads: int, pool=None) -> np.ndarray:
    """The idx32 loop, one contiguous row range per thread (ctypes releases the
    GIL); bench.py's informational all-cores figure.  Same results as the
    sequential call: rows are independent."""
    from concurrent.futures import ThreadPoolExecutor
    values = np.ascontiguousarray(values)
    fn = getattr(lib(), f"orc_csr_spmv_idx32_rows_{_sfx(values.dtype)}")
    fn.restype = None
    nrows = rowptr32.size - 1
    # ranges balanced by stored entries
    cuts = np.searchsorted(rowptr32, np.linspace(0, int(rowptr32[-1]), threads + 1)[1:-1]).tolist()
    bounds = [0] + [int(c) for c in cuts] + [nrows]
    args = (_p(rowptr32), _p(colind32), _p(values), _p(x), _p(y))

    def run(i):
        fn(C.c_uint64(bounds[i]), C.c_uint64(bounds[i + 1]), *args)

    own = pool is None
    pool = pool or ThreadPoolExecutor(max_workers=threads)
    try:
        list(pool.map(run, range(threads)))
    finally:
        if own:
            pool.shutdown()
    return y


def csc_spmv(nrows, colptr, rowind, values, x) -> np.ndarray:
    values = np.ascontiguousarray(values)
    sfx = _sfx(values.dtype)
    colptr, rowind = _u64(colptr), _u64(rowind)
    x = np.ascontiguousarray(x, dtype=values.dtype)
    ncols = colptr.size - 1
    y = np.empty(nrows, dtype=values.dtype)
    rc = getattr(lib(), f"orc_csc_spmv_{sfx}")(
        C.c_uint64(nrows), C.c_uint64(ncols), _p(colptr), _p(rowind),
        _p(values), _p(x), _p(y))
    if rc:
        raise MemoryError
    return y


def csr_abs_bound(rowptr, colind, values, x) -> np.ndarray:
    """bound[i] = sum_k |A_ik||x_k| (f64), the componentwise parity scale."""
    rowptr, colind = _u64(rowptr), _u64(colind)
    values = np.ascontiguousarray(values, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(rowptr.size - 1, dtype=np.float64)
    lib().orc_csr_abs_bound_f64(C.c_uint64(out.size), _p(rowptr), _p(colind),
                                _p(values), _p(x), _p(out))
    return out


# --------------------------------------------------------------------------
# transpose / CSR<->CSC  (counting sort, src/csr.rs:358-406)
# --------------------------------------------------------------------------
def transpose(nmajor, nminor, ptr, ind, values):
    """(ptr, ind, values) compressed by major -> compressed by minor."""
    values = np.ascontiguousarray(values)
    sfx = _sfx(values.dtype)
    ptr, ind = _u64(ptr), _u64(ind)
    nnz = int(ptr[-1])
    optr = np.empty(nminor + 1, dtype=np.uint64)
    oind = np.empty(nnz, dtype=np.uint64)
    oval = np.empty(nnz, dtype=values.dtype)
    rc = getattr(lib(), f"orc_transpose_{sfx}")(
        C.c_uint64(nmajor), C.c_uint64(nminor), _p(ptr), _p(ind), _p(values),
        _p(optr), _p(oind), _p(oval))
    if rc:
        raise MemoryError
    return optr, oind, oval


# --------------------------------------------------------------------------
# COO -> CSR / CSC  (src/csr/conv/coo.rs:4-115, src/csc/conv/coo.rs:4-115)
# --------------------------------------------------------------------------
def _from_coo(nmajor, nminor, maj, mino, vals):
    vals = np.ascontiguousarray(vals)
    sfx = _sfx(vals.dtype)
    maj, mino = _u64(maj), _u64(mino)
    n = vals.size
    optr = np.empty(nmajor + 1, dtype=np.uint64)
    oind = np.empty(max(n, 1), dtype=np.uint64)
    oval = np.empty(max(n, 1), dtype=vals.dtype)
    onnz = C.c_uint64(0)
    rc = getattr(lib(), f"orc_from_coo_{sfx}")(
        C.c_uint64(nmajor), C.c_uint64(nminor), C.c_uint64(n), _p(maj),
        _p(mino), _p(vals), _p(optr), _p(oind), _p(oval), C.byref(onnz))
    if rc:
        raise MemoryError
    k = int(onnz.value)
    return optr, oind[:k].copy(), oval[:k].copy()


def coo_to_csr(nrows, ncols, rows, cols, vals):
    return _from_coo(nrows, ncols, rows, cols, vals)


def coo_to_csc(nrows, ncols, rows, cols, vals):
    return _from_coo(ncols, nrows, cols, rows, vals)


# --------------------------------------------------------------------------
# the reference's literal sparse x sparse Mul
# --------------------------------------------------------------------------
def _spgemm(l_minor, l_major, lptr, lind, lval, r_minor, r_major, rptr, rind, rval):
    lval = np.ascontiguousarray(lval)
    sfx = _sfx(lval.dtype)
    rval = np.ascontiguousarray(rval, dtype=lval.dtype)
    lptr, lind, rptr, rind = _u64(lptr), _u64(lind), _u64(rptr), _u64(rind)
    optr, oind = _U64P(), _U64P()
    vt = C.c_double if sfx == "f64" else C.c_float
    oval = C.POINTER(vt)()
    rc = getattr(lib(), f"orc_spgemm_{sfx}")(
        C.c_uint64(l_minor), C.c_uint64(l_major), _p(lptr), _p(lind), _p(lval),
        C.c_uint64(r_minor), C.c_uint64(r_major), _p(rptr), _p(rind), _p(rval),
        C.byref(optr), C.byref(oind), C.byref(oval))
    if rc == -2:
        raise ValueError("dimension mismatch (reference: assert_eq! panics, mul.rs:9)")
    if rc:
        raise MemoryError
    try:
        p = np.ctypeslib.as_array(optr, shape=(r_major + 1,)).copy()
        nnz = int(p[-1])
        i = np.ctypeslib.as_array(oind, shape=(max(nnz, 1),))[:nnz].copy()
        v = np.ctypeslib.as_array(oval, shape=(max(nnz, 1),))[:nnz].copy()
    finally:
        lib().orc_free(optr)
        lib().orc_free(oind)
        lib().orc_free(oval)
    return p, i, v


def csc_mul(a_shape, a, b_shape, b):
    """CSC `&a * &b` (src/csc/ops/mul.rs:8-60). a, b = (colptr, rowind, values)."""
    (ar, ac), (br, bc) = a_shape, b_shape
    return _spgemm(ar, ac, *a, br, bc, *b)


def csr_mul(a_shape, a, b_shape, b):
    """CSR `&a * &b` (src/csr/ops/mul.rs:8-59). a, b = (rowptr, colind, values)."""
    (ar, ac), (br, bc) = a_shape, b_shape
    # CSR arrays of M are the CSC arrays of M^T; (AB)^T = B^T A^T.
    return _spgemm(bc, br, *b, ac, ar, *a)
