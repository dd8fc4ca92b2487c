"""Synthetic inputs of BASELINE.json's configs and the
algorithmic byte / flop counts the roofline figures are built from.

Bench / test support, NOT part of the product: the generators live in their own
host-only library (spal_synth/libspal_synth.so, built by `make -C spal_synth`;
SplitMix64 based, bit-exact across implementations -- tests/test_synth.py
restates them in pure Python).  libspal_hip.so does not contain them.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspal_synth.so")
SEED_X = 0xC0FFEE
HBM_PEAK_BYTES_PER_S = 8.0e12  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

_lib = None
u64 = C.c_uint64


def build() -> None:
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "spal_synth.cpp")
        if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.spal_synth_last_error.restype = C.c_char_p
    return _lib


def _check(status: int) -> None:
    if status:
        raise ValueError(lib().spal_synth_last_error().decode())


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _sfx(dt: np.dtype) -> str:
    if dt == np.float64:
        return "f64"
    if dt == np.float32:
        return "f32"
    raise TypeError("Scalar is implemented for f32 and f64 only (src/scalar.rs:56-57)")


def matrix_seed(cfg: int) -> int:
    return 0x5EED0000 + cfg


def banded_csr(nrows, ncols, per_row, window, seed, dtype=np.float64, rows=None):
    """Exactly `per_row` distinct sorted columns per row, drawn from a window of
    `window` columns centred on the diagonal (window == ncols: uniform).
    rows=(begin, end) generates only that row range (rowptr rebased to 0)."""
    dtype = np.dtype(dtype)
    r0, r1 = (0, nrows) if rows is None else rows
    n = r1 - r0
    rp = np.empty(n + 1, dtype=np.uint64)
    ci = np.empty(n * per_row, dtype=np.uint64)
    va = np.empty(n * per_row, dtype=dtype)
    _check(getattr(lib(), f"spal_synth_banded_csr_rows_{_sfx(dtype)}")(
        u64(nrows), u64(ncols), C.c_uint32(per_row), u64(window), u64(seed), u64(r0), u64(r1),
        _p(rp), _p(ci), _p(va)))
    return rp, ci, va


def ragged_csr(nrows, ncols, window, seed, dtype=np.float64, rows=None):
    """The robustness variant: row length 1 + next() % 27 (mean 14) from the
    first draw of the row's stream, then columns and values as banded_csr."""
    dtype = np.dtype(dtype)
    r0, r1 = (0, nrows) if rows is None else rows
    rp = np.empty(r1 - r0 + 1, dtype=np.uint64)
    _check(lib().spal_synth_ragged_rowptr(u64(nrows), u64(seed), u64(r0), u64(r1), _p(rp)))
    nnz = int(rp[-1])
    ci = np.empty(nnz, dtype=np.uint64)
    va = np.empty(nnz, dtype=dtype)
    _check(getattr(lib(), f"spal_synth_ragged_fill_{_sfx(dtype)}")(
        u64(nrows), u64(ncols), u64(window), u64(seed), u64(r0), u64(r1), _p(rp), _p(ci), _p(va)))
    return rp, ci, va


def ragged_rowptr(nrows, seed, rows=None):
    """rowptr of ragged_csr alone (for partitioning before any entry is generated)"""
    r0, r1 = (0, nrows) if rows is None else rows
    rp = np.empty(r1 - r0 + 1, dtype=np.uint64)
    _check(lib().spal_synth_ragged_rowptr(u64(nrows), u64(seed), u64(r0), u64(r1), _p(rp)))
    return rp


def vector(n, seed=SEED_X, dtype=np.float64):
    dtype = np.dtype(dtype)
    x = np.empty(n, dtype=dtype)
    _check(getattr(lib(), f"spal_synth_vector_{_sfx(dtype)}")(u64(n), u64(seed), _p(x)))
    return x


def coo(nrows, ncols, length, seed, dup_permille=0, cancel_permille=0, dtype=np.float64):
    dtype = np.dtype(dtype)
    r = np.empty(length, dtype=np.uint64)
    c = np.empty(length, dtype=np.uint64)
    v = np.empty(length, dtype=dtype)
    _check(getattr(lib(), f"spal_synth_coo_{_sfx(dtype)}")(
        u64(nrows), u64(ncols), u64(length), u64(seed), C.c_uint32(dup_permille),
        C.c_uint32(cancel_permille), _p(r), _p(c), _p(v)))
    return r, c, v


# BASELINE.json configs as concrete inputs
CONFIGS = {
    1: dict(kind="coo->csr", nrows=10_000, ncols=10_000, length=100_000, dup_permille=0,
            cancel_permille=0, note="CPU-runnable reference case"),
    2: dict(kind="csr", nrows=1_000_000, ncols=1_000_000, per_row=14, window=4096),
    3: dict(kind="csr", nrows=10_000_000, ncols=10_000_000, per_row=14, window=4096),
    4: dict(kind="csc", nrows=1_000_000, ncols=1_000_000, per_row=14, window=4096),
    5: dict(kind="coo->csr", nrows=5_000_000, ncols=5_000_000, length=50_000_000,
            dup_permille=10, cancel_permille=1),
}


def spmv_bytes(nnz, nptr, nrows, ncols, elem_size):
    """Algorithmic bytes of one SpMV: values + 32-bit
    indices once, 32-bit pointer array once, x once, y once."""
    return nnz * (elem_size + 4) + 4 * (nptr + 1) + elem_size * ncols + elem_size * nrows


def spmv_flops(nnz):
    return 2 * nnz


def assembly_bytes(length, nnz_out, nrows, elem_size=8):
    """Lower bound for COO->CSR: (u32, u32, T) in, CSR out."""
    return length * (8 + elem_size) + nnz_out * (4 + elem_size) + 4 * (nrows + 1)
