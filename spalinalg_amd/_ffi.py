"""ctypes binding of libspal_hip.so (C ABI: include/spal.h).

The library is built in-tree by `make -C spalinalg_amd/csrc` (see
__graft_entry__.build).  There is no fallback of any kind: if the shared
object is missing, or a compute entry point finds no HIP device, the call
raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPAL_HIP_LIB: developer override (A/B of differently built libraries, tools/)
LIB_PATH = os.environ.get("SPAL_HIP_LIB") or os.path.join(_HERE, "lib", "libspal_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "spal.h")

SPAL_OK = 0
SPAL_ERR_INVALID_ARGUMENT = 1
SPAL_ERR_INVARIANT = 2
SPAL_ERR_HIP = 3
SPAL_ERR_OUT_OF_MEMORY = 4
SPAL_ERR_UNSUPPORTED = 5
SPAL_ERR_NO_DEVICE = 6
SPAL_ERR_INDEX_OUT_OF_BOUNDS = 7


class SpalError(RuntimeError):
    """A libspal_hip call failed (HIP error, no device, out of memory ...)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"[spal status {status}] {message}")
        self.status = status


class Panic(AssertionError):
    """The reference would `panic!` here (a failed `assert!` in
    CsrMatrix::new, src/csr.rs:144-156, a dimension mismatch,
    src/csr/ops/mul.rs:9, an out-of-range COO entry, src/coo.rs:432-433).
    A Rust binding turns the same statuses into panic!()."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


_PANIC_STATUSES = {SPAL_ERR_INVALID_ARGUMENT, SPAL_ERR_INVARIANT, SPAL_ERR_INDEX_OUT_OF_BOUNDS}

_lib = None

u64 = C.c_uint64
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)
f32p = C.POINTER(C.c_float)
vp = C.c_void_p


def _preload_torch_hip():
    """torch ships its own libamdhip64 (SONAME libamdhip64.so.7).  If torch is
    going to be used in this process it must be loaded first so that
    libspal_hip.so binds to the same HIP runtime instead of a second copy."""
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C spalinalg_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    _preload_torch_hip()
    L = C.CDLL(LIB_PATH)
    L.spal_last_error.restype = C.c_char_p
    L.spal_version.restype = C.c_char_p
    for name in exported_names():
        fn = getattr(L, name)
        if name not in ("spal_last_error", "spal_version"):
            fn.restype = C.c_int
    _lib = L
    return L


def exported_names() -> list[str]:
    """Every function include/spal.h declares (parsed from the header)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spal_[a-z0-9_]+)\s*\(", text)))


def check(status: int) -> None:
    if status == SPAL_OK:
        return
    msg = lib().spal_last_error().decode("utf-8", "replace")
    if status in _PANIC_STATUSES:
        raise Panic(status, msg)
    raise SpalError(status, msg)


def device_count() -> int:
    n = C.c_int(0)
    check(lib().spal_device_count(C.byref(n)))
    return n.value


def cache_trim() -> None:
    """Hands the library's cached device blocks back to the driver (see spal_cache_trim in include/spal.h)."""
    check(lib().spal_cache_trim())
