"""spalinalg_amd -- MI355X (gfx950) SpMV / assembly path for spalinalg.

Host-side mirror of the reference's public types (`CsrMatrix`, `CscMatrix`,
`CooMatrix`; reference src/lib.rs:16-19) over the C ABI of libspal_hip.so
(include/spal.h).  The compute path is hand-written HIP only; there is no
CPU fallback.
"""
from ._ffi import Panic, SpalError, cache_trim, device_count  # noqa: F401
from .matrix import CooMatrix, CscMatrix, CsrMatrix, DeviceCoo, DeviceCsr, DeviceCsc, MultiGpuCsr  # noqa: F401

__all__ = ["CsrMatrix", "CscMatrix", "CooMatrix", "DeviceCsr", "DeviceCsc", "DeviceCoo", "MultiGpuCsr", "Panic",
           "SpalError", "device_count", "cache_trim"]
