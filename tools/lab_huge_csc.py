#!/usr/bin/env python3
"""CSR -> CSC on the device and both CSC routes at 1.4e9 entries (development check)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rp, ci, va = synth.banded_csr(n, n, 14, 4096, 7)
csr = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
x = torch.from_numpy(synth.vector(n)).cuda()
y_csr = csr.spmv_torch(x)
t0 = time.time()
csc = csr.to_csc()
torch.cuda.synchronize()
print(f"{n} x {n}, {int(rp[-1])} entries: CSR -> CSC on the device in {time.time() - t0:.2f} s", flush=True)
bound = None
for kernel in (2, 1):
    t0 = time.time()
    csc.set_option("kernel", kernel)
    y = csc.spmv_torch(x)
    torch.cuda.synchronize()
    first = time.time() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        csc.spmv_torch(x, out=y)
    e1.record()
    torch.cuda.synchronize()
    diff = float((y - y_csr).abs().max())
    print(f"kernel {kernel}: first call {first:.2f} s, then {e0.elapsed_time(e1) / 5:.2f} ms per product; max |y - y_csr| = {diff:.3e} "
          f"({'bit-identical' if bool(torch.equal(y, y_csr)) else 'within rounding' if diff < 1e-12 else 'WRONG'}); {csc.describe()}", flush=True)
back = csc.to_csr()
rp2, ci2, va2 = back.download()
print("CSC -> CSR returns the original arrays:", bool(np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(va2, va)), flush=True)
