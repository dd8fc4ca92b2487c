#!/usr/bin/env python3
"""Host time per launch through the Python mirror (development tool): a tiny matrix, so the
loop is bound by the host, not by the kernel."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402

n = 2048
rp, ci, va = synth.banded_csr(n, n, 14, 512, 3)
dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
x = torch.from_numpy(synth.vector(n)).cuda()
y = torch.empty_like(x)
for name, fn in [("dev.spmv_torch", lambda: dev.spmv_torch(x, out=y)),
                 ("torch.add (reference point)", lambda: torch.add(x, x, out=y))]:
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5000):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:32s} {1e6 * (t1 - t0) / 5000:7.2f} us/call issued, {1e6 * (t2 - t0) / 5000:7.2f} us/call completed", flush=True)
