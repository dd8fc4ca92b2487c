#!/usr/bin/env python3
"""Config 3 f64, plain and persistent form, for A/B runs of differently built libraries
(SPAL_HIP_LIB); development tool."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


n = 10_000_000
dt = np.float32 if "f32" in sys.argv[1:] else np.float64
rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(3), dtype=dt)
dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
x = torch.from_numpy(synth.vector(n, dtype=dt)).cuda()
y = torch.empty_like(x)
for _ in range(150):
    dev.spmv_torch(x, out=y)
out = []
forms = [("plain", [("tiles_per_wave", 4), ("persistent", 0)]), ("pers", [("tiles_per_wave", 4), ("persistent", 1)])]
if "tpw8" in sys.argv[1:]:
    forms += [("plain8", [("tiles_per_wave", 8), ("persistent", 0)]), ("pers8", [("tiles_per_wave", 8), ("persistent", 1)])]
for rnd in range(3):
    for name, opts in forms:
        for k, v in opts:
            dev.set_option(k, v)
        timeit(lambda: dev.spmv_torch(x, out=y), 20)
        out.append(f"{name} {timeit(lambda: dev.spmv_torch(x, out=y), 150):6.1f}")
print(os.environ.get("SPAL_HIP_LIB", "main").split("/")[-2] if os.environ.get("SPAL_HIP_LIB") else "main", " | ".join(out), flush=True)
