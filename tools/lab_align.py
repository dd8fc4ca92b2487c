#!/usr/bin/env python3
"""Does the placement of x / y relative to the matrix arrays change the kernel time?
(development tool; rocprof showed 266 us into one y buffer and 293 us into another in the same process)"""
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


def main():
    n = 10_000_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(3))
    dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        dev.set_option(k, int(v))
    xh = torch.from_numpy(synth.vector(n))
    slack = 1 << 21
    X = torch.zeros(n + slack, dtype=torch.float64, device="cuda")
    Y = torch.zeros(n + slack, dtype=torch.float64, device="cuda")
    Y2 = torch.zeros(n + slack, dtype=torch.float64, device="cuda")
    print("ptrs", hex(X.data_ptr()), hex(Y.data_ptr()), hex(Y2.data_ptr()), dev.describe(), flush=True)
    offs = [0, 16, 32, 64, 256, 512, 1024, 4096, 16384, 65536, 262144, 1 << 20]
    X[:n].copy_(xh)
    for _ in range(60):
        dev.spmv_torch(X[:n], out=Y[:n])
    torch.cuda.synchronize()
    res = {("y", o): [] for o in offs}
    res.update({("y2", o): [] for o in offs})
    res.update({("x", o): [] for o in offs})
    for rnd in range(3):
        for o in offs:
            res[("y", o)].append(timeit(lambda: dev.spmv_torch(X[:n], out=Y[o:o + n]), 20))
            res[("y2", o)].append(timeit(lambda: dev.spmv_torch(X[:n], out=Y2[o:o + n]), 20))
        for o in offs:
            X[o:o + n].copy_(xh)
            res[("x", o)].append(timeit(lambda: dev.spmv_torch(X[o:o + n], out=Y[:n]), 20))
        X[:n].copy_(xh)
    for key, v in res.items():
        print(f"{key[0]:3s} offset {key[1]:8d} elements: " + "  ".join(f"{t:7.1f}" for t in v) + f"   min {min(v):7.1f} us", flush=True)


if __name__ == "__main__":
    main()
