#!/usr/bin/env python3
"""Interleaved A/B timing of CSR SpMV plan variants in ONE process on ONE device
(N variants x M rounds, median and min reported): the only way to see
differences smaller than the run-to-run / box-to-box spread.  Development tool.

  python tools/lab_ab.py "kernel=2,tiles_per_wave=4" "kernel=2,tiles_per_wave=8"
"""
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402


def main():
    variants = [a for a in sys.argv[1:] if "=" in a and not a.startswith("@")]
    flags = [a for a in sys.argv[1:] if "=" not in a]
    shape = dict(kv[1:].split("=") for kv in sys.argv[1:] if kv.startswith("@"))   # @per_row=27 @window=8192 @rows=5000000
    n = int(shape.get("rows", 10_000_000))
    per_row = int(shape.get("per_row", 14))
    window = None if "uniform" in flags else int(shape.get("window", 4096))
    dtype = np.float32 if "f32" in flags else np.float64
    rounds, iters = 7, 25
    rp, ci, va = synth.banded_csr(n, n, per_row, window or n, synth.matrix_seed(3), dtype=dtype)
    devs = []
    for v in variants:  # one handle per variant: no re-planning inside the timed rounds
        d = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
        for kv in v.split(","):
            k, val = kv.split("=")
            d.set_option(k, int(val))
        devs.append(d)
    x = torch.from_numpy(synth.vector(n, dtype=dtype)).cuda()
    y = torch.empty_like(x)
    yref = devs[0].spmv_torch(x).clone()
    B = synth.spmv_bytes(n * per_row, n, n, n, np.dtype(dtype).itemsize)
    times = [[] for _ in variants]
    for r in range(rounds):
        for i, d in enumerate(devs):
            for _ in range(3):
                d.spmv_torch(x, out=y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                d.spmv_torch(x, out=y)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / iters * 1e3)
    for v, t, d in zip(variants, times, devs):
        ok = bool(torch.equal(d.spmv_torch(x), yref))
        pl = d.describe()
        v = f"{v} [{pl['kernel']} rpt={pl.get('rows_per_tile')} stream={pl['stream_row_fraction']}]"
        med, mn = statistics.median(t), min(t)
        print(f"{v:72s} median {med:7.1f} us  min {mn:7.1f} us  {B/med/1e3:7.1f} GB/s ({100*B/med/1e3/8000:5.2f} %)  "
              f"bit-equal-to-first={ok}  rounds={[round(q) for q in t]}", flush=True)


if __name__ == "__main__":
    main()
