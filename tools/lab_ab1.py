#!/usr/bin/env python3
"""Interleaved A/B timing of plan variants on ONE handle (one placement of the arrays in device memory:
handles of the same matrix differ by up to 8 % with identical code, tools/lab_place.py), the options of
each variant re-applied before its rounds.  Development tool.

  python tools/lab_ab1.py "prefetch=1" "prefetch=2" [@rows=... @per_row=... @window=...] [f32] [uniform] [ragged]
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
    shape = dict(kv[1:].split("=") for kv in sys.argv[1:] if kv.startswith("@"))
    n = int(shape.get("rows", 10_000_000))
    per_row = int(shape.get("per_row", 14))
    window = n if "uniform" in flags else int(shape.get("window", 4096))
    dtype = np.float32 if "f32" in flags else np.float64
    rounds, iters = int(shape.get("rounds", 5)), 25
    if "ragged" in flags:
        rp, ci, va = synth.ragged_csr(n, n, window, synth.matrix_seed(3), dtype=dtype)
    else:
        rp, ci, va = synth.banded_csr(n, n, per_row, window, synth.matrix_seed(3), dtype=dtype)
    nnz = int(rp[-1])
    d = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    x = torch.from_numpy(synth.vector(n, dtype=dtype)).cuda()
    y = torch.empty_like(x)
    yref = d.spmv_torch(x).clone()
    B = synth.spmv_bytes(nnz, n, n, n, np.dtype(dtype).itemsize)
    times = [[] for _ in variants]
    equal, plans = [None] * len(variants), [None] * len(variants)
    defaults = {"slide": -1, "slide_on": 1, "uniform_rows": 1, "prefetch": 1, "persistent": 0, "nt_store": 0, "diag": 0,
                "tiles_per_wave": 4, "rows_per_tile": 0, "persistent_blocks": 0, "slide_run": 0, "panel_on": 1, "panel_pages": 192, "panel_window": 0, "kernel": 0, "skew": -1, "window_pages": 0, "stream_global": 1}
    named = {kv.split("=")[0] for v in variants for kv in v.split(",")}
    for r in range(rounds):
        for i, v in enumerate(variants):
            opts = {k: defaults[k] for k in named if k in defaults}   # every option any variant names, back to its default
            opts.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in v.split(",")})
            for k, val in opts.items():
                d.set_option(k, val)
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
            equal[i] = bool(torch.equal(y, yref))
            plans[i] = d.describe()
    for v, t, ok, pl in zip(variants, times, equal, plans):
        tag = f"{v} [{pl['kernel']} rpt={pl.get('rows_per_tile')} uni={pl.get('uniform_row_fraction')} pf={pl.get('prefetch')} pers={pl.get('persistent')}]"
        med, mn = statistics.median(t), min(t)
        print(f"{tag:84s} median {med:7.1f} us  min {mn:7.1f} us  {B/med/1e3:7.1f} GB/s alg ({100*B/med/1e3/8000:5.2f} %)  "
              f"bit-equal-to-first={ok}  rounds={[round(q) for q in t]}", flush=True)


if __name__ == "__main__":
    main()
