#!/usr/bin/env python3
"""Stencil matrices (development tool): the columns of a row cluster in a few narrow bands far
apart -- the x 'window' of a super-tile spans far more than LDS holds although few distinct
columns are touched."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402


def stencil_csr(m, points):
    """m^3 grid, 7-point (faces) or 27-point (faces, edges, corners) stencil, row-major numbering."""
    n = m ** 3
    idx = np.arange(n, dtype=np.int64)
    i, j, k = idx // (m * m), (idx // m) % m, idx % m
    offs = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)
            if points == 27 or abs(a) + abs(b) + abs(c) <= 1]
    cols, valid = [], []
    for a, b, c in offs:                       # ascending column order by construction
        ok = (i + a >= 0) & (i + a < m) & (j + b >= 0) & (j + b < m) & (k + c >= 0) & (k + c < m)
        cols.append(idx + a * m * m + b * m + c)
        valid.append(ok)
    cols, valid = np.stack(cols, 1), np.stack(valid, 1)
    lens = valid.sum(1)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ci = cols[valid].astype(np.uint64)
    rng = np.random.default_rng(3)
    va = rng.uniform(-1, 1, ci.size)
    return n, rp, ci, va


def timeit(fn, iters, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    for m, points in ((216, 7), (150, 27)):
        n, rp, ci, va = stencil_csr(m, points)
        dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
        x = torch.from_numpy(synth.vector(n)).cuda()
        y = torch.empty_like(x)
        nnz = int(rp[-1])
        B = synth.spmv_bytes(nnz, n, n, n, 8)
        for opts in ((), (("persistent", 1),), (("kernel", 1),)):
            for k, v in opts:
                dev.set_option(k, v)
            t = timeit(lambda: dev.spmv_torch(x, out=y), 30)
            d = dev.describe()
            print(f"{m}^3 {points}-point  n={n} nnz={nnz} opts={opts}: {t*1e3:8.1f} us  {B/t/1e6:8.1f} GB/s = {100*B/t/1e6/8000:5.1f} % "
                  f"[{d['kernel']} stream={d['stream_row_fraction']} lds_rows={d['lds_row_fraction']} rpt={d['rows_per_tile']}]", flush=True)


if __name__ == "__main__":
    main()
