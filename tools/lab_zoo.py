#!/usr/bin/env python3
"""A small zoo of sparsity patterns through the automatic plan (development tool): where are the cliffs?"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402


def timeit(fn, iters=20, warm=3):
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


def from_lens(lens, col_fn, rng):
    """rows with lens[r] entries, columns from col_fn(row index array (repeated), position in row, rng) then sorted / deduplicated per row"""
    n = lens.size
    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    pos = np.arange(rows.size, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
    cols = col_fn(rows, pos, rng)
    key = rows * (int(cols.max()) + 1) + cols
    key = np.unique(key)                      # sorts by (row, col), drops duplicates
    rows2, cols2 = key // (int(cols.max()) + 1), key % (int(cols.max()) + 1)
    lens2 = np.bincount(rows2, minlength=n)
    rp = np.concatenate([[0], np.cumsum(lens2)]).astype(np.uint64)
    return rp, cols2.astype(np.uint64), rng.uniform(-1, 1, cols2.size)


def main():
    rng = np.random.default_rng(5)
    cases = []
    n = 4_000_000
    cases.append(("diagonal (1/row)", n, lambda: from_lens(np.ones(n, np.int64), lambda r, p, g: r, rng)))
    cases.append(("tridiagonal (3/row)", n, lambda: from_lens(np.full(n, 3, np.int64), lambda r, p, g: np.clip(r + p - 1, 0, n - 1), rng)))
    cases.append(("band 14/row, W=4096 (config-3-like)", n, lambda: from_lens(np.full(n, 14, np.int64), lambda r, p, g: np.clip(r - 2048 + g.integers(0, 4096, r.size), 0, n - 1), rng)))
    cases.append(("band 30/row, W=1024", n, lambda: from_lens(np.full(n, 30, np.int64), lambda r, p, g: np.clip(r - 512 + g.integers(0, 1024, r.size), 0, n - 1), rng)))
    m = 1_000_000
    cases.append(("band 100/row, W=2048", m, lambda: from_lens(np.full(m, 100, np.int64), lambda r, p, g: np.clip(r - 1024 + g.integers(0, 2048, r.size), 0, m - 1), rng)))
    cases.append(("band 400/row, W=4096", 250_000, lambda: from_lens(np.full(250_000, 400, np.int64), lambda r, p, g: np.clip(r - 2048 + g.integers(0, 4096, r.size), 0, 249_999), rng)))
    # power-law row lengths (graph-like), columns uniform
    pl = np.minimum((rng.pareto(1.6, 2_000_000) * 6 + 1).astype(np.int64), 5000)
    cases.append(("power-law rows (mean %.1f, max %d), uniform columns" % (pl.mean(), pl.max()), pl.size, lambda: from_lens(pl, lambda r, p, g: g.integers(0, pl.size, r.size), rng)))
    # power-law rows, local columns
    cases.append(("power-law rows, columns within +-5000", pl.size, lambda: from_lens(pl, lambda r, p, g: np.clip(r - 5000 + g.integers(0, 10000, r.size), 0, pl.size - 1), rng)))
    # block diagonal, dense 64 x 64 blocks
    nb = 1_000_000
    cases.append(("block-diagonal, dense 64 x 64 blocks", nb, lambda: from_lens(np.full(nb, 64, np.int64), lambda r, p, g: (r // 64) * 64 + p, rng)))
    # two bands far apart (coupled systems)
    cases.append(("two bands 7 + 7 per row, 2M columns apart", n, lambda: from_lens(np.full(n, 14, np.int64), lambda r, p, g: np.clip(np.where(p < 7, r - 100 + g.integers(0, 200, r.size), (r + 2_000_000) % n - 100 + g.integers(0, 200, r.size)), 0, n - 1), rng)))
    for name, nrows, make in cases:
        t0 = time.time()
        rp, ci, va = make()
        ncols = nrows
        dev = sp.CsrMatrix(nrows, ncols, rp, ci, va).device()
        x = torch.from_numpy(synth.vector(ncols)).cuda()
        y = torch.empty(nrows, dtype=torch.float64, device="cuda")
        t = timeit(lambda: dev.spmv_torch(x, out=y))
        plan = dev.autotune(x, y, iters=10)
        t2 = timeit(lambda: dev.spmv_torch(x, out=y))
        nnz = int(rp[-1])
        B = synth.spmv_bytes(nnz, nrows, nrows, ncols, 8)
        d = dev.describe()
        print(f"{name:52s} nnz {nnz:>10d}  {t*1e3:8.1f} us -> autotuned {t2*1e3:8.1f} us = {100*B/(t2*1e-3)/8e12:5.1f} % of 8 TB/s  "
              f"[{d['kernel']} rpt={d['rows_per_tile']} stream={d['stream_row_fraction']:.2f} lds={d['lds_row_fraction']:.2f} "
              f"win={d['lds_window_bytes']//1024}K pers={d['persistent']}]  (host {time.time()-t0:.0f} s)", flush=True)
        del dev


if __name__ == "__main__":
    main()
