#!/usr/bin/env python3
"""Rows of 1 ... 5 entries (diagonal, tridiagonal, 5-point): which kernel / geometry (development tool)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402
from tools.lab_zoo import from_lens, timeit  # noqa: E402


def main():
    rng = np.random.default_rng(5)
    n = 4_000_000
    cases = [("diagonal (1/row)", np.ones(n, np.int64), lambda r, p, g: r),
             ("tridiagonal (3/row)", np.full(n, 3, np.int64), lambda r, p, g: np.clip(r + p - 1, 0, n - 1)),
             ("5-point 2000x2000", np.full(n, 5, np.int64), lambda r, p, g: np.clip(r + np.array([-2000, -1, 0, 1, 2000])[p], 0, n - 1))]
    for name, lens, fn in cases:
        rp, ci, va = from_lens(lens, fn, rng)
        nnz = int(rp[-1])
        B = synth.spmv_bytes(nnz, n, n, n, 8)
        dev = sp.CsrMatrix(n, n, rp, ci, va).device()
        x = torch.from_numpy(synth.vector(n)).cuda()
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        print(f"{name}: nnz {nnz}, floor {B/8e12*1e6:.1f} us", flush=True)
        variants = [[("kernel", 0)], [("kernel", 2), ("rows_per_tile", 64), ("persistent", 0)], [("kernel", 2), ("rows_per_tile", 128), ("persistent", 0)], [("kernel", 2), ("rows_per_tile", 256), ("persistent", 0)],
                    [("kernel", 2), ("rows_per_tile", 128), ("persistent", 1)], [("rows_per_tile", 0), ("kernel", 2), ("persistent", 1)], [("kernel", 2), ("persistent", 0), ("tiles_per_wave", 8)],
                    [("tiles_per_wave", 4), ("kernel", 1), ("lanes_per_row", 2), ("unroll", 4)],
                    [("kernel", 1), ("lanes_per_row", 4), ("unroll", 4)],
                    [("kernel", 1), ("lanes_per_row", 2), ("unroll", 4), ("rows_per_block", 4096)],
                    [("kernel", 1), ("lanes_per_row", 2), ("unroll", 4), ("rows_per_block", 2048), ("lds_x", 0)],
                    [("kernel", 1), ("lanes_per_row", 4), ("unroll", 2), ("rows_per_block", 2048)]]
        for opts in variants:
            try:
                for k, v in opts:
                    dev.set_option(k, v)
            except Exception as e:  # noqa: BLE001
                print(f"  {str(dict(opts)):90s} rejected: {e}")
                continue
            t = timeit(lambda: dev.spmv_torch(x, out=y))
            d = dev.describe()
            print(f"  {str(dict(opts)):90s} {t*1e3:7.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %  [{d['kernel']} rpt={d['rows_per_tile']} L={d['lanes_per_row']} U={d['unroll']} "
                  f"R={d['rows_per_block']} lds={d['lds_x']} pers={d['persistent']}]", flush=True)
        del dev


if __name__ == "__main__":
    main()
