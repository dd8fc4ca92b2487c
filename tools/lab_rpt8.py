#!/usr/bin/env python3
"""Rows of 56 ... 128 entries: the stream kernel with 8-row tiles against the vector kernel (development tool)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402
from tools.lab_zoo import timeit  # noqa: E402


def main():
    sizes = ((54, 2_000_000), (64, 1_500_000), (81, 1_200_000), (100, 1_000_000), (120, 800_000))
    if "long" in sys.argv[1:]:
        sizes = ((150, 640_000), (200, 480_000), (250, 400_000))
    if "pow2" in sys.argv[1:]:
        sizes = tuple((k, 3_000_000 if k <= 17 else 1_500_000 if k < 70 else 1_000_000) for k in (8, 15, 16, 24, 32, 48, 63, 64, 65, 81, 96, 100, 120))
    if "mid" in sys.argv[1:]:
        sizes = tuple((k, 3_000_000 if k <= 20 else 1_500_000 if k < 70 else 1_000_000) for k in (17, 20, 24, 33, 40, 48, 70, 81, 100))
    for per_row, n in sizes:
        rp, ci, va = synth.banded_csr(n, n, per_row, 2048, 7)
        nnz = int(rp[-1])
        B = synth.spmv_bytes(nnz, n, n, n, 8)
        dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
        x = torch.from_numpy(synth.vector(n)).cuda()
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        print(f"band {per_row}/row, W=2048, {n} rows: nnz {nnz}, floor {B/8e12*1e6:.1f} us", flush=True)
        long_opts = ([("kernel", 0)], [("kernel", 2), ("stream_row_max", 256), ("rows_per_tile", 4), ("persistent", 0)],
                     [("kernel", 2), ("stream_row_max", 256), ("rows_per_tile", 8), ("persistent", 0)])
        mid_opts = ([("kernel", 0)], [("kernel", 0), ("persistent", 0)]) + tuple([("kernel", 2), ("rows_per_tile", r), ("persistent", 0)] for r in (64, 48, 32, 24, 16, 12, 8) if 128 <= r * per_row <= 1024)
        pow2_opts = ([("kernel", 0)], [("kernel", 2), ("persistent", 0), ("skew", 0)], [("skew", 1)], [("skew", -1), ("kernel", 1)])
        for opts in mid_opts if "mid" in sys.argv[1:] else pow2_opts if "pow2" in sys.argv[1:] else long_opts if "long" in sys.argv[1:] else ([("kernel", 0)], [("kernel", 2), ("rows_per_tile", 16), ("persistent", 0)],
                     [("kernel", 2), ("rows_per_tile", 8), ("persistent", 0)], [("kernel", 2), ("rows_per_tile", 8), ("persistent", 1)],
                     [("rows_per_tile", 0), ("kernel", 1)], [("kernel", 0)]):
            for k, v in opts:
                dev.set_option(k, v)
            t = timeit(lambda: dev.spmv_torch(x, out=y))
            d = dev.describe()
            print(f"  {str(dict(opts)):64s} {t*1e3:7.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %  [{d['kernel']} rpt={d['rows_per_tile']} L={d['lanes_per_row']} "
                  f"stream={d['stream_row_fraction']:.2f} skew={d['skew']} overflow={d['overflow_tiles']} win={d['lds_window_bytes']//1024}K pers={d['persistent']}]", flush=True)
        dev.set_option("rows_per_tile", 0)
        dev.set_option("skew", -1)
        dev.set_option("stream_row_max", 128)
        del dev


if __name__ == "__main__":
    main()
