#!/usr/bin/env python3
"""Power-law row lengths (graph-like), local and uniform columns: the stream plan with different thresholds
for handing a tile to the overflow kernel, and the vector kernel (development tool)."""
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
    n = 2_000_000
    pl = np.minimum((rng.pareto(1.6, n) * 6 + 1).astype(np.int64), 5000)
    cases = [("local +-5000", lambda r, p, g: np.clip(r - 5000 + g.integers(0, 10000, r.size), 0, n - 1)),
             ("uniform", lambda r, p, g: g.integers(0, n, r.size))]
    only = [a for a in sys.argv[1:] if a in ("local", "uniform")]
    quick = "quick" in sys.argv[1:]
    for name, fn in cases:
        if only and name.split()[0] not in only:
            continue
        rp, ci, va = from_lens(pl, fn, rng)
        nnz = int(rp[-1])
        B = synth.spmv_bytes(nnz, n, n, n, 8)
        dev = sp.CsrMatrix(n, n, rp, ci, va).device()
        x = torch.from_numpy(synth.vector(n)).cuda()
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        print(f"power-law rows (mean {pl.mean():.1f}, max {pl.max()}), columns {name}: nnz {nnz}, floor {B/8e12*1e6:.1f} us", flush=True)
        variants = ([("kernel", 0)], [("stream_row_max", 128), ("rows_per_tile", 64)], [("rows_per_tile", 64), ("window_pages", 24)],
                    [("rows_per_tile", 16), ("window_pages", 24)]) if quick else None
        for opts in variants or ([("kernel", 0)], [("stream_row_max", 1024)], [("stream_row_max", 256)], [("stream_row_max", 128)],
                     [("stream_row_max", 64)], [("stream_row_max", 32)],
                     [("stream_row_max", 128), ("rows_per_tile", 64)], [("stream_row_max", 128), ("rows_per_tile", 16)],
                     [("stream_row_max", 128), ("rows_per_tile", 0), ("window_pages", 24)],
                     [("window_pages", 0), ("kernel", 1)]):
            for k, v in opts:
                dev.set_option(k, v)
            for pers in (0, 1):
                d = dev.describe()
                if d["kernel"] != "stream" and pers:
                    continue
                if d["kernel"] == "stream":
                    dev.set_option("persistent", pers)
                t = timeit(lambda: dev.spmv_torch(x, out=y))
                d = dev.describe()
                print(f"  {str(dict(opts)):70s} pers={pers}: {t*1e3:8.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %  "
                      f"[{d['kernel']} rpt={d['rows_per_tile']} stream={d['stream_row_fraction']:.3f} "
                      f"overflow_tiles={d.get('overflow_tiles')} win={d['lds_window_bytes']//1024}K]", flush=True)
        dev.set_option("kernel", 0)
        del dev


if __name__ == "__main__":
    main()
