#!/usr/bin/env python3
"""Long rows (100 / 400 per row) through the vector kernel's knobs (development tool)."""
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
    cases = ((70, 1_400_000, 2048), (100, 1_000_000, 2048), (200, 500_000, 4096), (400, 250_000, 4096), (1500, 64_000, 8192))
    if "rsweep" in sys.argv[1:]:
        cases = ((200, 500_000, 4096),)
    elif "threads" in sys.argv[1:]:
        cases = ((150, 640_000, 4096), (300, 320_000, 4096), (400, 250_000, 4096), (600, 160_000, 4096), (800, 120_000, 4096), (1000, 96_000, 8192))
    for per, n, W in cases:
        rp, ci, va = from_lens(np.full(n, per, np.int64), lambda r, p, g: np.clip(r - W // 2 + g.integers(0, W, r.size), 0, n - 1), rng)
        nnz = int(rp[-1])
        B = synth.spmv_bytes(nnz, n, n, n, 8)
        dev = sp.CsrMatrix(n, n, rp, ci, va).device()
        x = torch.from_numpy(synth.vector(n)).cuda()
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        t = timeit(lambda: dev.spmv_torch(x, out=y))
        d = dev.describe()
        print(f"{per}/row n={n} nnz={nnz}: auto {t*1e3:8.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %  [{d['kernel']} L={d['lanes_per_row']} U={d['unroll']} R={d['rows_per_block']} lds={d['lds_x']} index_bits={d['index_bits']}]", flush=True)
        if d["kernel"] == "vector":      # the same plan with 32-bit columns, and back
            for col16 in (0, 1, 0, 1):
                dev.set_option("col16", col16)
                t = timeit(lambda: dev.spmv_torch(x, out=y))
                print(f"    col16={col16}: {t*1e3:8.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %  [index_bits={dev.describe()['index_bits']}]", flush=True)
        if "rsweep" in sys.argv[1:] and d["kernel"] == "vector":
            for R in (448, 480, 496, 512, 528, 576, 640, 496, 512):
                dev.set_option("rows_per_block", R)
                t = timeit(lambda: dev.spmv_torch(x, out=y), iters=20)
                dd = dev.describe()
                print(f"    R={R}: {t*1e3:8.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %  [blocks={dd['blocks']} win={dd['lds_window_bytes']} bits={dd['index_bits']}]", flush=True)
            continue
        if "threads" in sys.argv[1:] and d["kernel"] == "vector":
            for threads in (512, 1024, 512, 1024):
                dev.set_option("threads", threads)
                t = timeit(lambda: dev.spmv_torch(x, out=y), iters=20)
                dd = dev.describe()
                print(f"    threads={threads}: {t*1e3:8.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %  [R={dd['rows_per_block']} lds={dd['lds_x']} bits={dd['index_bits']}]", flush=True)
            continue
        if "sweep" in sys.argv[1:] and d["kernel"] == "vector":
            for threads in (512, 1024):
                for R in (32, 64, 128, 256, 512, 1024):
                    try:
                        dev.set_option("threads", threads)
                        dev.set_option("rows_per_block", R)
                    except Exception as exc:  # noqa: BLE001
                        print("   ", threads, R, "refused:", exc)
                        continue
                    t = timeit(lambda: dev.spmv_torch(x, out=y), iters=20)
                    dd = dev.describe()
                    print(f"    threads={threads} R={R:4d}: {t*1e3:8.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %  [L={dd['lanes_per_row']} U={dd['unroll']} lds={dd['lds_x']} bits={dd['index_bits']}]", flush=True)
            dev.set_option("rows_per_block", 0)
            dev.set_option("threads", 0)
        for L in ():
            for U in (1, 2, 4):
                for R in (64, 128, 256, 512, 1024):
                    try:
                        dev.set_option("kernel", 1)
                        dev.set_option("lanes_per_row", L)
                        dev.set_option("unroll", U)
                        dev.set_option("rows_per_block", R)
                    except Exception as exc:  # noqa: BLE001
                        print("   ", L, U, R, "refused:", exc)
                        continue
                    t = timeit(lambda: dev.spmv_torch(x, out=y), iters=10)
                    print(f"    L={L:2d} U={U} R={R:4d}: {t*1e3:8.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %", flush=True)


if __name__ == "__main__":
    main()
