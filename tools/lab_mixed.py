#!/usr/bin/env python3
"""Banded 14/row with a heavy row every so many super-tiles: what does the stream kernel's in-kernel
fallback cost? (development tool)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402
from tools.lab_zoo import timeit  # noqa: E402

n = 4_000_000
rp, ci, va = synth.banded_csr(n, n, 14, 4096, 3)
rng = np.random.default_rng(2)
for every, heavy in ((0, 0), (20, 3000), (20, 300), (4, 3000), (4, 300), (1, 1500)):
    if every:
        rows = np.arange(500, n - 5000, every * 1024)
        lens = np.diff(rp.astype(np.int64)).copy()
        cols = np.split(ci, rp[1:-1].astype(np.int64))
        vals = np.split(va, rp[1:-1].astype(np.int64))
        for r in rows:
            lo = max(0, min(r - 2048, n - 4096))
            cols[r] = (lo + np.sort(rng.choice(4096, heavy, replace=False))).astype(np.uint64)
            vals[r] = rng.uniform(-1, 1, heavy)
        lens2 = np.array([c.size for c in cols])
        rp2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.uint64)
        ci2, va2 = np.concatenate(cols), np.concatenate(vals)
    else:
        rp2, ci2, va2 = rp, ci, va
    dev = sp.CsrMatrix._trusted(n, n, rp2, ci2, va2).device()
    x = torch.from_numpy(synth.vector(n)).cuda()
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    out = []
    for pers in (0, 1):
        dev.set_option("persistent", pers)
        t = timeit(lambda: dev.spmv_torch(x, out=y))
        out.append(f"{'persistent' if pers else 'plain'} {t*1e3:7.1f} us")
    d = dev.describe()
    print(f"heavy row of {heavy} every {every} super-tiles: " + " | ".join(out) + f"  [stream={d['stream_row_fraction']:.3f} {os.environ.get('SPAL_HIP_LIB','main').split('/')[-2] if os.environ.get('SPAL_HIP_LIB') else 'main'}]", flush=True)
