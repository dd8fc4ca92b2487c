#!/usr/bin/env python3
"""Entry offsets above 2^31 / above 2^32 (development check): 160M x 160M banded, 2.24e9 stored entries (18 GB of
values) by default; `lab_huge.py 330000000` = 4.62e9 entries, more than one set of 32-bit device offsets addresses:
the handle then keeps row blocks (spal_csr_describe: "row_blocks").
Row sums for x = 1 against numpy, random x on sampled rows (incl. the rows around every cut), every kernel form."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 160_000_000
t0 = time.time()
rp, ci, va = synth.banded_csr(n, n, 14, 4096, 99)
print(f"generated {n} rows, {int(rp[-1])} entries (> 2^31: {int(rp[-1]) > 2**31}) in {time.time() - t0:.0f} s", flush=True)
t0 = time.time()
dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
print(f"uploaded + planned in {time.time() - t0:.0f} s: {dev.describe()}", flush=True)
rowsum = va.reshape(n, 14).sum(axis=1)
ones = torch.ones(n, dtype=torch.float64, device="cuda")
xr = torch.from_numpy(synth.vector(n)).cuda()
xh = xr.cpu().numpy()
rows = np.concatenate([np.arange(0, 2000), np.arange(n - 2000, n), np.random.default_rng(1).integers(0, n, 20000),
                       np.arange((2**31) // 14 - 1000, (2**31) // 14 + 1000), np.arange((2**32 - 2**30) // 14, (2**32 - 2**30) // 14 + 1000)])
d = dev.describe()
if d["kernel"] == "row_blocks":
    print(f"row blocks: {d['parts']} parts, cuts at rows {d['part_rows']}", flush=True)
    rows = np.concatenate([rows] + [np.arange(max(c - 1500, 0), min(c + 1500, n)) for c in d["part_rows"][1:-1]])
rows = rows[rows < n]
for opts in ((("persistent", 0),), (("persistent", 1),), (("kernel", 1),)):
    for k, v in opts:
        dev.set_option(k, v)
    y1 = dev.spmv_torch(ones).cpu().numpy()
    ok1 = bool(np.allclose(y1, rowsum, rtol=0, atol=1e-12))
    y2 = dev.spmv_torch(xr).cpu().numpy()
    bad = 0
    for r in rows:
        lo, hi = int(rp[r]), int(rp[r + 1])
        ref = float(np.dot(va[lo:hi], xh[ci[lo:hi].astype(np.int64)]))
        if abs(y2[r] - ref) > 1e-12:
            bad += 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dev.spmv_torch(xr, out=ones)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    B = synth.spmv_bytes(int(rp[-1]), n, n, n, 8)
    ones.fill_(1.0)
    print(f"{opts}: row sums ok={ok1}, sampled rows wrong={bad} of {rows.size}, {t:.2f} ms = {100 * B / (t * 1e-3) / 8e12:.1f} % of 8 TB/s", flush=True)
