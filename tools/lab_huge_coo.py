#!/usr/bin/env python3
"""COO assembly with more than 2^31 triplets (development check): product on the assembled CSR against the
product summed straight from the triplets (torch index_add, another order), CSR invariants on a sample."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402

length = int(sys.argv[1]) if len(sys.argv) > 1 else 2_300_000_000
nr = length // 10
t0 = time.time()
r, c, v = synth.coo(nr, nr, length, 1234, 10, 1)
print(f"generated {length} triplets (> 2^31: {length > 2**31}) into {nr} x {nr} in {time.time() - t0:.0f} s", flush=True)
t0 = time.time()
d = sp.CooMatrix.with_triplets(nr, nr, r, c, v).upload()
print(f"uploaded in {time.time() - t0:.0f} s", flush=True)
ts = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a = d.assemble_csr()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
    nnz = a.shape()[2]
    route = d.describe()
    if _ < 2:
        a.close()
print(f"assembled: {min(ts) * 1e3:.1f} ms = {length / min(ts) / 1e9:.1f} G entries/s, nnz {nnz} (< len: {nnz < length}), {route}", flush=True)
x = torch.from_numpy(synth.vector(nr)).cuda()
y = a.spmv_torch(x)
chunk = 200_000_000
y_direct = torch.zeros(nr, dtype=torch.float64, device="cuda")
bound = torch.zeros(nr, dtype=torch.float64, device="cuda")
for lo in range(0, length, chunk):
    hi = min(length, lo + chunk)
    rt = torch.from_numpy(r[lo:hi].astype(np.int64)).cuda()
    ct = torch.from_numpy(c[lo:hi].astype(np.int64)).cuda()
    p = torch.from_numpy(v[lo:hi]).cuda() * x[ct]
    y_direct.index_add_(0, rt, p)
    bound.index_add_(0, rt, p.abs())
    del rt, ct, p
ok = bool(torch.all((y - y_direct).abs() <= 1e-10 * bound + 1e-300))
print(f"product on the result == product from the triplets (1e-10 of sum |a||x|): {ok}; max |diff| {float((y - y_direct).abs().max()):.3e}", flush=True)
