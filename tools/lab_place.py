#!/usr/bin/env python3
"""Does the PLACEMENT of a handle's arrays in device memory change the kernel's time?  Several handles of
the same matrix (config 3), same plan, timed interleaved in one process; prints each handle's array
addresses next to its time.  Between handles a dummy block of an odd size is allocated so that the
placements differ.  Development tool."""
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402

n = 10_000_000
rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(3))
opts = [kv.split("=") for kv in sys.argv[1:] if "=" in kv]
nh = 8
devs, dummies = [], []
for i in range(nh):
    d = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    for k, v in opts:
        d.set_option(k, int(v))
    devs.append(d)
    dummies.append(torch.empty((i + 1) * 1_234_567 * 8 + 4096 * i, dtype=torch.uint8, device="cuda"))
x = torch.from_numpy(synth.vector(n)).cuda()
ys = [torch.empty_like(x) for _ in range(nh)]
times = [[] for _ in range(nh)]
if "tune" in sys.argv[1:]:
    # what the setup-time autotune makes of each handle's placement
    for i, d in enumerate(devs):
        pl = d.autotune(x, ys[i], iters=20)
        print(f"handle {i}: autotune_us {pl['autotune_us']} placement_us {pl['placement_us']} tries {pl['placement_tries']} "
              f"slide {pl['slide']} nt {pl['nt_store']}", flush=True)
if "pmc" in sys.argv[1:]:
    # counter passes (rocprofv3 --pmc): 6 launches per handle, handle after handle, so that the stream kernel's
    # dispatches group by handle in the CSV (tools/pmc_by_handle.py); first a timing of each for the record
    for i, d in enumerate(devs):
        for _ in range(2):
            d.spmv_torch(x, out=ys[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            d.spmv_torch(x, out=ys[i])
        e1.record()
        torch.cuda.synchronize()
        a = d.describe()["addr"]
        print(f"handle {i}: values {a[0]} col16 {a[1]} {e0.elapsed_time(e1) / 4 * 1e3:6.1f} us", flush=True)
    sys.exit(0)
for rnd in range(4):
    for i, d in enumerate(devs):
        for _ in range(3):
            d.spmv_torch(x, out=ys[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(25):
            d.spmv_torch(x, out=ys[i])
        e1.record()
        torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / 25 * 1e3)
for i, d in enumerate(devs):
    a = d.describe()["addr"]
    print(f"handle {i}: values {a[0]} col16 {a[1]} rowptr {a[2]} y {ys[i].data_ptr():x}  "
          f"median {statistics.median(times[i]):6.1f} us  rounds {[round(t, 1) for t in times[i]]}", flush=True)
# one handle, its y moved around
d = devs[0]
big = torch.empty(n + 4096, dtype=torch.float64, device="cuda")
for off in (0, 32, 64, 128, 256, 512, 1024, 2048):
    yy = big[off:off + n]
    for _ in range(3):
        d.spmv_torch(x, out=yy)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(25):
        d.spmv_torch(x, out=yy)
    e1.record()
    torch.cuda.synchronize()
    print(f"handle 0, y at +{off * 8:6d} B: {e0.elapsed_time(e1) / 25 * 1e3:6.1f} us", flush=True)
