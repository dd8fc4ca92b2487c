"""Development check: does the product's time depend on how many launches run back to back (clocks under sustained load)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spalinalg_amd as sp, spal_synth as synth
n = 10_000_000
rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(3))
d = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
x = torch.from_numpy(synth.vector(n)).cuda(); y = torch.empty_like(x)
for _ in range(10): d.spmv_torch(x, out=y)
torch.cuda.synchronize()
import time
for rnd in range(3):
    for k in (10, 30, 100, 300, 1000, 30, 10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k): d.spmv_torch(x, out=y)
        e1.record(); torch.cuda.synchronize()
        print(f"{k:5d} launches: {e0.elapsed_time(e1)/k*1e3:7.1f} us each", flush=True)
        time.sleep(0.05)
