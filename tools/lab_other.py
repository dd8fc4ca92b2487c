#!/usr/bin/env python3
"""Timing of the non-headline configs (development tool): config 2 / 3 variants
of the CSR kernel, config 4 (CSC scatter), config 5 (COO -> CSR assembly)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402


def timeit(fn, iters, warm=3):
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


def csr_case(name, n, window, dtype, iters=30, opts=()):
    esz = np.dtype(dtype).itemsize
    rp, ci, va = synth.banded_csr(n, n, 14, window or n, synth.matrix_seed(3), dtype=dtype)
    dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    for k, v in opts:
        dev.set_option(k, v)
    x = torch.from_numpy(synth.vector(n, dtype=dtype)).cuda()
    y = torch.empty_like(x)
    t = timeit(lambda: dev.spmv_torch(x, out=y), iters)
    B = synth.spmv_bytes(n * 14, n, n, n, esz)
    d = dev.describe()
    print(f"{name:34s} {t*1e3:9.1f} us {B/t/1e6:8.1f} GB/s {100*B/t/1e6/8000:6.2f} %peak  [{d['kernel']} "
          f"stream={d['stream_row_fraction']} lds={d['lds_x']}]", flush=True)
    return rp, ci, va


def main():
    which = sys.argv[1:] or ["csr", "csc", "coo"]
    if "csr" in which:
        csr_case("cfg3 banded f64", 10_000_000, 4096, np.float64)
        csr_case("cfg3 banded f32", 10_000_000, 4096, np.float32)
        csr_case("cfg3 uniform f64 (stress)", 10_000_000, None, np.float64, iters=10)
        csr_case("cfg3 W=65536 f64", 10_000_000, 65536, np.float64, iters=10)
        csr_case("cfg3 W=16384 f64", 10_000_000, 16384, np.float64, iters=10)
        csr_case("cfg3 W=8192 f64", 10_000_000, 8192, np.float64, iters=10)
        csr_case("cfg3 W=8192 f64 persistent", 10_000_000, 8192, np.float64, iters=10, opts=[("persistent", 1)])
        csr_case("cfg3 banded f64 vector kernel", 10_000_000, 4096, np.float64, opts=[("kernel", 1)])
        csr_case("cfg2 banded f64 (fits MALL)", 1_000_000, 4096, np.float64, iters=100)
    if "csc" in which:
        import scipy.sparse as sps
        n = 1_000_000
        rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(2))
        csc = sps.csr_matrix((va, ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n)).tocsc()
        csc.sort_indices()
        cp, ri, cv = csc.indptr.astype(np.uint64), csc.indices.astype(np.uint64), csc.data
        dev = sp.CscMatrix._trusted(n, n, cp, ri, cv).device()
        x = torch.from_numpy(synth.vector(n)).cuda()
        y = torch.empty_like(x)
        B = synth.spmv_bytes(n * 14, n, n, n, 8)
        dev.set_option("kernel", 1)
        for lds, flush in ((1, 1), (1, 0), (0, 0)):
            dev.set_option("lds", lds)
            dev.set_option("flush", flush)
            t = timeit(lambda: dev.spmv_torch(x, out=y), 50)
            print(f"cfg4 CSC scatter f64 lds={lds} flush={flush}     {t*1e3:9.1f} us {B/t/1e6:8.1f} GB/s "
                  f"{100*B/t/1e6/8000:6.2f} %peak {dev.describe()}", flush=True)
    if "coo" in which:
        for length, nr in ((5_000_000, 500_000), (50_000_000, 5_000_000)):
            r, c, v = synth.coo(nr, nr, length, synth.matrix_seed(5), 10, 1)
            coo = sp.CooMatrix.with_triplets(nr, nr, r, c, v)
            t0 = time.time()
            d = coo.upload()
            t_up = time.time() - t0
            torch.cuda.synchronize()
            ts = []
            for _ in range(8):
                t0 = time.perf_counter()
                csr = d.assemble_csr()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
                nnz = csr.shape()[2]
                plan = csr.describe()
                csr.close()
            t = min(ts)
            lb = synth.assembly_bytes(length, nnz, nr)
            print(f"cfg5 COO->CSR len={length:>9d} nnz_out={nnz:>9d}: {t*1e3:8.2f} ms  {length/t/1e6:8.1f} Mentries/s "
                  f"lower-bound bytes {lb/1e9:.2f} GB -> {lb/t/1e9:7.1f} GB/s eff ({100*lb/t/8e12:.2f} % of peak); "
                  f"upload {t_up:.2f}s; plan kernel={plan['kernel']}", flush=True)
            del d


if __name__ == "__main__":
    main()
