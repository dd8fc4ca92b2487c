#!/usr/bin/env python3
"""Kernel lab: sweeps the CSR SpMV plan knobs on one matrix and prints a table
(time, algorithmic GB/s, fraction of the 8 TB/s HBM peak), next to two
streaming ceilings measured on the same device.  Development tool only."""
import argparse
import itertools
import json
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--per-row", type=int, default=14)
    ap.add_argument("--window", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--sweep", default="default")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    n = args.rows
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    esz = np.dtype(np_dt).itemsize
    t0 = time.time()
    rp, ci, va = synth.banded_csr(n, n, args.per_row, args.window or n, synth.matrix_seed(3), dtype=np_dt)
    print(f"generated in {time.time()-t0:.1f}s", flush=True)
    t0 = time.time()
    dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    print(f"uploaded in {time.time()-t0:.1f}s; default plan {dev.describe()}", flush=True)
    x = torch.from_numpy(synth.vector(n, dtype=np_dt)).cuda()
    y = torch.empty(n, dtype=x.dtype, device="cuda")
    nnz = n * args.per_row
    B = synth.spmv_bytes(nnz, n, n, n, esz)
    results = []

    # ceilings: a pure read (sum) and a copy of as many bytes as the matrix stream
    big = torch.empty(nnz * (esz + 4) // 8, dtype=torch.float64, device="cuda").normal_()
    dst = torch.empty_like(big)
    t = timeit(lambda: big.sum(), args.iters)
    print(f"ceiling  torch.sum  read {big.numel()*8/1e9:.2f} GB: {t*1e3:8.1f} us  {big.numel()*8/t/1e6:8.1f} GB/s")
    t = timeit(lambda: dst.copy_(big), args.iters)
    print(f"ceiling  torch.copy r+w  {2*big.numel()*8/1e9:.2f} GB: {t*1e3:8.1f} us  {2*big.numel()*8/t/1e6:8.1f} GB/s")
    del big, dst

    # reference result with a conservative plan
    dev.set_option("kernel", 1)
    dev.set_option("lds_x", 0)
    dev.set_option("unroll", 1)
    dev.set_option("threads", 512)
    yref = dev.spmv_torch(x).clone()
    dev.set_option("kernel", 0)

    if args.sweep == "default":
        grid = dict(threads=[512, 1024], rows_per_block=[1024, 4096], unroll=[2, 4],
                    lanes_per_row=[16], lds_x=[1])
        extra = [dict(threads=512, rows_per_block=1024, unroll=2, lanes_per_row=16, lds_x=0),
                 dict(threads=1024, rows_per_block=4096, unroll=4, lanes_per_row=16, lds_x=0)]
    else:
        grid = json.loads(args.sweep)
        extra = []
    combos = [dict(zip(grid, v)) for v in itertools.product(*grid.values())] + extra
    # the stream kernel (fixed geometry)
    try:
        dev.set_option("kernel", 2)
        t = timeit(lambda: dev.spmv_torch(x, out=y), args.iters)
        ok = bool(torch.allclose(y, yref, rtol=1e-10 if esz == 8 else 1e-4, atol=1e-11 if esz == 8 else 1e-4))
        gbs = B / t / 1e6
        print(f"stream kernel: {t*1e3:9.1f} us {gbs:8.1f} GB/s {100*gbs/8000:6.2f} %peak ok={ok} {dev.describe()}", flush=True)
        results.append(dict(kernel=2, us=t * 1e3, gbs=gbs, ok=ok, plan=dev.describe()))
    except Exception as e:  # noqa: BLE001
        print("stream kernel FAILED", e, flush=True)
    dev.set_option("kernel", 1)
    print(f"{'threads':>7} {'R':>6} {'U':>2} {'L':>3} {'lds':>3} | {'us':>9} {'GB/s':>8} {'%peak':>6}  ok")
    for c in combos:
        try:
            dev.set_option("kernel", 1)
            for k, v in c.items():
                dev.set_option(k, v)
            t = timeit(lambda: dev.spmv_torch(x, out=y), args.iters)
            ok = bool(torch.allclose(y, yref, rtol=1e-10 if esz == 8 else 1e-4, atol=1e-11 if esz == 8 else 1e-4))
            gbs = B / t / 1e6
            print(f"{c['threads']:>7} {c['rows_per_block']:>6} {c['unroll']:>2} {c['lanes_per_row']:>3} "
                  f"{c['lds_x']:>3} | {t*1e3:9.1f} {gbs:8.1f} {100*gbs/8000:6.2f}  {ok}", flush=True)
            results.append(dict(c, us=t * 1e3, gbs=gbs, ok=ok, plan=dev.describe()))
        except Exception as e:  # noqa: BLE001
            print(c, "FAILED", e, flush=True)
    if args.out:
        json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
