#!/usr/bin/env python3
"""Small shards (what one GPU of eight gets at config 3: 1.25M rows): tile geometry / form A/B
(development tool, interleaved rounds)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spalinalg_amd as sp  # noqa: E402
import spal_synth as synth  # noqa: E402


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    ncols = 10_000_000
    rp, ci, va = synth.banded_csr(10_000_000, ncols, 14, 4096, synth.matrix_seed(3), rows=(0, n))
    variants = {
        "plain rpt64": [("persistent", 0)],
        "persistent rpt64": [("persistent", 1)],
        "plain rpt32": [("rows_per_tile", 32), ("persistent", 0)],
        "persistent rpt32": [("rows_per_tile", 32), ("persistent", 1)],
        "persistent rpt64 1024 blocks": [("persistent", 1), ("persistent_blocks", 1024)],
        "persistent rpt32 1024 blocks": [("rows_per_tile", 32), ("persistent", 1), ("persistent_blocks", 1024)],
    }
    # several copies so the 235 MB shard is not served from the Infinity Cache
    copies = 3
    devs = {}
    for name, opts in variants.items():
        devs[name] = []
        for _ in range(copies):
            d = sp.CsrMatrix._trusted(n, ncols, rp, ci, va).device()
            for k, v in opts:
                d.set_option(k, v)
            devs[name].append(d)
    xs = [torch.from_numpy(synth.vector(ncols)).cuda() for _ in range(copies)]
    ys = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(copies)]
    res = {k: [] for k in variants}
    for rnd in range(4):
        for name in variants:
            state = {"i": 0}

            def fn():
                i = state["i"] % copies
                devs[name][i].spmv_torch(xs[i], out=ys[i])
                state["i"] += 1
            timeit(fn, 30)
            res[name].append(timeit(fn, 150))
    B = synth.spmv_bytes(n * 14, n, n, 0, 8) + 0
    for name, v in res.items():
        print(f"{name:32s} " + " ".join(f"{t:7.2f}" for t in v) + f"   min {min(v):7.2f} us  {devs[name][0].describe()['blocks']} blocks", flush=True)


if __name__ == "__main__":
    main()
