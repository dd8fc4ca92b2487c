#!/usr/bin/env python3
"""Development labs (NOT product, NOT tests): the measurement scripts behind the tables in profiles/ and DESIGN.md,
one sub-command each.  Every lab builds its inputs with spal_synth, runs the product path through the Python mirror of
the C ABI on cuda:0 and prints what it measured; none of them is imported by the package, the tests or bench.py.

    python tools/lab.py <name> [arguments of that lab ...]        python tools/lab.py --list

(Round 3: these were 25 separate tools/lab_<name>.py files; profiles/r01 and profiles/r02 cite them under that name.)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LABS = {}


def lab(fn):
    LABS[fn.__name__] = fn
    return fn


@lab
def ab():
    """Interleaved A/B timing of CSR SpMV plan variants in ONE process on ONE device
    (N variants x M rounds, median and min reported): the only way to see
    differences smaller than the run-to-run / box-to-box spread.  Development tool.

      python tools/lab_ab.py "kernel=2,tiles_per_wave=4" "kernel=2,tiles_per_wave=8"
    (was tools/lab_ab.py)"""
    import os
    import statistics
    import sys

    import numpy as np

    import torch  # noqa: E402
    import spalinalg_amd as sp  # noqa: E402
    import spal_synth as synth  # noqa: E402

    def main():
        variants = [a for a in sys.argv[1:] if "=" in a and not a.startswith("@")]
        flags = [a for a in sys.argv[1:] if "=" not in a]
        shape = dict(kv[1:].split("=") for kv in sys.argv[1:] if kv.startswith("@"))   # @per_row=27 @window=8192 @rows=5000000
        n = int(shape.get("rows", 10_000_000))
        per_row = int(shape.get("per_row", 14))
        window = None if "uniform" in flags else int(shape.get("window", 4096))
        dtype = np.float32 if "f32" in flags else np.float64
        rounds, iters = 7, 25
        rp, ci, va = synth.banded_csr(n, n, per_row, window or n, synth.matrix_seed(3), dtype=dtype)
        devs = []
        for v in variants:  # one handle per variant: no re-planning inside the timed rounds
            d = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
            for kv in v.split(","):
                k, val = kv.split("=")
                d.set_option(k, int(val))
            devs.append(d)
        x = torch.from_numpy(synth.vector(n, dtype=dtype)).cuda()
        y = torch.empty_like(x)
        yref = devs[0].spmv_torch(x).clone()
        B = synth.spmv_bytes(n * per_row, n, n, n, np.dtype(dtype).itemsize)
        times = [[] for _ in variants]
        for r in range(rounds):
            for i, d in enumerate(devs):
                for _ in range(3):
                    d.spmv_torch(x, out=y)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    d.spmv_torch(x, out=y)
                e1.record()
                torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / iters * 1e3)
        for v, t, d in zip(variants, times, devs):
            ok = bool(torch.equal(d.spmv_torch(x), yref))
            pl = d.describe()
            v = f"{v} [{pl['kernel']} rpt={pl.get('rows_per_tile')} stream={pl['stream_row_fraction']}]"
            med, mn = statistics.median(t), min(t)
            print(f"{v:72s} median {med:7.1f} us  min {mn:7.1f} us  {B/med/1e3:7.1f} GB/s ({100*B/med/1e3/8000:5.2f} %)  "
                  f"bit-equal-to-first={ok}  rounds={[round(q) for q in t]}", flush=True)

    if True:  # (was the script's __main__ block)
        main()


@lab
def ab1():
    """Interleaved A/B timing of plan variants on ONE handle (one placement of the arrays in device memory:
    handles of the same matrix differ by up to 8 % with identical code, tools/lab_place.py), the options of
    each variant re-applied before its rounds.  Development tool.

      python tools/lab_ab1.py "prefetch=1" "prefetch=2" [@rows=... @per_row=... @window=...] [f32] [uniform] [ragged]
    (was tools/lab_ab1.py)"""
    import os
    import statistics
    import sys

    import numpy as np

    import torch  # noqa: E402
    import spalinalg_amd as sp  # noqa: E402
    import spal_synth as synth  # noqa: E402

    def main():
        variants = [a for a in sys.argv[1:] if "=" in a and not a.startswith("@")]
        flags = [a for a in sys.argv[1:] if "=" not in a]
        shape = dict(kv[1:].split("=") for kv in sys.argv[1:] if kv.startswith("@"))
        n = int(shape.get("rows", 10_000_000))
        per_row = int(shape.get("per_row", 14))
        window = n if "uniform" in flags else int(shape.get("window", 4096))
        dtype = np.float32 if "f32" in flags else np.float64
        rounds, iters = int(shape.get("rounds", 5)), 25
        if "ragged" in flags:
            rp, ci, va = synth.ragged_csr(n, n, window, synth.matrix_seed(3), dtype=dtype)
        else:
            rp, ci, va = synth.banded_csr(n, n, per_row, window, synth.matrix_seed(3), dtype=dtype)
        nnz = int(rp[-1])
        d = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
        x = torch.from_numpy(synth.vector(n, dtype=dtype)).cuda()
        y = torch.empty_like(x)
        yref = d.spmv_torch(x).clone()
        B = synth.spmv_bytes(nnz, n, n, n, np.dtype(dtype).itemsize)
        times = [[] for _ in variants]
        equal, plans = [None] * len(variants), [None] * len(variants)
        defaults = {"slide": -1, "slide_on": 1, "uniform_rows": 1, "prefetch": 1, "persistent": 0, "nt_store": 0, "diag": 0,
                    "tiles_per_wave": 4, "rows_per_tile": 0, "persistent_blocks": 0, "slide_run": 0, "panel_on": 1, "panel_pages": 192, "panel_window": 0, "kernel": 0, "skew": -1, "window_pages": 0, "stream_global": 1, "cblock": -1}
        named = {kv.split("=")[0] for v in variants for kv in v.split(",")}
        for r in range(rounds):
            for i, v in enumerate(variants):
                opts = {k: defaults[k] for k in named if k in defaults}   # every option any variant names, back to its default
                opts.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in v.split(",")})
                for k, val in opts.items():
                    d.set_option(k, val)
                for _ in range(3):
                    d.spmv_torch(x, out=y)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    d.spmv_torch(x, out=y)
                e1.record()
                torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / iters * 1e3)
                equal[i] = bool(torch.equal(y, yref))
                plans[i] = d.describe()
        for v, t, ok, pl in zip(variants, times, equal, plans):
            tag = f"{v} [{pl['kernel']} rpt={pl.get('rows_per_tile')} uni={pl.get('uniform_row_fraction')} pf={pl.get('prefetch')} pers={pl.get('persistent')}]"
            med, mn = statistics.median(t), min(t)
            print(f"{tag:84s} median {med:7.1f} us  min {mn:7.1f} us  {B/med/1e3:7.1f} GB/s alg ({100*B/med/1e3/8000:5.2f} %)  "
                  f"bit-equal-to-first={ok}  rounds={[round(q) for q in t]}", flush=True)

    if True:  # (was the script's __main__ block)
        main()


@lab
def align():
    """Does the placement of x / y relative to the matrix arrays change the kernel time?
    (development tool; rocprof showed 266 us into one y buffer and 293 us into another in the same process)
    (was tools/lab_align.py)"""
    import os
    import sys

    import numpy as np

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
        n = 10_000_000
        rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(3))
        dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
        for kv in sys.argv[1:]:
            k, v = kv.split("=")
            dev.set_option(k, int(v))
        xh = torch.from_numpy(synth.vector(n))
        slack = 1 << 21
        X = torch.zeros(n + slack, dtype=torch.float64, device="cuda")
        Y = torch.zeros(n + slack, dtype=torch.float64, device="cuda")
        Y2 = torch.zeros(n + slack, dtype=torch.float64, device="cuda")
        print("ptrs", hex(X.data_ptr()), hex(Y.data_ptr()), hex(Y2.data_ptr()), dev.describe(), flush=True)
        offs = [0, 16, 32, 64, 256, 512, 1024, 4096, 16384, 65536, 262144, 1 << 20]
        X[:n].copy_(xh)
        for _ in range(60):
            dev.spmv_torch(X[:n], out=Y[:n])
        torch.cuda.synchronize()
        res = {("y", o): [] for o in offs}
        res.update({("y2", o): [] for o in offs})
        res.update({("x", o): [] for o in offs})
        for rnd in range(3):
            for o in offs:
                res[("y", o)].append(timeit(lambda: dev.spmv_torch(X[:n], out=Y[o:o + n]), 20))
                res[("y2", o)].append(timeit(lambda: dev.spmv_torch(X[:n], out=Y2[o:o + n]), 20))
            for o in offs:
                X[o:o + n].copy_(xh)
                res[("x", o)].append(timeit(lambda: dev.spmv_torch(X[o:o + n], out=Y[:n]), 20))
            X[:n].copy_(xh)
        for key, v in res.items():
            print(f"{key[0]:3s} offset {key[1]:8d} elements: " + "  ".join(f"{t:7.1f}" for t in v) + f"   min {min(v):7.1f} us", flush=True)

    if True:  # (was the script's __main__ block)
        main()


@lab
def burst():
    """Development check: does the product's time depend on how many launches run back to back (clocks under sustained load)?
    (was tools/lab_burst.py)"""
    import os, sys
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


@lab
def coo_once():
    """Development tool: config-5 triplets, a few assemblies (no oracle check) -- for profiling variants of the COO kernels.
    (was tools/lab_coo_once.py)"""
    import os
    import sys
    import spalinalg_amd as sp
    import spal_synth as synth

    cfg = synth.CONFIGS[5]
    n = cfg["nrows"]
    r, c, v = synth.coo(n, n, cfg["length"], synth.matrix_seed(5), cfg["dup_permille"], cfg["cancel_permille"])
    d = sp.CooMatrix.with_triplets(n, n, r, c, v).upload()
    for _ in range(8):
        d.assemble_csr().close()
    print("ok")


@lab
def coo_stress():
    """Development check: repeated COO -> CSR assemblies of one handle (the look-back placement under repetition): every
    result's arrays are compared with the first assembly's, bit for bit.
    (was tools/lab_coo_stress.py)"""
    import os, sys
    import numpy as np
    import spalinalg_amd as sp, spal_synth as synth

    n, length = 2_000_000, 20_000_000
    r, c, v = synth.coo(n, n, length, 11, 10, 1)
    d = sp.CooMatrix.with_triplets(n, n, r, c, v).upload()
    first = None
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    bad = 0
    for i in range(N):
        a = d.assemble_csr()
        rp, ci, va = a.download()
        a.close()
        if first is None:
            first = (rp, ci, va.view(np.uint64))
            print("nnz", ci.size, d.describe(), flush=True)
        elif not (np.array_equal(rp, first[0]) and np.array_equal(ci, first[1]) and np.array_equal(va.view(np.uint64), first[2])):
            bad += 1
    print(f"{N} assemblies, {bad} differ from the first", flush=True)


@lab
def csc_capture():
    """Development check: the CSC scatter entry point under torch's graph capture, per flush form.
    (was tools/lab_csc_capture.py)"""
    import os, sys
    import numpy as np, torch
    import spalinalg_amd as sp, spal_synth as synth
    import scipy.sparse as sps
    n = 100_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 33)
    m = sps.csr_matrix((va, ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n)).tocsc()
    m.sort_indices()
    cp, ri, cv = m.indptr.astype(np.uint64), m.indices.astype(np.uint64), m.data
    x = synth.vector(n)
    y_ref = m @ x
    xt = torch.from_numpy(x).cuda()
    for flush in (2, 1, 0):
        dev = sp.CscMatrix(n, n, cp, ri, cv).device()
        dev.set_option("kernel", 1)
        dev.set_option("flush", flush)
        print("flush", flush, dev.describe()["flush"], flush=True)
        g = torch.cuda.CUDAGraph()
        yg = torch.zeros(n, dtype=torch.float64, device="cuda")
        cap = torch.cuda.Stream()
        with torch.cuda.stream(cap):
            dev.spmv_torch(xt, yg)
            torch.cuda.synchronize()
            print("  eager max err", float(np.abs(yg.cpu().numpy() - y_ref).max()), flush=True)
            with torch.cuda.graph(g, stream=cap):
                dev.spmv_torch(xt, yg)
        for rep in range(2):
            yg.fill_(float("nan"))
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            yh = yg.cpu().numpy()
            print("  replay", rep, "nan count", int(np.isnan(yh).sum()), "max err", float(np.nanmax(np.abs(yh - y_ref))), yh[:4], flush=True)


@lab
def csc_stress():
    """Development check: the CSC scatter kernel's neighbour hand-off under repetition -- thousands of launches on several
    streams, every result compared with the first (a missed hand-off would leave a row without one super-tile's share).
    (was tools/lab_csc_stress.py)"""
    import os, sys
    import numpy as np, torch
    import scipy.sparse as sps
    import spalinalg_amd as sp, spal_synth as synth

    n = 1_000_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(2))
    m = sps.csr_matrix((va, ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n)).tocsc()
    m.sort_indices()
    x = synth.vector(n)
    y_ref = torch.from_numpy(m @ x).cuda()
    dev = sp.CscMatrix(n, n, m.indptr.astype(np.uint64), m.indices.astype(np.uint64), m.data).device()
    dev.set_option("kernel", 1)
    print(dev.describe()["flush"], flush=True)
    xt = torch.from_numpy(x).cuda()
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in streams]
    scale = float(y_ref.abs().max())
    bad = 0
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    for i in range(N):
        k = i % 3
        with torch.cuda.stream(streams[k]):
            outs[k].fill_(float("nan"))
            dev.spmv_torch(xt, outs[k])
            err = float((outs[k] - y_ref).abs().max())
        if not (err <= 1e-11 * scale):
            bad += 1
            if bad < 5:
                print("launch", i, "max abs err", err, flush=True)
    torch.cuda.synchronize()
    print(f"{N} launches, {bad} with a wrong result", flush=True)


@lab
def csr():
    """Kernel lab: sweeps the CSR SpMV plan knobs on one matrix and prints a table
    (time, algorithmic GB/s, fraction of the 8 TB/s HBM peak), next to two
    streaming ceilings measured on the same device.  Development tool only.
    (was tools/lab_csr.py)"""
    import argparse
    import itertools
    import json
    import os
    import sys
    import time

    import numpy as np

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

    if True:  # (was the script's __main__ block)
        main()


@lab
def fem():
    """Stencil matrices (development tool): the columns of a row cluster in a few narrow bands far
    apart -- the x 'window' of a super-tile spans far more than LDS holds although few distinct
    columns are touched.
    (was tools/lab_fem.py)"""
    import os
    import sys

    import numpy as np

    import torch  # noqa: E402
    import spalinalg_amd as sp  # noqa: E402
    import spal_synth as synth  # noqa: E402

    def stencil_csr(m, points):
        """m^3 grid, 7-point (faces) or 27-point (faces, edges, corners) stencil, row-major numbering."""
        n = m ** 3
        idx = np.arange(n, dtype=np.int64)
        i, j, k = idx // (m * m), (idx // m) % m, idx % m
        offs = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)
                if points == 27 or abs(a) + abs(b) + abs(c) <= 1]
        cols, valid = [], []
        for a, b, c in offs:                       # ascending column order by construction
            ok = (i + a >= 0) & (i + a < m) & (j + b >= 0) & (j + b < m) & (k + c >= 0) & (k + c < m)
            cols.append(idx + a * m * m + b * m + c)
            valid.append(ok)
        cols, valid = np.stack(cols, 1), np.stack(valid, 1)
        lens = valid.sum(1)
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        ci = cols[valid].astype(np.uint64)
        rng = np.random.default_rng(3)
        va = rng.uniform(-1, 1, ci.size)
        return n, rp, ci, va

    def timeit(fn, iters, warm=5):
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
        for m, points in ((216, 7), (150, 27)):
            n, rp, ci, va = stencil_csr(m, points)
            dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
            x = torch.from_numpy(synth.vector(n)).cuda()
            y = torch.empty_like(x)
            nnz = int(rp[-1])
            B = synth.spmv_bytes(nnz, n, n, n, 8)
            for opts in ((), (("persistent", 1),), (("kernel", 1),)):
                for k, v in opts:
                    dev.set_option(k, v)
                t = timeit(lambda: dev.spmv_torch(x, out=y), 30)
                d = dev.describe()
                print(f"{m}^3 {points}-point  n={n} nnz={nnz} opts={opts}: {t*1e3:8.1f} us  {B/t/1e6:8.1f} GB/s = {100*B/t/1e6/8000:5.1f} % "
                      f"[{d['kernel']} stream={d['stream_row_fraction']} lds_rows={d['lds_row_fraction']} rpt={d['rows_per_tile']}]", flush=True)

    if True:  # (was the script's __main__ block)
        main()


@lab
def headline():
    """Config 3 f64, plain and persistent form, for A/B runs of differently built libraries
    (SPAL_HIP_LIB); development tool.
    (was tools/lab_headline.py)"""
    import os
    import sys

    import numpy as np

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

    n = 10_000_000
    dt = np.float32 if "f32" in sys.argv[1:] else np.float64
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(3), dtype=dt)
    dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    x = torch.from_numpy(synth.vector(n, dtype=dt)).cuda()
    y = torch.empty_like(x)
    for _ in range(150):
        dev.spmv_torch(x, out=y)
    out = []
    forms = [("plain", [("tiles_per_wave", 4), ("persistent", 0)]), ("pers", [("tiles_per_wave", 4), ("persistent", 1)])]
    if "tpw8" in sys.argv[1:]:
        forms += [("plain8", [("tiles_per_wave", 8), ("persistent", 0)]), ("pers8", [("tiles_per_wave", 8), ("persistent", 1)])]
    for rnd in range(3):
        for name, opts in forms:
            for k, v in opts:
                dev.set_option(k, v)
            timeit(lambda: dev.spmv_torch(x, out=y), 20)
            out.append(f"{name} {timeit(lambda: dev.spmv_torch(x, out=y), 150):6.1f}")
    print(os.environ.get("SPAL_HIP_LIB", "main").split("/")[-2] if os.environ.get("SPAL_HIP_LIB") else "main", " | ".join(out), flush=True)


@lab
def host_overhead():
    """Host time per launch through the Python mirror (development tool): a tiny matrix, so the
    loop is bound by the host, not by the kernel.
    (was tools/lab_host_overhead.py)"""
    import os
    import sys
    import time

    import numpy as np

    import torch  # noqa: E402
    import spalinalg_amd as sp  # noqa: E402
    import spal_synth as synth  # noqa: E402

    n = 2048
    rp, ci, va = synth.banded_csr(n, n, 14, 512, 3)
    dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    x = torch.from_numpy(synth.vector(n)).cuda()
    y = torch.empty_like(x)
    for name, fn in [("dev.spmv_torch", lambda: dev.spmv_torch(x, out=y)),
                     ("torch.add (reference point)", lambda: torch.add(x, x, out=y))]:
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5000):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name:32s} {1e6 * (t1 - t0) / 5000:7.2f} us/call issued, {1e6 * (t2 - t0) / 5000:7.2f} us/call completed", flush=True)


@lab
def huge():
    """Entry offsets above 2^31 / above 2^32 (development check): 160M x 160M banded, 2.24e9 stored entries (18 GB of
    values) by default; `lab_huge.py 330000000` = 4.62e9 entries, more than one set of 32-bit device offsets addresses:
    the handle then keeps row blocks (spal_csr_describe: "row_blocks").
    Row sums for x = 1 against numpy, random x on sampled rows (incl. the rows around every cut), every kernel form.
    (was tools/lab_huge.py)"""
    import os
    import sys
    import time

    import numpy as np

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


@lab
def huge_coo():
    """COO assembly with more than 2^31 triplets (development check): product on the assembled CSR against the
    product summed straight from the triplets (torch index_add, another order), CSR invariants on a sample.
    (was tools/lab_huge_coo.py)"""
    import os
    import sys
    import time

    import numpy as np

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


@lab
def huge_csc():
    """CSR -> CSC on the device and both CSC routes at 1.4e9 entries (development check).
    (was tools/lab_huge_csc.py)"""
    import os
    import sys
    import time

    import numpy as np

    import torch  # noqa: E402
    import spalinalg_amd as sp  # noqa: E402
    import spal_synth as synth  # noqa: E402

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 7)
    csr = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    x = torch.from_numpy(synth.vector(n)).cuda()
    y_csr = csr.spmv_torch(x)
    t0 = time.time()
    csc = csr.to_csc()
    torch.cuda.synchronize()
    print(f"{n} x {n}, {int(rp[-1])} entries: CSR -> CSC on the device in {time.time() - t0:.2f} s", flush=True)
    bound = None
    for kernel in (2, 1):
        t0 = time.time()
        csc.set_option("kernel", kernel)
        y = csc.spmv_torch(x)
        torch.cuda.synchronize()
        first = time.time() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            csc.spmv_torch(x, out=y)
        e1.record()
        torch.cuda.synchronize()
        diff = float((y - y_csr).abs().max())
        print(f"kernel {kernel}: first call {first:.2f} s, then {e0.elapsed_time(e1) / 5:.2f} ms per product; max |y - y_csr| = {diff:.3e} "
              f"({'bit-identical' if bool(torch.equal(y, y_csr)) else 'within rounding' if diff < 1e-12 else 'WRONG'}); {csc.describe()}", flush=True)
    back = csc.to_csr()
    rp2, ci2, va2 = back.download()
    print("CSC -> CSR returns the original arrays:", bool(np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(va2, va)), flush=True)


@lab
def longrows():
    """Long rows (100 / 400 per row) through the vector kernel's knobs (development tool).
    (was tools/lab_longrows.py)"""
    import os
    import sys

    import numpy as np

    import torch  # noqa: E402
    import spalinalg_amd as sp  # noqa: E402
    import spal_synth as synth  # noqa: E402

    def timeit(fn, iters=20, warm=3):
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

    def from_lens(lens, col_fn, rng):
        n = lens.size
        rows = np.repeat(np.arange(n, dtype=np.int64), lens)
        pos = np.arange(rows.size, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
        cols = col_fn(rows, pos, rng)
        key = np.unique(rows * (int(cols.max()) + 1) + cols)
        rows2, cols2 = key // (int(cols.max()) + 1), key % (int(cols.max()) + 1)
        rp = np.concatenate([[0], np.cumsum(np.bincount(rows2, minlength=n))]).astype(np.uint64)
        return rp, cols2.astype(np.uint64), rng.uniform(-1, 1, cols2.size)

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

    if True:  # (was the script's __main__ block)
        main()


@lab
def mixed():
    """Banded 14/row with a heavy row every so many super-tiles: what does the stream kernel's in-kernel
    fallback cost? (development tool)
    (was tools/lab_mixed.py)"""
    import os
    import sys

    import numpy as np

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


@lab
def other():
    """Timing of the non-headline configs (development tool): config 2 / 3 variants
    of the CSR kernel, config 4 (CSC scatter), config 5 (COO -> CSR assembly).
    (was tools/lab_other.py)"""
    import os
    import sys
    import time

    import numpy as np

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

    if True:  # (was the script's __main__ block)
        main()


@lab
def place():
    """Does the PLACEMENT of a handle's arrays in device memory change the kernel's time?  Several handles of
    the same matrix (config 3), same plan, timed interleaved in one process; prints each handle's array
    addresses next to its time.  Between handles a dummy block of an odd size is allocated so that the
    placements differ.  Development tool.
    (was tools/lab_place.py)"""
    import os
    import statistics
    import sys

    import numpy as np

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


@lab
def place2():
    """Placement classes against the walking order (development tool): several handles of config 3 in one process; for each,
    the sliding kernel fully persistent (512 fronts), with runs dealt round-robin (compact fronts per XCD), with half the
    grid, and the one-super-tile-per-workgroup kernel.  Does the slower placement class hurt the scattered fronts more?
    (was tools/lab_place2.py)"""
    import os, sys, statistics
    import numpy as np
    import torch
    import spalinalg_amd as sp, spal_synth as synth

    n = 10_000_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(3))
    x = torch.from_numpy(synth.vector(n)).cuda()
    y = torch.empty_like(x)
    variants = [("slide", {"slide_on": 1, "slide_run": 0, "persistent_blocks": 0}),
                ("slide run=8", {"slide_on": 1, "slide_run": 8, "persistent_blocks": 0}),
                ("slide run=24", {"slide_on": 1, "slide_run": 24, "persistent_blocks": 0}),
                ("slide 256 wgs", {"slide_on": 1, "slide_run": 0, "persistent_blocks": 256}),
                ("plain", {"slide_on": 0, "slide_run": 0, "persistent_blocks": 0})]
    keep = []
    for h in range(6):
        d = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
        keep.append(torch.empty((h + 1) * 3_456_789 * 8, dtype=torch.uint8, device="cuda"))
        row = []
        for name, opts in variants:
            for k, v in opts.items():
                d.set_option(k, v)
            ts = []
            for rnd in range(3):
                for _ in range(3):
                    d.spmv_torch(x, out=y)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    d.spmv_torch(x, out=y)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
            row.append(min(ts))
        print(f"handle {h} values@{d.describe()['addr'][0]}: " + "  ".join(f"{nm} {t:6.1f}" for (nm, _), t in zip(variants, row)), flush=True)
        keep.append(d)


@lab
def powerlaw():
    """Power-law row lengths (graph-like), local and uniform columns: the stream plan with different thresholds
    for handing a tile to the overflow kernel, and the vector kernel (development tool).
    (was tools/lab_powerlaw.py)"""
    import os
    import sys

    import numpy as np

    import torch  # noqa: E402
    import spalinalg_amd as sp  # noqa: E402
    import spal_synth as synth  # noqa: E402

    def timeit(fn, iters=20, warm=3):
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

    def from_lens(lens, col_fn, rng):
        n = lens.size
        rows = np.repeat(np.arange(n, dtype=np.int64), lens)
        pos = np.arange(rows.size, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
        cols = col_fn(rows, pos, rng)
        key = np.unique(rows * (int(cols.max()) + 1) + cols)
        rows2, cols2 = key // (int(cols.max()) + 1), key % (int(cols.max()) + 1)
        rp = np.concatenate([[0], np.cumsum(np.bincount(rows2, minlength=n))]).astype(np.uint64)
        return rp, cols2.astype(np.uint64), rng.uniform(-1, 1, cols2.size)

    def main():
        rng = np.random.default_rng(5)
        n = 2_000_000
        pl = np.minimum((rng.pareto(1.6, n) * 6 + 1).astype(np.int64), 5000)
        if "sorted" in sys.argv[1:]:   # the same rows, ordered by length inside every block of 1024: what a permuted copy would see
            nb = n // 1024 * 1024
            pl[:nb] = -np.sort(-pl[:nb].reshape(-1, 1024), axis=1).reshape(-1)
            pl[nb:] = -np.sort(-pl[nb:])
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
            variants = ([("blockwin", 1)], [("blockwin", -1)], [("blockwin", 0), ("row_split", -1)], [("row_split", 0)],
                        [("row_split", -1), ("row_split_threshold", 64)],
                        [("row_split", -1), ("row_split_threshold", 256)]) if quick else None
            if "sorted" in sys.argv[1:]:
                variants = ([("row_split", 1)], [("row_split", 1), ("row_split_threshold", 64)], [("row_split", 1), ("row_split_threshold", 32)])
            for opts in variants or ([("kernel", 0)], [("stream_row_max", 1024)], [("stream_row_max", 256)], [("stream_row_max", 128)],
                         [("stream_row_max", 64)], [("stream_row_max", 32)],
                         [("stream_row_max", 128), ("rows_per_tile", 64)], [("stream_row_max", 128), ("rows_per_tile", 16)],
                         [("stream_row_max", 128), ("rows_per_tile", 0), ("window_pages", 24)],
                         [("window_pages", 0), ("kernel", 1)]):
                for k, v in opts:
                    dev.set_option(k, v)
                for pers in (0, 1):
                    d = dev.describe()
                    if d["kernel"] == "blockwin":
                        if not pers:
                            t = timeit(lambda: dev.spmv_torch(x, out=y))
                            print(f"  {str(dict(opts)):70s}        : {t*1e3:8.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %  [blockwin rows={d['block_rows']} "
                                  f"window={d['window_columns']} setup_us={d['setup_us']}]", flush=True)
                        continue
                    split = d["kernel"] == "split"
                    if split:
                        d = d["short_part"]
                    if d["kernel"] != "stream" and pers:
                        continue
                    if d["kernel"] == "stream":
                        dev.set_option("persistent", pers)
                    t = timeit(lambda: dev.spmv_torch(x, out=y))
                    d = dev.describe()
                    if split:
                        print(f"  (split: {d['split_long_rows']} long rows, {d['split_long_entries']} entries)", end="")
                        d = d["short_part"]
                    print(f"  {str(dict(opts)):70s} pers={pers}: {t*1e3:8.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %  "
                          f"[{d['kernel']} rpt={d['rows_per_tile']} stream={d['stream_row_fraction']:.3f} "
                          f"overflow_tiles={d.get('overflow_tiles')} win={d['lds_window_bytes']//1024}K]", flush=True)
            dev.set_option("kernel", 0)
            del dev

    if True:  # (was the script's __main__ block)
        main()


@lab
def rpt8():
    """Rows of 56 ... 128 entries: the stream kernel with 8-row tiles against the vector kernel (development tool).
    (was tools/lab_rpt8.py)"""
    import os
    import sys

    import numpy as np

    import torch  # noqa: E402
    import spalinalg_amd as sp  # noqa: E402
    import spal_synth as synth  # noqa: E402
    from tools.lab_zoo import timeit  # noqa: E402

    def main():
        sizes = ((54, 2_000_000), (64, 1_500_000), (81, 1_200_000), (100, 1_000_000), (120, 800_000))
        if "long" in sys.argv[1:]:
            sizes = ((150, 640_000), (200, 480_000), (250, 400_000))
        if "pow2" in sys.argv[1:]:
            sizes = tuple((k, 3_000_000 if k <= 17 else 1_500_000 if k < 70 else 1_000_000) for k in (8, 15, 16, 24, 32, 48, 63, 64, 65, 81, 96, 100, 120))
        if "mid" in sys.argv[1:]:
            sizes = tuple((k, 3_000_000 if k <= 20 else 1_500_000 if k < 70 else 1_000_000) for k in (17, 20, 24, 33, 40, 48, 70, 81, 100))
        for per_row, n in sizes:
            rp, ci, va = synth.banded_csr(n, n, per_row, 2048, 7)
            nnz = int(rp[-1])
            B = synth.spmv_bytes(nnz, n, n, n, 8)
            dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
            x = torch.from_numpy(synth.vector(n)).cuda()
            y = torch.empty(n, dtype=torch.float64, device="cuda")
            print(f"band {per_row}/row, W=2048, {n} rows: nnz {nnz}, floor {B/8e12*1e6:.1f} us", flush=True)
            long_opts = ([("kernel", 0)], [("kernel", 2), ("stream_row_max", 256), ("rows_per_tile", 4), ("persistent", 0)],
                         [("kernel", 2), ("stream_row_max", 256), ("rows_per_tile", 8), ("persistent", 0)])
            mid_opts = ([("kernel", 0)], [("kernel", 0), ("persistent", 0)]) + tuple([("kernel", 2), ("rows_per_tile", r), ("persistent", 0)] for r in (64, 48, 32, 24, 16, 12, 8) if 128 <= r * per_row <= 1024)
            pow2_opts = ([("kernel", 0)], [("kernel", 2), ("persistent", 0), ("skew", 0)], [("skew", 1)], [("skew", -1), ("kernel", 1)])
            for opts in mid_opts if "mid" in sys.argv[1:] else pow2_opts if "pow2" in sys.argv[1:] else long_opts if "long" in sys.argv[1:] else ([("kernel", 0)], [("kernel", 2), ("rows_per_tile", 16), ("persistent", 0)],
                         [("kernel", 2), ("rows_per_tile", 8), ("persistent", 0)], [("kernel", 2), ("rows_per_tile", 8), ("persistent", 1)],
                         [("rows_per_tile", 0), ("kernel", 1)], [("kernel", 0)]):
                for k, v in opts:
                    dev.set_option(k, v)
                t = timeit(lambda: dev.spmv_torch(x, out=y))
                d = dev.describe()
                print(f"  {str(dict(opts)):64s} {t*1e3:7.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %  [{d['kernel']} rpt={d['rows_per_tile']} L={d['lanes_per_row']} "
                      f"stream={d['stream_row_fraction']:.2f} skew={d['skew']} overflow={d['overflow_tiles']} win={d['lds_window_bytes']//1024}K pers={d['persistent']}]", flush=True)
            dev.set_option("rows_per_tile", 0)
            dev.set_option("skew", -1)
            dev.set_option("stream_row_max", 128)
            del dev

    if True:  # (was the script's __main__ block)
        main()


@lab
def shortrows():
    """Rows of 1 ... 5 entries (diagonal, tridiagonal, 5-point): which kernel / geometry (development tool).
    (was tools/lab_shortrows.py)"""
    import os
    import sys

    import numpy as np

    import torch  # noqa: E402
    import spalinalg_amd as sp  # noqa: E402
    import spal_synth as synth  # noqa: E402

    def timeit(fn, iters=20, warm=3):
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

    def from_lens(lens, col_fn, rng):
        n = lens.size
        rows = np.repeat(np.arange(n, dtype=np.int64), lens)
        pos = np.arange(rows.size, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
        cols = col_fn(rows, pos, rng)
        key = np.unique(rows * (int(cols.max()) + 1) + cols)
        rows2, cols2 = key // (int(cols.max()) + 1), key % (int(cols.max()) + 1)
        rp = np.concatenate([[0], np.cumsum(np.bincount(rows2, minlength=n))]).astype(np.uint64)
        return rp, cols2.astype(np.uint64), rng.uniform(-1, 1, cols2.size)

    def main():
        rng = np.random.default_rng(5)
        n = 4_000_000
        cases = [("diagonal (1/row)", np.ones(n, np.int64), lambda r, p, g: r),
                 ("tridiagonal (3/row)", np.full(n, 3, np.int64), lambda r, p, g: np.clip(r + p - 1, 0, n - 1)),
                 ("5-point 2000x2000", np.full(n, 5, np.int64), lambda r, p, g: np.clip(r + np.array([-2000, -1, 0, 1, 2000])[p], 0, n - 1))]
        for name, lens, fn in cases:
            rp, ci, va = from_lens(lens, fn, rng)
            nnz = int(rp[-1])
            B = synth.spmv_bytes(nnz, n, n, n, 8)
            dev = sp.CsrMatrix(n, n, rp, ci, va).device()
            x = torch.from_numpy(synth.vector(n)).cuda()
            y = torch.empty(n, dtype=torch.float64, device="cuda")
            print(f"{name}: nnz {nnz}, floor {B/8e12*1e6:.1f} us", flush=True)
            variants = [[("kernel", 0)], [("kernel", 2), ("rows_per_tile", 64), ("persistent", 0)], [("kernel", 2), ("rows_per_tile", 128), ("persistent", 0)], [("kernel", 2), ("rows_per_tile", 256), ("persistent", 0)],
                        [("kernel", 2), ("rows_per_tile", 128), ("persistent", 1)], [("rows_per_tile", 0), ("kernel", 2), ("persistent", 1)], [("kernel", 2), ("persistent", 0), ("tiles_per_wave", 8)],
                        [("tiles_per_wave", 4), ("kernel", 1), ("lanes_per_row", 2), ("unroll", 4)],
                        [("kernel", 1), ("lanes_per_row", 4), ("unroll", 4)],
                        [("kernel", 1), ("lanes_per_row", 2), ("unroll", 4), ("rows_per_block", 4096)],
                        [("kernel", 1), ("lanes_per_row", 2), ("unroll", 4), ("rows_per_block", 2048), ("lds_x", 0)],
                        [("kernel", 1), ("lanes_per_row", 4), ("unroll", 2), ("rows_per_block", 2048)]]
            for opts in variants:
                try:
                    for k, v in opts:
                        dev.set_option(k, v)
                except Exception as e:  # noqa: BLE001
                    print(f"  {str(dict(opts)):90s} rejected: {e}")
                    continue
                t = timeit(lambda: dev.spmv_torch(x, out=y))
                d = dev.describe()
                print(f"  {str(dict(opts)):90s} {t*1e3:7.1f} us = {100*B/(t*1e-3)/8e12:5.1f} %  [{d['kernel']} rpt={d['rows_per_tile']} L={d['lanes_per_row']} U={d['unroll']} "
                      f"R={d['rows_per_block']} lds={d['lds_x']} pers={d['persistent']}]", flush=True)
            del dev

    if True:  # (was the script's __main__ block)
        main()


@lab
def small():
    """Small shards (what one GPU of eight gets at config 3: 1.25M rows): tile geometry / form A/B
    (development tool, interleaved rounds).
    (was tools/lab_small.py)"""
    import os
    import sys

    import numpy as np

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

    if True:  # (was the script's __main__ block)
        main()


@lab
def shard():
    """Shard-sized launches: BASELINE config 2 (1M x 1M, 14 per row inside 4096 columns; what one GPU
    of eight holds of config 3), launches rotating over THREE handles that own their arrays (the Infinity Cache), every
    variant's options re-applied before its rounds, rounds interleaved.
        python tools/lab.py shard ["opt=val,opt=val" ...] [@rows=N] [f32] [csc]
    Without variants: the forms the autotune chooses between."""
    import statistics
    import sys

    import numpy as np
    import torch
    import spalinalg_amd as sp
    import spal_synth as synth

    variants = [a for a in sys.argv[1:] if "=" in a and not a.startswith("@")]
    shape = dict(kv[1:].split("=") for kv in sys.argv[1:] if kv.startswith("@"))
    flags = [a for a in sys.argv[1:] if "=" not in a]
    n = int(shape.get("rows", 1_000_000))
    dtype = np.float32 if "f32" in flags else np.float64
    if not variants:
        variants = ["slide_on=1,nt_store=0", "slide_on=1,nt_store=1", "slide_on=0,nt_store=0", "slide_on=0,nt_store=1",
                    "slide_on=1,nt_store=0,arith_bounds=0"]
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(2), dtype=dtype)
    copies = 3
    m = sp.CsrMatrix._trusted(n, n, rp, ci, va)
    devs = [m.device_copy() for _ in range(copies)]
    xs = [torch.from_numpy(synth.vector(n, dtype=dtype)).cuda() for _ in range(copies)]
    ys = [torch.empty_like(xs[0]) for _ in range(copies)]
    yref = devs[0].spmv_torch(xs[0]).clone()
    B = synth.spmv_bytes(int(rp[-1]), n, n, n, np.dtype(dtype).itemsize)
    defaults = {"slide_on": 1, "nt_store": 0, "arith_bounds": 1, "slide_run": 0, "persistent_blocks": 0, "persistent": 0, "prefetch": 1,
                "xcd_chunk": 32}
    named = {kv.split("=")[0] for v in variants for kv in v.split(",")}
    times = [[] for _ in variants]
    ok = [True] * len(variants)
    for rnd in range(int(shape.get("rounds", 5))):
        for i, v in enumerate(variants):
            opts = {k: defaults[k] for k in named if k in defaults}
            opts.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in v.split(",")})
            for d in devs:
                for k, val in opts.items():
                    d.set_option(k, val)
            for j in range(6):
                devs[j % copies].spmv_torch(xs[j % copies], out=ys[j % copies])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 150
            e0.record()
            for j in range(iters):
                devs[j % copies].spmv_torch(xs[j % copies], out=ys[j % copies])
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / iters * 1e3)
            ok[i] = ok[i] and bool(torch.equal(ys[0], yref))
    for v, t, good in zip(variants, times, ok):
        med = statistics.median(t)
        print(f"{v:48s} median {med:6.2f} us  min {min(t):6.2f} us  frac {B / med / 1e3 / 8000:5.3f}  bit-identical={good}  "
              f"rounds={[round(q, 1) for q in t]}", flush=True)
    print("plan:", {k: devs[0].describe().get(k) for k in ("kernel", "slide", "blocks", "ring_pages", "tile_steps", "uniform_row_fraction")}, flush=True)


@lab
def zoo():
    """A small zoo of sparsity patterns through the automatic plan (development tool): where are the cliffs?
    (was tools/lab_zoo.py)"""
    import os
    import sys
    import time

    import numpy as np

    import torch  # noqa: E402
    import spalinalg_amd as sp  # noqa: E402
    import spal_synth as synth  # noqa: E402

    def timeit(fn, iters=20, warm=3):
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

    def from_lens(lens, col_fn, rng):
        """rows with lens[r] entries, columns from col_fn(row index array (repeated), position in row, rng) then sorted / deduplicated per row"""
        n = lens.size
        rows = np.repeat(np.arange(n, dtype=np.int64), lens)
        pos = np.arange(rows.size, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
        cols = col_fn(rows, pos, rng)
        key = rows * (int(cols.max()) + 1) + cols
        key = np.unique(key)                      # sorts by (row, col), drops duplicates
        rows2, cols2 = key // (int(cols.max()) + 1), key % (int(cols.max()) + 1)
        lens2 = np.bincount(rows2, minlength=n)
        rp = np.concatenate([[0], np.cumsum(lens2)]).astype(np.uint64)
        return rp, cols2.astype(np.uint64), rng.uniform(-1, 1, cols2.size)

    def main():
        rng = np.random.default_rng(5)
        cases = []
        n = 4_000_000
        cases.append(("diagonal (1/row)", n, lambda: from_lens(np.ones(n, np.int64), lambda r, p, g: r, rng)))
        cases.append(("tridiagonal (3/row)", n, lambda: from_lens(np.full(n, 3, np.int64), lambda r, p, g: np.clip(r + p - 1, 0, n - 1), rng)))
        cases.append(("band 14/row, W=4096 (config-3-like)", n, lambda: from_lens(np.full(n, 14, np.int64), lambda r, p, g: np.clip(r - 2048 + g.integers(0, 4096, r.size), 0, n - 1), rng)))
        cases.append(("band 30/row, W=1024", n, lambda: from_lens(np.full(n, 30, np.int64), lambda r, p, g: np.clip(r - 512 + g.integers(0, 1024, r.size), 0, n - 1), rng)))
        m = 1_000_000
        cases.append(("band 100/row, W=2048", m, lambda: from_lens(np.full(m, 100, np.int64), lambda r, p, g: np.clip(r - 1024 + g.integers(0, 2048, r.size), 0, m - 1), rng)))
        cases.append(("band 400/row, W=4096", 250_000, lambda: from_lens(np.full(250_000, 400, np.int64), lambda r, p, g: np.clip(r - 2048 + g.integers(0, 4096, r.size), 0, 249_999), rng)))
        # power-law row lengths (graph-like), columns uniform
        pl = np.minimum((rng.pareto(1.6, 2_000_000) * 6 + 1).astype(np.int64), 5000)
        cases.append(("power-law rows (mean %.1f, max %d), uniform columns" % (pl.mean(), pl.max()), pl.size, lambda: from_lens(pl, lambda r, p, g: g.integers(0, pl.size, r.size), rng)))
        # power-law rows, local columns
        cases.append(("power-law rows, columns within +-5000", pl.size, lambda: from_lens(pl, lambda r, p, g: np.clip(r - 5000 + g.integers(0, 10000, r.size), 0, pl.size - 1), rng)))
        # block diagonal, dense 64 x 64 blocks
        nb = 1_000_000
        cases.append(("block-diagonal, dense 64 x 64 blocks", nb, lambda: from_lens(np.full(nb, 64, np.int64), lambda r, p, g: (r // 64) * 64 + p, rng)))
        # two bands far apart (coupled systems)
        cases.append(("two bands 7 + 7 per row, 2M columns apart", n, lambda: from_lens(np.full(n, 14, np.int64), lambda r, p, g: np.clip(np.where(p < 7, r - 100 + g.integers(0, 200, r.size), (r + 2_000_000) % n - 100 + g.integers(0, 200, r.size)), 0, n - 1), rng)))
        for name, nrows, make in cases:
            t0 = time.time()
            rp, ci, va = make()
            ncols = nrows
            dev = sp.CsrMatrix(nrows, ncols, rp, ci, va).device()
            x = torch.from_numpy(synth.vector(ncols)).cuda()
            y = torch.empty(nrows, dtype=torch.float64, device="cuda")
            t = timeit(lambda: dev.spmv_torch(x, out=y))
            plan = dev.autotune(x, y, iters=10)
            t2 = timeit(lambda: dev.spmv_torch(x, out=y))
            nnz = int(rp[-1])
            B = synth.spmv_bytes(nnz, nrows, nrows, ncols, 8)
            d = dev.describe()
            if d["kernel"] == "split":
                name = name + f" [split: {d['split_long_rows']} long rows]"
                d = d["short_part"]
            if d["kernel"] == "blockwin":
                print(f"{name:52s} nnz {nnz:>10d}  {t*1e3:8.1f} us -> autotuned {t2*1e3:8.1f} us = {100*B/(t2*1e-3)/8e12:5.1f} % of 8 TB/s  "
                      f"[blockwin rows={d['block_rows']} window={d['window_columns']}]  (host {time.time()-t0:.0f} s)", flush=True)
            else:
                print(f"{name:52s} nnz {nnz:>10d}  {t*1e3:8.1f} us -> autotuned {t2*1e3:8.1f} us = {100*B/(t2*1e-3)/8e12:5.1f} % of 8 TB/s  "
                      f"[{d['kernel']} rpt={d['rows_per_tile']} stream={d['stream_row_fraction']:.2f} lds={d['lds_row_fraction']:.2f} "
                      f"win={d['lds_window_bytes']//1024}K pers={d['persistent']}]  (host {time.time()-t0:.0f} s)", flush=True)
                # the block-window kernel by name on the same matrix (development: where else would it win?)
                dev.set_option("blockwin", 1)
                db = dev.describe()
                if db["kernel"] == "blockwin":
                    tb = timeit(lambda: dev.spmv_torch(x, out=y))
                    print(f"{'':52s}     block-window kernel by name: {tb*1e3:8.1f} us = {100*B/(tb*1e-3)/8e12:5.1f} %  [rows={db['block_rows']} window={db['window_columns']}]", flush=True)
            del dev

    if True:  # (was the script's __main__ block)
        main()


@lab
def cblock():
    """Column-blocked kernel (csr_cblock.hpp): sweep of the row-block height and the column-block width on matrices
    with uniform-random columns -- `lab.py cblock [nrows] [per_row]` (default 5M x 5M, 10 per row: the shape of the
    assembled config-5 matrix; 1000000 14 = config 2 with uniform columns).  us per product, stream kernels beside."""
    import numpy as np
    import torch
    import spalinalg_amd as sp
    import spal_synth as synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    per_row = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    window = int(os.environ.get("LAB_WINDOW", "0")) or n          # (columns within a window this wide around the diagonal; default: anywhere)
    form = int(os.environ.get("LAB_FORM", "-1"))                  # cblock_form: -1 auto, 0 entry-parallel, 1 rows form
    dt = np.float32 if os.environ.get("LAB_DTYPE", "f64") == "f32" else np.float64
    rp, ci, va = synth.banded_csr(n, n, per_row, window, synth.matrix_seed(2), dtype=dt)
    d = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    x = torch.from_numpy(synth.vector(n, dtype=dt)).cuda()
    y = torch.empty_like(x)

    def us(reps=30):
        for _ in range(5):
            d.spmv_torch(x, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            d.spmv_torch(x, out=y)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    d.set_option("cblock", 0)
    print(f"{n} x {n}, {per_row} per row, uniform columns: stream kernels {us():7.1f} us", flush=True)
    ref = y.clone()
    d.set_option("cblock", 1 if window != n else -1)
    d.set_option("cblock_rows", 0)
    d.set_option("cblock_form", form)
    p = d.describe()
    t = us()
    print(f"  automatic plan ({p.get('cblock_form')} form, {p.get('cblock_run')} entries per run): columns per block {p['cblock_cols']} ({p['cblock_col_blocks']} blocks), rows per block {p['cblock_rows']} "
          f"({p['cblock_row_blocks']} workgroups): {t:7.1f} us   same bits as the stream kernels: {bool(torch.equal(y, ref))}", flush=True)
    shifts = [int(v) for v in os.environ.get("LAB_SHIFTS", "15,16,17,18,19").split(",")]
    heights = [int(v) for v in os.environ.get("LAB_ROWS", "1024,2048,3072,4096").split(",")]
    for shift in shifts:
        for rows in heights:
            d.set_option("cblock_shift", shift)
            d.set_option("cblock_rows", rows)
            p = d.describe()
            if p["kernel"] != "cblock":
                print(f"  columns per block 2^{shift}, rows per block {rows}: does not qualify")
                continue
            t = us()
            print(f"  {p.get('cblock_form')} form, columns per block 2^{shift} ({p['cblock_col_blocks']} blocks), rows per block {rows} ({p['cblock_row_blocks']} workgroups): "
                  f"{t:7.1f} us   same bits as the stream kernels: {bool(torch.equal(y, ref))}", flush=True)


@lab
def csc_ypairs():
    """Config 4 (CSC scatter kernel), ONE handle: the product into 12 different allocations of y (1 GiB of other
    allocations between them) and with 6 allocations of x -- does the pair (matrix arrays, y) matter at this size too?"""
    import numpy as np
    import scipy.sparse as sps
    import torch
    import spalinalg_amd as sp
    import spal_synth as synth
    cfg = synth.CONFIGS[4]
    n, per_row = cfg["nrows"], cfg["per_row"]
    rp, ci, va = synth.banded_csr(n, n, per_row, cfg["window"], synth.matrix_seed(2))
    csc = sps.csr_matrix((va, ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n)).tocsc()
    csc.sort_indices()
    d = sp.CscMatrix(n, n, csc.indptr.astype(np.uint64), csc.indices.astype(np.uint64), csc.data).device()
    d.set_option("kernel", 1)
    xh = torch.from_numpy(synth.vector(n))
    spacers, ys, xs = [], [], []
    for _ in range(12):
        ys.append(torch.empty(n, dtype=torch.float64, device="cuda"))
        spacers.append(torch.empty(1 << 30, dtype=torch.uint8, device="cuda"))
    for _ in range(6):
        xs.append(xh.cuda())
        spacers.append(torch.empty(1 << 30, dtype=torch.uint8, device="cuda"))
    flush = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")     # 512 MB through the caches between products

    def us(xx, yy, reps=30):
        tot = 0.0
        for i in range(reps + 3):
            flush.add_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            d.spmv_torch(xx, out=yy)
            e1.record()
            torch.cuda.synchronize()
            if i >= 3:
                tot += e0.elapsed_time(e1)
        return tot / reps * 1e3
    print("y allocation:   " + "".join(f"{i:7d}" for i in range(len(ys))))
    print("scatter kernel: " + "".join(f"{us(xs[0], y):7.1f}" for y in ys))
    print("x allocation:   " + "".join(f"{us(xx, ys[0]):7.1f}" for xx in xs))


@lab
def cblock_once():
    """The column-blocked kernel's automatic plan on a uniform-column matrix, 30 products (for counter passes):
    `lab.py cblock_once [nrows] [per_row]`."""
    import torch
    import spalinalg_amd as sp
    import spal_synth as synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    per_row = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rp, ci, va = synth.banded_csr(n, n, per_row, n, synth.matrix_seed(2))
    d = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    x = torch.from_numpy(synth.vector(n)).cuda()
    y = torch.empty_like(x)
    for _ in range(30):
        d.spmv_torch(x, out=y)
    torch.cuda.synchronize()
    print(d.describe())


@lab
def bands_auto():
    """What the plan picks BY ITSELF for bands of 16 384 ... 262 144 columns (4M x 4M, 14 per row): kernel, form, us per product."""
    import torch
    import spalinalg_amd as sp
    import spal_synth as synth
    n = 4_000_000
    for W in (16384, 32768, 65536, 262144):
        rp, ci, va = synth.banded_csr(n, n, 14, W, synth.matrix_seed(2))
        d = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
        x = torch.from_numpy(synth.vector(n)).cuda()
        y = torch.empty_like(x)

        def us(reps=30):
            for _ in range(5):
                d.spmv_torch(x, out=y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                d.spmv_torch(x, out=y)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        p = d.describe()
        t = us()
        d.set_option("cblock", 0)
        t0 = us()
        print(f"band of {W:7d} columns: {p['kernel']:7s} {p.get('cblock_form') or '-':6s} nonlocal rows {p['nonlocal_row_fraction']:.3f} panel tiles {p['panel_tiles']:6d}  {t:7.1f} us"
              f"   (cblock = 0: {t0:7.1f} us)", flush=True)
        del d


@lab
def tiny():
    """BASELINE config 1 (10k x 10k, 100k triplets): the product on the assembled matrix, per plan form (us per launch)."""
    import torch
    import spalinalg_amd as sp
    import spal_synth as synth
    cfg = synth.CONFIGS[1]
    nr, nc, length = cfg["nrows"], cfg["ncols"], cfg["length"]
    r, c, v = synth.coo(nr, nc, length, synth.matrix_seed(1))
    csr = sp.CooMatrix.with_triplets(nr, nc, r, c, v).upload().assemble_csr()
    x = torch.from_numpy(synth.vector(nc)).cuda()
    y = torch.empty(nr, dtype=x.dtype, device="cuda")

    def us(reps=200):
        for _ in range(20):
            csr.spmv_torch(x, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            csr.spmv_torch(x, out=y)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    print("default plan:", {k: csr.describe()[k] for k in ("kernel", "slide", "rows_per_tile", "blocks", "persistent", "lds_x")}, f"{us():.2f} us")
    for opts in ([("window_pages", 8)], [("window_pages", 8), ("rows_per_tile", 16)], [("window_pages", 8), ("rows_per_tile", 8)], [("window_pages", 0), ("rows_per_tile", 0), ("slide_on", 0)], [("slide_on", 0), ("tiles_per_wave", 1), ("rows_per_tile", 64)], [("rows_per_tile", 16)], [("tiles_per_wave", 4), ("slide_on", 0), ("rows_per_tile", 32)], [("slide_on", 0), ("rows_per_tile", 16)], [("slide_on", 0), ("rows_per_tile", 8)],
                 [("kernel", 1)], [("kernel", 1), ("lanes_per_row", 8)], [("kernel", 1), ("lanes_per_row", 4), ("rows_per_block", 512)]):
        try:
            for k, val in opts:
                csr.set_option(k, val)
        except Exception as e:  # noqa: BLE001
            print(opts, "rejected:", e)
            continue
        d = csr.describe()
        print(opts, {k: d[k] for k in ("kernel", "slide", "rows_per_tile", "blocks", "lanes_per_row")}, f"{us():.2f} us", flush=True)


@lab
def csc_big():
    """CSC scatter over row tiles on a larger band (default 20M x 20M, 14 per row: 280M entries, 78K tiles): against the
    transposed route (bit-identical to the reference's order) and per launch -- `lab.py csc_big [n]`."""
    import numpy as np
    import scipy.sparse as sps
    import torch
    import spalinalg_amd as sp
    import spal_synth as synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(2))
    csc = sps.csr_matrix((va, ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n)).tocsc()
    csc.sort_indices()
    d = sp.CscMatrix(n, n, csc.indptr.astype(np.uint64), csc.indices.astype(np.uint64), csc.data).device()
    x = torch.from_numpy(synth.vector(n)).cuda()
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")

    def us(reps=20):
        for _ in range(3):
            d.spmv_torch(x, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            d.spmv_torch(x, out=y)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    B = synth.spmv_bytes(int(rp[-1]), n, n, n, 8)
    t2 = us()
    y2 = y.clone()
    d.set_option("kernel", 1)
    p = d.describe()
    t1 = us()
    print(f"{n} x {n}, {int(rp[-1])} entries: transposed route {t2:8.1f} us = {B / t2 / 8e6:.3f}; scatter over row tiles ({p['row_tiles']}, {p['row_tile_count']} tiles of "
          f"{p['row_tile_rows']} rows, x window {p['row_tile_x_window']}) {t1:8.1f} us = {B / t1 / 8e6:.3f}; max |diff| / max |y| = "
          f"{float((y - y2).abs().max() / y2.abs().max()):.2e}")
    for rows in (2048, 1024):
        d.set_option("row_tile_rows", rows)
        p = d.describe()
        y.fill_(float("nan"))
        t = us()
        print(f"  row tiles of {p['row_tile_rows']} rows ({p['row_tile_count']} tiles, x window {p['row_tile_x_window']}): {t:8.1f} us = {B / t / 8e6:.3f}; "
              f"max |diff| / max |y| = {float((y - y2).abs().max() / y2.abs().max()):.2e}")
    d.set_option("row_tiles", 0)
    y.fill_(float("nan"))
    t0 = us()
    print(f"  column tiles ({d.describe()['flush']}): {t0:8.1f} us = {B / t0 / 8e6:.3f}; max |diff| / max |y| = {float((y - y2).abs().max() / y2.abs().max()):.2e}")


def main():
    if len(sys.argv) < 2 or sys.argv[1] in ("--list", "-h", "--help") or sys.argv[1] not in LABS:
        for name, fn in sorted(LABS.items()):
            print(f"{name:14s} {(fn.__doc__ or '').strip().splitlines()[0]}")
        return 0 if len(sys.argv) > 1 and sys.argv[1] in ("--list", "-h", "--help") else 2
    name = sys.argv[1]
    sys.argv = [f"lab.py {name}"] + sys.argv[2:]
    LABS[name]()
    return 0


if __name__ == "__main__":
    sys.exit(main())
