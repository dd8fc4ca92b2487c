#!/usr/bin/env python3
"""Headline benchmark: CSR SpMV (f64) at 10M x 10M / 140M nnz, 1/2/4/8 GPUs.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the whole matrix with every input
already resident in HBM: each rank runs the HIP kernel on its row range.  With
N > 1 the timed region is, as north_star words it, {x broadcast once from rank
0 over RCCL -> K rank-local SpMVs -> the per-GPU y slices all-gathered once at
the end}; every rank then checks rows of its own shard in the gathered y.
--exchange allgather / halo / auto measure ITERATIVE use instead (something
moves after every step).  The matrix is fixed as N grows ("scaling":
"strong").  value = 2*nnz / (timed region / K), whole job; compute_only is the
max-over-ranks kernel time alone.

Rank 0 prints ONE JSON line.  `roofline` is computed from the ALGORITHMIC
bytes of the rank-local launch (SURVEY.md section 8d:
nnz*(8+4) + 4*(rows+1) + 8*ncols + 8*rows) and the launch's average duration
measured with HIP events on the launch stream; `cpu_baseline` times the CPU
oracle (single thread, like the reference) on the same matrix, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=3, choices=[1, 2, 3, 4, 5],
                    help="BASELINE config: 3 = CSR 10Mx10M/140M nnz (headline, the default line), "
                         "2 = CSR 1Mx1M/14M nnz, 4 = CSC scatter 1Mx1M, 5 = COO->CSR assembly 50M entries, "
                         "1 = the reference's CPU-sized case (10k x 10k, 100k triplets) end to end")
    ap.add_argument("--dist", default="banded", choices=["banded", "uniform", "ragged"],
                    help="column distribution: banded W=4096 (headline), uniform (stress row: columns anywhere), "
                         "ragged (robustness row of SURVEY 8d: banded columns, row length 1 + next() % 27)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ceiling", action="store_true",
                    help="skip the copy-ceiling child process (profiling runs: one process under the profiler)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--opt", action="append", default=[], help="kernel option key=value (tuning)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo only to rehearse the N > 1 path on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--copies", type=int, default=0,
                    help="single-GPU configs 2 and 4: rotate the launches over this many independent copies of "
                         "(matrix, x, y) so that the 188 MB working set is not served from the 256 MB Infinity "
                         "Cache (SURVEY 8d).  Default 0 = 3 copies for configs 2 and 4, 1 for config 3 (1.9 GB)")
    ap.add_argument("--exchange", default="end", choices=["end", "auto", "allgather", "halo"],
                    help="N > 1 only: what moves between the GPUs inside the timed region.  end (default) = "
                         "north_star literally: x broadcast once over RCCL, K rank-local SpMVs, the per-GPU y "
                         "slices all-gathered once at the end.  The others model ITERATIVE use (y of one step "
                         "is the x of the next, so something must move every step): allgather = the whole y "
                         "after every step; halo = per step every rank receives only the entries of y its rows "
                         "reference as columns (point-to-point; banded shards: +-W/2 from the neighbours), y "
                         "slices all-gathered once at the end; auto = halo when the halo is at most a quarter "
                         "of a slice and its plan succeeded on every rank, else allgather")
    ap.add_argument("--plain-collectives", action="store_true",
                    help="N > 1, --exchange end: use broadcast / all-gather even when every rank reads only a "
                         "window of x (default: rank 0 scatters the windows and gathers the y slices)")
    ap.add_argument("--host", default="torch", choices=["torch", "mg"],
                    help="who drives N > 1 GPUs: torch = one process per GPU over torch.distributed (the driver's "
                         "launch contract, spalinalg_amd/dist.py); mg = THIS one process through the C ABI's spal_mg_* "
                         "(what a single-process caller such as the Rust crate binds): python bench.py --host mg --gpus N")
    ap.add_argument("--devices", default="",
                    help="--host mg: comma-separated device list; repeats (e.g. 0,0,0,0) put several shards on one "
                         "GPU over the copy transport -- a rehearsal of the multi-GPU path on a 1-GPU box")
    ap.add_argument("--transport", default=None, choices=["rccl", "copy"], help="--host mg: exchange transport")
    ap.add_argument("--torch-vectors", action="store_true",
                    help="one GPU: x and y from torch's allocator instead of the handle's placed vectors (spal_csr_alloc_vectors)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run (config 3, banded, f64, one GPU): do not append the compact records of configs "
                         "1, 2, 4, 5 and of config 3 in f32 (child processes of this script, ~40 s)")
    ap.add_argument("--extras", action="store_true",
                    help="N > 1, --exchange end: also time K steps with an all-gather after every step and "
                         "report it beside the headline (extra collectives; off by default)")
    return ap.parse_args()


def timed(fn, steps, warmup, torch):
    """K calls of fn between HIP events on torch's current stream (= the launch stream)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def rotating(fns):
    """Call fns[0], fns[1], ... in turn, one per invocation."""
    state = {"i": 0}

    def call():
        fns[state["i"] % len(fns)]()
        state["i"] += 1
    return call


def copy_ceiling(nrows, per_row):
    """What the HBM of THIS box delivers for the stream kernel's footprint (tools/micro/stream_ceiling.hip, built by
    __graft_entry__.build(): the same loads -- 16-byte value pairs, 4-byte column pairs, the x window through LDS,
    y stores -- with no gather and no arithmetic), measured by a child process in the same run; plus the plain
    16-byte copy / read rates.  None when the binary is missing."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "micro", "stream_ceiling")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "--quick", str(nrows), str(per_row)], capture_output=True, text=True, timeout=180)
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001  (informational)
        return None


def traffic_entry(key):
    """HBM bytes per launch measured with PMC passes (profiles/traffic.json), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key, {}).get("hbm_bytes_per_launch")
    except Exception:  # noqa: BLE001
        return None


def bench_csc(args):
    """BASELINE config 4: CscMatrix f64 SpMV (atomic scatter path), CSC of the config-2 matrix."""
    import torch
    import scipy.sparse as sps
    import spalinalg_amd as sp
    import spal_synth as synth
    cfg = synth.CONFIGS[4]
    n, per_row = cfg["nrows"], cfg["per_row"]
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    esz = np.dtype(np_dt).itemsize
    rp, ci, va = synth.banded_csr(n, n, per_row, cfg["window"], synth.matrix_seed(2), dtype=np_dt)
    csc = sps.csr_matrix((va, ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n)).tocsc()
    csc.sort_indices()
    cp, ri, cv = csc.indptr.astype(np.uint64), csc.indices.astype(np.uint64), csc.data.astype(np_dt)
    m = sp.CscMatrix(n, n, cp, ri, cv)
    # 188 MB of matrix + vectors would sit in the 256 MB Infinity Cache: rotate over copies
    copies = args.copies if args.copies > 0 else 3
    devs = [m.device()] + [m.device_copy() for _ in range(copies - 1)]   # (handles with arrays of their own)
    for d in devs:
        for kv in args.opt:
            k, v = kv.split("=")
            d.set_option(k, int(v))
    dev = devs[0]
    xh = synth.vector(n, dtype=np_dt)
    xs = [torch.from_numpy(xh).cuda() for _ in range(copies)]
    ys = [torch.empty_like(xs[0]) for _ in range(copies)]
    x, y = xs[0], ys[0]
    # the library's default for CSC handles is the transposed route (CSC -> CSR once
    # on the device, then the CSR stream kernel); config 4 names the atomic scatter
    # path, so THAT is what `value` measures; the other route is reported beside it
    for d, xx, yy in zip(devs, xs, ys):
        d.set_option("kernel", 2)
        d.autotune(xx, yy, iters=30)
    launches = [(lambda d=d, xx=xx, yy=yy: d.spmv_torch(xx, out=yy)) for d, xx, yy in zip(devs, xs, ys)]
    ms_transposed = timed(rotating(launches), args.steps, args.warmup, torch)
    y_transposed = y.clone()
    for d in devs:
        d.set_option("kernel", 1)
    ms = timed(rotating(launches), args.steps, args.warmup, torch)
    nnz = n * per_row
    B = synth.spmv_bytes(nnz, n, n, n, esz)
    out = base_record(args, "CSC SpMV GFLOP/s (f64, 1Mx1M, 14M nnz, atomic scatter)",
                      synth.spmv_flops(nnz) / (ms * 1e-3) / 1e9, "GFLOP/s", ms,
                      f"CscMatrix {args.dtype} SpMV y=A*x by atomic scatter, {n}x{n}, {nnz} nnz, CSC of the "
                      f"config-2 banded matrix (BASELINE configs[3]), single GPU; launches rotate over {copies} "
                      f"independent copies of (A, x, y) = {copies * B / 1e6:.0f} MB > the 256 MB Infinity Cache",
                      dev.describe())
    out["roofline"] = {"bound": "hbm", "achieved": round(B / (ms * 1e-3) / 1e9, 2), "peak": 8000.0,
                       "unit": "GB/s", "frac": round(B / (ms * 1e-3) / 8e12, 4),
                       "traffic": traffic_entry("config4_scatter_f64") if args.dtype == "f64" else None,
                       "kernel": ("csc_spmv_rowtiles (LDS-privatised atomic scatter over row tiles: no memset, no global atomics, no hand-off)"
                                  if dev.describe().get("row_tiles") else
                                  "csc_spmv_scatter" + (" (neighbour hand-off: no memset, no global atomics)"
                                                        if dev.describe().get("flush") == "neighbour_handoff" else " (+ y zero fill)")),
                       "kernel_ms": round(ms, 6),
                       "algorithmic_bytes_per_launch": B,
                       "note": "LDS float atomics per entry; row tiles: every row of y belongs to one workgroup and is stored once "
                               "(column tiles, option row_tiles = 0: rows shared with the neighbouring super-tile updated behind a flag)"}
    out["transposed_route"] = {"ms_per_step": round(ms_transposed, 6),
                               "gflops": round(synth.spmv_flops(nnz) / (ms_transposed * 1e-3) / 1e9, 2),
                               "roofline_frac": round(B / (ms_transposed * 1e-3) / 8e12, 4),
                               "agrees_with_scatter": bool(torch.allclose(y, y_transposed, rtol=1e-10, atol=1e-12)),
                               "note": "kernel=2 (library default): device CSC->CSR once, then csr_spmv_stream; "
                                       "deterministic, bit-identical to the reference's k-ascending order"}
    if not args.no_cpu_baseline:
        import oracle  # CPU baseline leg only
        t0, passes = time.perf_counter(), 0
        while True:
            yh = oracle.csc_spmv(n, cp, ri, cv, xh)
            passes += 1
            el = time.perf_counter() - t0
            if el >= args.cpu_seconds or passes >= 2000:
                break
        out["cpu_baseline"] = {"value": round(synth.spmv_flops(nnz) * passes / el / 1e9, 4), "unit": "GFLOP/s",
                               "cores": 1, "kind": "port",
                               "sample": f"{passes} full passes over the same matrix in {el:.1f} s, 1 thread",
                               "gpu_agrees_with_cpu": bool(np.allclose(y.cpu().numpy(), yh, rtol=1e-10 if esz == 8 else 1e-4,
                                                                       atol=1e-12 if esz == 8 else 1e-5))}
    print(json.dumps(out))


def bench_small(args):
    """BASELINE config 1 (the reference's own CPU-runnable case): 10k x 10k, 100k random triplets ->
    CsrMatrix on the device -> SpMV; the CPU baseline leg runs the same on one core and compares in full."""
    import torch
    import spalinalg_amd as sp
    import spal_synth as synth
    cfg = synth.CONFIGS[1]
    nr, nc, length = cfg["nrows"], cfg["ncols"], cfg["length"]
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    r, c, v = synth.coo(nr, nc, length, synth.matrix_seed(1), dtype=np_dt)
    d = sp.CooMatrix.with_triplets(nr, nc, r, c, v).upload()
    csr = d.assemble_csr()
    x = torch.from_numpy(synth.vector(nc, dtype=np_dt)).cuda()
    y = torch.empty(nr, dtype=x.dtype, device="cuda")
    ms = timed(lambda: csr.spmv_torch(x, out=y), args.steps, args.warmup, torch)
    t0 = time.perf_counter()
    for _ in range(20):
        d.assemble_csr().close()
    torch.cuda.synchronize()
    asm_ms = (time.perf_counter() - t0) * 1e3 / 20
    rp, ci, va = csr.download()
    nnz = int(rp[-1])
    B = synth.spmv_bytes(nnz, nr, nr, nc, np.dtype(np_dt).itemsize)
    out = base_record(args, f"CSR SpMV GFLOP/s ({args.dtype}, config 1)", synth.spmv_flops(nnz) / (ms * 1e-3) / 1e9,
                      "GFLOP/s", ms, f"CooMatrix {length} triplets -> CsrMatrix {nr}x{nc} ({nnz} stored) -> y=A*x "
                      f"(BASELINE configs[0], the reference's CPU-sized case; launch-bound on a GPU), single GPU",
                      csr.describe())
    out["roofline"] = {"bound": "hbm", "achieved": round(B / (ms * 1e-3) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                       "frac": round(B / (ms * 1e-3) / 8e12, 5), "traffic": None,
                       "kernel": ("csr_spmv_slide" if csr.describe().get("slide") and csr.describe()["kernel"] == "stream"
                                  else "csr_spmv_" + csr.describe()["kernel"]),
                       "kernel_ms": round(ms, 6), "algorithmic_bytes_per_launch": B,
                       "note": "1.4 MB of work: the launch itself is the cost"}
    out["assembly_ms"] = round(asm_ms, 4)
    if not args.no_cpu_baseline:
        import oracle  # CPU baseline leg only
        t0 = time.perf_counter()
        p, i, w = oracle.coo_to_csr(nr, nc, r, c, v)
        asm_cpu_ms = (time.perf_counter() - t0) * 1e3
        bits = np.uint64 if np_dt == np.float64 else np.uint32
        xh = x.cpu().numpy()
        t0, passes = time.perf_counter(), 0
        while time.perf_counter() - t0 < 2.0:
            yh = oracle.csr_spmv(p, i, w, xh)
            passes += 1
        el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(synth.spmv_flops(nnz) * passes / el / 1e9, 4), "unit": "GFLOP/s", "cores": 1,
                               "kind": "port", "sample": f"{passes} passes of the same matrix in {el:.1f} s, 1 thread; "
                                                         f"assembly of the {length} triplets on the CPU: {asm_cpu_ms:.2f} ms",
                               "gpu_assembly_equals_cpu_bit_for_bit": bool(np.array_equal(rp, p) and np.array_equal(ci, i)
                                                                          and np.array_equal(va.view(bits), w.view(bits))),
                               "gpu_agrees_with_cpu": bool(np.allclose(y.cpu().numpy(), yh, rtol=1e-10 if np_dt == np.float64 else 1e-4,
                                                                       atol=1e-12 if np_dt == np.float64 else 1e-5))}
    print(json.dumps(out))


def bench_coo(args):
    """BASELINE config 5: CooMatrix -> CsrMatrix on the device, 50M random triplets
    (+1 % duplicates, +0.1 % cancelling pairs) into 5M x 5M, then one SpMV on the result."""
    import torch
    import spalinalg_amd as sp
    import spal_synth as synth
    cfg = synth.CONFIGS[5]
    nr, length = cfg["nrows"], cfg["length"]
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    esz = np.dtype(np_dt).itemsize
    r, c, v = synth.coo(nr, nr, length, synth.matrix_seed(5), cfg["dup_permille"], cfg["cancel_permille"],
                           dtype=np_dt)
    coo = sp.CooMatrix.with_triplets(nr, nr, r, c, v)
    d = coo.upload()      # triplets + workspace resident in HBM before the timed region
    torch.cuda.synchronize()
    steps = max(1, min(args.steps, 20))
    warm = max(1, min(args.warmup, 3))
    for _ in range(warm):
        d.assemble_csr().close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per_step = []
    for _ in range(steps):
        ts = time.perf_counter()
        csr = d.assemble_csr()      # includes its one host sync; the result is the complete CSR matrix (round 4: the product
                                    # kernels' plan is built by the first product / spal_csr_plan -- timed below, reported beside)
        nnz = csr.shape()[2]
        if _ + 1 < steps:
            csr.close()
        per_step.append((time.perf_counter() - ts) * 1e3)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    if os.environ.get("SPAL_BENCH_DEBUG"):
        print("per-step ms:", [round(t, 2) for t in per_step], file=sys.stderr)
    # the assembled matrix multiplies (the config's second half).  Setup of the product on the result, each timed once with
    # the host clock around call + synchronise: the kernels' plan (spal_csr_plan), then the first product (which builds the
    # column-blocked copy where the plan wants one).
    x = torch.from_numpy(synth.vector(nr, dtype=np_dt)).cuda()
    y = torch.empty_like(x)
    warm_up = d.assemble_csr()      # (the planner's kernels run for the first time in this process: not what is timed)
    warm_up.plan()
    warm_up.close()
    torch.cuda.synchronize()
    ts = time.perf_counter()
    csr.plan()
    torch.cuda.synchronize()
    plan_ms = (time.perf_counter() - ts) * 1e3
    ts = time.perf_counter()
    csr.spmv_torch(x, out=y)
    torch.cuda.synchronize()
    first_ms = (time.perf_counter() - ts) * 1e3
    spmv_ms = timed(lambda: csr.spmv_torch(x, out=y), 20, 3, torch)
    lb = synth.assembly_bytes(length, nnz, nr, esz)
    args.steps, args.warmup = steps, warm
    out = base_record(args, "COO->CSR assembly Mentries/s (f64, 50M triplets into 5Mx5M)",
                      length / (ms * 1e-3) / 1e6, "Mentries/s", ms,
                      f"CooMatrix->CsrMatrix device assembly, {length} random triplets (+1% duplicates, +0.1% "
                      f"cancelling pairs) into {nr}x{nr} -> {nnz} stored entries (BASELINE configs[4]), single GPU",
                      csr.describe())
    out["roofline"] = {"bound": "hbm", "achieved": round(lb / (ms * 1e-3) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                       "frac": round(lb / (ms * 1e-3) / 8e12, 4),
                       "traffic": traffic_entry("config5_assembly_f64") if args.dtype == "f64" else None,
                       "kernel": "radix_hist + radix_scatter, twice (the second writes column | row-in-group in one word) + group_offsets "
                                 "+ coo_group_sort (look-back placement): the whole assembly call, its host synchronisation included",
                       "kernel_ms": round(ms, 6), "algorithmic_bytes_per_launch": lb,
                       "note": "algorithmic = lower bound 16*len + 12*nnz_out + 4*(nrows+1); a multi-pass sort "
                               "inherently moves several times this",
                       "route": d.describe()}
    rd = csr.describe()
    rb = synth.spmv_bytes(nnz, nr, nr, nr, esz)
    out["product_plan_ms"] = round(plan_ms, 4)
    out["ms_per_step_plus_product_plan"] = round(ms + plan_ms, 6)
    out["spmv_on_result"] = {"ms": round(spmv_ms, 6), "plan_ms": round(plan_ms, 4), "first_product_ms": round(first_ms, 4),
                             "gflops": round(synth.spmv_flops(nnz) / (spmv_ms * 1e-3) / 1e9, 2),
                             "kernel": "csr_spmv_" + rd["kernel"] + ("_rows" if rd.get("cblock_form") == "rows" else ""),
                             "algorithmic_bytes_per_launch": rb, "roofline_frac": round(rb / (spmv_ms * 1e-3) / 8e12, 4),
                             "traffic": traffic_entry("config5_result_spmv_f64") if args.dtype == "f64" else None,
                             "plan": {k: rd.get(k) for k in ("cblock_form", "cblock_run", "cblock_rows", "cblock_cols", "cblock_col_blocks", "cblock_row_blocks")}}
    if not args.no_cpu_baseline:
        import oracle  # CPU baseline leg only
        sample = min(length, 50_000_000)   # the whole config-5 input: 5-15 s on one core
        t0 = time.perf_counter()
        p, i, w = oracle.coo_to_csr(nr, nr, r[:sample], c[:sample], v[:sample])
        el = time.perf_counter() - t0
        # the oracle's result for the WHOLE input is in hand: compare the GPU's CSR with it bit for bit
        grp, gci, gva = csr.download()
        bits = np.uint64 if np_dt == np.float64 else np.uint32
        exact = bool(sample == length and np.array_equal(grp, p) and np.array_equal(gci, i)
                     and np.array_equal(gva.view(bits), w.view(bits)))
        out["cpu_baseline"] = {"value": round(sample / el / 1e6, 3), "unit": "Mentries/s", "cores": 1, "kind": "port",
                               "sample": f"{'all' if sample == length else 'the first ' + str(sample) + ' of the'} {length} triplets "
                                         f"(same {nr}x{nr} shape) in {el:.1f} s, "
                                         f"1 thread (restatement of src/csr/conv/coo.rs:4-115)",
                               "gpu_assembly_equals_cpu_bit_for_bit": exact,
                               "compared": f"rowptr ({grp.size}), colind ({gci.size}) and the bit patterns of values "
                                           f"({gva.size}) of the full-size result"}
        if not exact:
            print(json.dumps(out))
            sys.exit("config 5: the GPU assembly differs from the CPU oracle")
    print(json.dumps(out))


def steady(fn, sync, reps=12):
    """median and min wall time (ms) of fn() + sync() over `reps` calls after two warm-up calls"""
    for _ in range(2):
        fn()
        sync()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        sync()
        out.append((time.perf_counter() - t0) * 1e3)
    out.sort()
    return out[len(out) // 2], out[0]


def bench_mg(args):
    """--host mg: the row-partitioned product driven from ONE process through spal_mg_* (C ABI): x windows
    scattered from GPU 0 (grouped ncclSend/ncclRecv) -> K local SpMVs -> y slices gathered on GPU 0."""
    import spalinalg_amd as sp
    import spal_synth as synth
    if sp.device_count() < 1:
        sys.exit("bench.py needs a GPU: libspal_hip has no CPU fallback")
    cfg = synth.CONFIGS[args.config]
    nrows, ncols, per_row = cfg["nrows"], cfg["ncols"], cfg["per_row"]
    window = ncols if args.dist == "uniform" else cfg["window"]
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    esz = np.dtype(np_dt).itemsize
    devices = [int(d) for d in args.devices.split(",")] if args.devices else None
    G = len(devices) if devices else args.gpus
    virtual = devices is not None and len(set(devices)) < len(devices)
    t0 = time.time()
    if args.dist == "ragged":
        rp, ci, va = synth.ragged_csr(nrows, ncols, cfg["window"], synth.matrix_seed(args.config), dtype=np_dt)
    else:
        rp, ci, va = synth.banded_csr(nrows, ncols, per_row, window, synth.matrix_seed(args.config), dtype=np_dt)
    nnz = int(rp[-1])
    xh = synth.vector(ncols, dtype=np_dt)
    t_gen = time.time() - t0
    t0 = time.time()
    mg = sp.MultiGpuCsr(sp.CsrMatrix._trusted(nrows, ncols, rp, ci, va), G, devices=devices, transport=args.transport)
    t_upload = time.time() - t0
    eb = mg.exchange_bytes()
    mg.set_x(xh)
    windows_pay = eb["x_scatter"] * 4 <= (G - 1) * ncols * esz * 3
    dist_x = mg.scatter_x if (windows_pay and not args.plain_collectives) else mg.broadcast_x
    # warm-up, then the timed region: x once -> K local products -> y once
    dist_x()
    for _ in range(args.warmup):
        mg.spmv_local()
    mg.gather_y()
    mg.synchronize()
    t0 = time.perf_counter()
    dist_x()
    for _ in range(args.steps):
        mg.spmv_local()
    mg.gather_y()
    mg.synchronize()
    total_ms = (time.perf_counter() - t0) * 1e3
    ms_per_step = total_ms / args.steps
    y = mg.y_gathered()
    # steady-state pieces: wall clock around call + synchronize, and the library's HIP events (longest over the GPUs)
    comm = {}
    for name, fn, phase in (("x_distribution", dist_x, "x_distribution"), ("y_collection", mg.gather_y, "y_collection")):
        med, mn = steady(fn, mg.synchronize)
        comm[name] = {"wall_ms_median": round(med, 4), "wall_ms_min": round(mn, 4), "event_ms": round(mg.timing()[phase], 4)}
    med, mn = steady(mg.spmv_local, mg.synchronize, reps=30)
    compute = {"wall_ms_median": round(med, 4), "wall_ms_min": round(mn, 4), "event_ms": round(mg.timing()["compute"], 4)}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mg.spmv_local()
    mg.synchronize()
    compute["ms_per_step_back_to_back"] = round((time.perf_counter() - t0) * 1e3 / args.steps, 6)
    # north_star's wording to the letter, measured the same way: the WHOLE vector broadcast once (ncclBroadcast), K local
    # products, the slices all-gathered at the end (the K-th product is the one that all-gathers)
    mg.set_x(xh)
    mg.broadcast_x()
    mg.spmv_resident()
    mg.synchronize()
    t0 = time.perf_counter()
    mg.broadcast_x()
    for _ in range(args.steps - 1):
        mg.spmv_local()
    mg.spmv_resident()
    mg.synchronize()
    bcast_total_ms = (time.perf_counter() - t0) * 1e3
    y_all = mg.y_allgathered()
    allgather_equals_gather = bool(np.array_equal(y_all.view(np.uint64 if esz == 8 else np.uint32),
                                                  y.view(np.uint64 if esz == 8 else np.uint32)))
    halo = None
    if nrows == ncols:
        mg.set_x(xh)
        dist_x()
        med, mn = steady(mg.spmv_halo, mg.synchronize, reps=20)
        tm = mg.timing()
        halo = {"step_wall_ms_median": round(med, 4), "step_wall_ms_min": round(mn, 4), "exchange_event_ms": round(tm["halo"], 4),
                "bytes_per_step": eb["halo"]}
    # cpu_baseline leg: the oracle timed on one core, and GPU 0's gathered y compared with it (bit for bit on the stream path)
    exact, cpu = None, None
    if not args.no_cpu_baseline:
        import oracle  # CPU baseline leg only
        t0 = time.perf_counter()
        yh = oracle.csr_spmv(rp, ci, va, xh)
        el = time.perf_counter() - t0
        bits = np.uint64 if esz == 8 else np.uint32
        exact = bool(np.array_equal(y.view(bits), yh.view(bits)))
        if not exact:
            bound = oracle.csr_abs_bound(rp, ci, va, xh)
            tol = 1e-10 if esz == 8 else 1e-4
            if not np.all(np.abs(y.astype(np.float64) - yh.astype(np.float64)) <= tol * bound + 1e-300):
                sys.exit("--host mg: the gathered y differs from the CPU oracle")
        cpu = {"value": round(synth.spmv_flops(nnz) / el / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": "port",
               "sample": f"one full pass over the same matrix in {el:.2f} s, 1 thread, 64-bit indices",
               "gpu_equals_cpu_bit_for_bit": exact}
    whole_bytes = synth.spmv_bytes(nnz, nrows, nrows, ncols, esz)
    kern_ms = compute["ms_per_step_back_to_back"]
    out = {
        "metric": "CSR SpMV GFLOP/s (f64, 10Mx10M, 140M nnz)" if args.config == 3 and esz == 8 and args.dist == "banded"
                  else f"CSR SpMV GFLOP/s ({args.dtype}, config {args.config}, {args.dist})",
        "value": round(synth.spmv_flops(nnz) / (ms_per_step * 1e-3) / 1e9, 3), "unit": "GFLOP/s",
        "n_gpus": len(set(devices)) if devices else G, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 6), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"CsrMatrix {args.dtype} SpMV y=A*x, {nrows}x{ncols}, {nnz} nnz, {args.dist} columns, rows "
                               f"partitioned into {G} shards" + (f" on devices {devices} (virtual shards: a rehearsal)" if virtual else f" over {G} GPUs")
                               + f", ONE process through the C ABI (spal_mg_*, transport {mg.transport}): x "
                               + ("windows scattered from GPU 0" if dist_x == mg.scatter_x else "broadcast from GPU 0")
                               + ", K local SpMVs, y slices gathered on GPU 0 -- all inside the timed region",
                   "host": "mg", "shards": G, "transport": mg.transport, "partition": [int(b) for b in mg.partition()],
                   "exchange_bytes": eb},
        "algorithmic_bytes_per_step": whole_bytes,
        "roofline": {"bound": "hbm", "achieved": round(whole_bytes / (kern_ms * 1e-3) / 1e9 / max(1, (len(set(devices)) if devices else G)), 2),
                     "peak": 8000.0, "unit": "GB/s",
                     "frac": round(whole_bytes / (kern_ms * 1e-3) / 8e12 / max(1, (len(set(devices)) if devices else G)), 4), "traffic": None,
                     "kernel": "csr_spmv_stream / csr_spmv_slide per shard", "kernel_ms": kern_ms,
                     "note": "per GPU: the whole matrix's algorithmic bytes / the shards' concurrent kernels / the GPUs"},
        "end_to_end_windows_ms": round(total_ms, 4) if dist_x == mg.scatter_x else None,
        "end_to_end_broadcast_allgather_ms": round(bcast_total_ms, 4),
        "end_to_end_broadcast_allgather_value": round(synth.spmv_flops(nnz) * args.steps / (bcast_total_ms * 1e-3) / 1e9, 3),
        "allgather_equals_gather_bit_for_bit": allgather_equals_gather,
        "compute_only": compute, "comm_ms": comm, "halo": halo,
        "efficiency_inputs": {"total_ms": round(total_ms, 4), "K": args.steps,
                              "note": "total = x_distribution + K * compute + y_collection: recompute for any K"},
        "cpu_baseline": cpu,
        "setup_s": {"generate": round(t_gen, 2), "validate_narrow_upload_plan": round(t_upload, 2)},
    }
    print(json.dumps(out))
    mg.close()



def add_hbm_frac(roofline, peak_gbs=8000.0):
    """`frac` prices the ALGORITHMIC bytes (32-bit indices, every pointer read) -- the contract's definition; a
    kernel that moves fewer bytes than that (16-bit columns, no row pointers) can exceed 1 by it (config 3 in f32: 1.04).
    `hbm_frac` = the bytes the HBM really moved per launch (rocprofv3 PMC passes, profiles/traffic.json) / the launch's
    duration / the peak: the fraction of the hardware's rate, never above 1.  `frac_to_quote` names the one to read."""
    t, ms = roofline.get("traffic"), roofline.get("kernel_ms")
    if t and ms:
        roofline["hbm_frac"] = round(t / (ms * 1e-3) / (peak_gbs * 1e9), 4)
    roofline["frac_to_quote"] = "hbm_frac" if (roofline.get("frac") or 0) > 1.0 and roofline.get("hbm_frac") else "frac"
    return roofline


def other_configs(args):
    """Compact records of the BASELINE configs the default line does not measure (1, 2, 4, 5) and of config 3 in f32:
    each is this script run as a child process (`--config N`, its own CPU-baseline leg on a short budget), reduced to
    the fields a reader needs to recompute its roofline fraction from profiles/r03/kernel_stats_configN.csv."""
    import subprocess
    runs = {"1": ["--config", "1", "--steps", "200", "--warmup", "20"],
            "2": ["--config", "2", "--steps", "200", "--warmup", "20", "--cpu-seconds", "3", "--no-ceiling"],
            "4": ["--config", "4", "--steps", "200", "--warmup", "20", "--cpu-seconds", "3"],
            "5": ["--config", "5", "--steps", "10", "--warmup", "2"],
            "3_f32": ["--config", "3", "--dtype", "f32", "--steps", str(args.steps), "--warmup", str(args.warmup),
                      "--cpu-seconds", "3", "--no-ceiling", "--no-other-configs"]}
    out = {}
    for name, extra in runs.items():
        t0 = time.perf_counter()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1"] + extra,
                               capture_output=True, text=True, timeout=240)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not line:
                out[name] = {"error": (p.stderr or p.stdout)[-400:], "rc": p.returncode}
                continue
            r = json.loads(line[-1])
        except Exception as exc:  # noqa: BLE001  (the headline must still be printed)
            out[name] = {"error": str(exc)}
            continue
        rf, cb = r.get("roofline", {}), r.get("cpu_baseline") or {}
        rec = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
               "steps": r["steps"], "dtype": r["dtype"], "workload": r["config"]["workload"],
               "roofline": add_hbm_frac({k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel",
                                                                "kernel_ms", "algorithmic_bytes_per_launch", "moved_bytes_per_launch",
                                                                "moved_frac")}),
               "cpu_baseline": {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")} if cb else None,
               "parity": {k: v for k, v in cb.items() if k.startswith("gpu_")} or None,
               "wall_s": round(time.perf_counter() - t0, 1)}
        for k in ("transposed_route", "spmv_on_result", "assembly_ms", "product_plan_ms", "ms_per_step_plus_product_plan"):
            if k in r:
                rec[k] = r[k]
        out[name] = rec
    return out

def base_record(args, metric, value, unit, ms, workload, plan):
    return {"metric": metric, "value": round(value, 3), "unit": unit, "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 6), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload, "plan": plan}}


def main():
    args = parse()
    if args.config in (1, 4, 5):
        if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
            sys.exit("configs 1, 4 and 5 are single-GPU (BASELINE.json); only the CSR configs shard over GPUs")
        import torch
        import spalinalg_amd as sp
        import spal_synth as synth
        if not torch.cuda.is_available() or sp.device_count() < 1:
            sys.exit("bench.py needs a GPU: libspal_hip has no CPU fallback")
        return bench_small(args) if args.config == 1 else bench_csc(args) if args.config == 4 else bench_coo(args)
    if args.host == "mg":
        if int(os.environ.get("WORLD_SIZE", "1")) != 1:
            sys.exit("--host mg is ONE process driving every GPU: run it without torch.distributed.run")
        return bench_mg(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                     "(one process per GPU); see the module docstring")
        sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist
    import spalinalg_amd as sp
    import spal_synth as synth
    from spalinalg_amd.dist import RowPartitionedSpmv, even_rows

    if not torch.cuda.is_available() or sp.device_count() < 1:
        sys.exit("bench.py needs a GPU: libspal_hip has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    cfg = synth.CONFIGS[args.config]
    nrows, ncols, per_row = cfg["nrows"], cfg["ncols"], cfg["per_row"]
    window = ncols if args.dist == "uniform" else cfg["window"]
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    t_dt = torch.float64 if args.dtype == "f64" else torch.float32
    esz = 8 if args.dtype == "f64" else 4

    def gen_rows(a, b):
        if args.dist == "ragged":
            return synth.ragged_csr(nrows, ncols, window, synth.matrix_seed(args.config), dtype=np_dt, rows=(a, b))
        return synth.banded_csr(nrows, ncols, per_row, window, synth.matrix_seed(args.config), dtype=np_dt, rows=(a, b))

    # ---- the rank's shard: rows [r0, r1), generated on the host, uploaded through the C ABI
    t0 = time.time()
    if args.dist == "ragged":   # rows of 1 ... 27 entries: ranges with balanced stored entries
        from spalinalg_amd.dist import partition_rows
        rp_all = synth.ragged_rowptr(nrows, synth.matrix_seed(args.config))
        nnz = int(rp_all[-1])
        bounds = partition_rows(rp_all, world)
        del rp_all
    else:
        nnz = nrows * per_row
        bounds = even_rows(nrows, world)  # same entries in every row: even rows == even entries
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    rp, ci, va = gen_rows(r0, r1)
    t_gen = time.time() - t0
    t0 = time.time()
    shard = sp.CsrMatrix._trusted(r1 - r0, ncols, rp, ci, va)   # generator output is valid by construction;
    dev = shard.device(local_rank)                               # create() re-validates it anyway
    t_upload = time.time() - t0
    for kv in args.opt:
        k, v = kv.split("=")
        dev.set_option(k, int(v))
    plan = dev.describe()

    op = RowPartitionedSpmv.from_shard(dev, bounds, rank, world, device)
    exchange = "none" if world == 1 else args.exchange
    if world > 1 and exchange in ("auto", "halo"):
        ok = 1
        try:
            if nrows != ncols:
                raise ValueError("halo exchange needs a square matrix (y feeds back as x)")
            op.plan_halo(int(ci.min()), int(ci.max()) + 1)
            small = op.halo_bytes * 4 <= (r1 - r0) * esz
        except Exception as exc:  # noqa: BLE001  (every rank must still reach the agreement below)
            ok, small = 0, False
            print(f"[rank {rank}] halo plan failed: {exc}", file=sys.stderr)
        agree = torch.tensor([ok, int(small)], dtype=torch.int32, device=device)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree[0]) == 0:
            if args.exchange == "halo":
                sys.exit("--exchange halo: the halo plan failed (see stderr)")
            exchange = "allgather"
        elif args.exchange == "auto":
            exchange = "halo" if int(agree[1]) else "allgather"

    # ---- x: generated on rank 0, broadcast once over RCCL.  One GPU: x and y are the handle's own vectors
    # (spal_csr_alloc_vectors: placed so that the stores of y do not collide with the matrix stream)
    y_placed = None
    t_vectors = 0.0
    if world == 1 and not args.torch_vectors:
        t0 = time.time()
        x, y_placed = dev.vectors_torch()
        t_vectors = time.time() - t0
        x.copy_(torch.from_numpy(synth.vector(ncols, dtype=np_dt)))
    elif rank == 0:
        x = torch.from_numpy(synth.vector(ncols, dtype=np_dt)).to(device)
    else:
        x = torch.zeros(ncols, dtype=t_dt, device=device)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    tb = time.perf_counter()
    op.broadcast_x(x)
    torch.cuda.synchronize()
    x_bcast_ms = (time.perf_counter() - tb) * 1e3 if world > 1 else 0.0
    y = y_placed if y_placed is not None else torch.empty(nrows, dtype=t_dt, device=device)
    if exchange == "halo":
        # dry run of one halo step; any rank failing sends everyone to the all-gather
        ok = 1
        try:
            op.spmv_halo(x, y)
            torch.cuda.synchronize()
            # ... and its result is checked against the all-gather of the same kernel's
            # slices on the columns this rank's rows reference (must be bit-identical)
            y_ag = torch.empty_like(y)
            op.spmv(x, y_ag)
            torch.cuda.synchronize()
            lo, hi = int(ci.min()), int(ci.max()) + 1
            if not torch.equal(y[lo:hi], y_ag[lo:hi]):
                raise RuntimeError("halo exchange disagrees with the all-gather on the needed columns")
            del y_ag
        except Exception as exc:  # noqa: BLE001
            ok = 0
            print(f"[rank {rank}] halo exchange failed: {exc}", file=sys.stderr)
        agree = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree[0]) == 0:
            if args.exchange == "halo":
                sys.exit("--exchange halo: the halo exchange failed (see stderr)")
            exchange = "allgather"
    # --exchange end: "x broadcast once ... y slices gathered at the end".  When every rank reads only
    # a window of x (banded shards: its slice +- W/2) rank 0 scatters the windows instead of
    # broadcasting the whole vector, and the slices are gathered on rank 0 instead of all-gathered:
    # 1/N of the bytes per link either way.  One verified dry run; any rank failing -> plain collectives.
    x_mode, y_mode, x_needs = "broadcast", "allgather", None
    if world > 1 and exchange == "end" and not args.plain_collectives:
        ok = 1
        try:
            x_needs = op.plan_x_windows(int(ci.min()), int(ci.max()) + 1)
            if op.xw_len * 4 > ncols * 3 or not op.equal:
                raise ValueError("windows too wide / unequal slices: plain collectives")
            probe = x if rank == 0 else torch.full_like(x, float("nan"))
            a0, a1 = op.distribute_x(probe, ncols, x_needs)
            torch.cuda.synchronize()
            if not (a0 <= int(ci.min()) and int(ci.max()) < a1 and torch.equal(probe[a0:a1], x[a0:a1])):
                raise RuntimeError("scattered x window differs from the broadcast vector")
            del probe
            op.local_only(x)
            y_ag = torch.empty_like(y)
            op.gather_y(y_ag)
            op.gather_y_root(y)
            torch.cuda.synchronize()
            if rank == 0 and not torch.equal(y, y_ag):
                raise RuntimeError("gather on rank 0 differs from the all-gather")
            del y_ag
        except Exception as exc:  # noqa: BLE001  (every rank must still reach the agreement below)
            ok = 0
            print(f"[rank {rank}] scatter/gather plan not used: {exc}", file=sys.stderr)
        agree = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree[0]):
            x_mode, y_mode = "scatter_windows", "gather_root"
    # setup: let the library pick between its kernel variants on this device (results are identical)
    t0 = time.time()
    plan = dev.autotune(x, y if world == 1 else op.y_local[: r1 - r0], iters=30)   # (on the vectors the timed steps use)
    t_autotune = time.time() - t0
    # config 2 on one GPU: 188 MB would be served from the 256 MB Infinity Cache, so the
    # launches rotate over independent copies of (A, x, y); config 3 is 1.9 GB
    copies = args.copies if args.copies > 0 else (3 if (args.config == 2 and world == 1) else 1)
    if world > 1:
        copies = 1
    single = [lambda: dev.spmv_torch(x, out=y)]     # y is the rank's (= the whole) slice
    keep = []
    for _ in range(copies - 1):
        d2 = shard.device_copy(local_rank)            # (a handle with arrays of its own)
        for kv in args.opt:
            k, v = kv.split("=")
            d2.set_option(k, int(v))
        if args.torch_vectors:
            x2, y2 = x.clone(), torch.empty_like(y)
        else:
            x2, y2 = d2.vectors_torch()
            x2.copy_(x)
        d2.autotune(x2, y2, iters=30)
        keep.append((d2, x2, y2))
        single.append(lambda d2=d2, x2=x2, y2=y2: d2.spmv_torch(x2, out=y2))
    single_step = rotating(single)

    def step():
        if world == 1:
            single_step()
        elif exchange == "end":
            op.local_only(x)        # the rank's rows into its slice buffer; nothing moves
        elif exchange == "halo":
            op.spmv_halo(x, y)
        else:
            op.spmv(x, y)

    def timed_region(xm, ym):
        """EXACTLY K steps between a barrier + device sync on both sides; with --exchange end the x distribution and
        the y collection (forms xm / ym) are inside.  Returns the region's time in ms, MAX over ranks."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_wall0 = time.perf_counter()
        e0.record()
        if exchange == "end":       # "x broadcast once via RCCL": inside the timed region
            if xm == "scatter_windows":
                op.distribute_x(x, ncols, x_needs)
            else:
                op.broadcast_x(x)
        for _ in range(args.steps):
            step()
        if exchange == "end" and ym == "gather_root":
            op.gather_y_root(y)     # "per-GPU y slices gathered at the end": once, inside the timed region
        elif exchange in ("halo", "end"):
            op.gather_y(y)
        e1.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t_wall0) * 1e3
        ev_ms = e0.elapsed_time(e1)
        elapsed = torch.tensor([wall_ms if world > 1 else ev_ms], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        return float(elapsed.item())

    for _ in range(args.warmup):
        step()
    total_ms = timed_region(x_mode, y_mode)
    ms_per_step = total_ms / args.steps
    # N > 1, --exchange end: the SAME region with the other pair of collectives, so that both end-to-end totals are
    # headline fields: north_star's wording to the letter (ncclBroadcast of all of x, all-gather of the y slices) and
    # the windows form (each GPU receives only the columns its rows read; y slices gathered on rank 0)
    end_to_end = None
    if world > 1 and exchange == "end":
        end_to_end = {"windows_scatter_gather_root_ms": None, "broadcast_allgather_ms": None}
        key = "windows_scatter_gather_root_ms" if x_mode == "scatter_windows" else "broadcast_allgather_ms"
        end_to_end[key] = round(total_ms, 4)
        if x_mode == "scatter_windows":
            end_to_end["broadcast_allgather_ms"] = round(timed_region("broadcast", "allgather"), 4)

    # ---- for transparency: the same K steps with an all-gather of y after EVERY step
    allgather_ms = None
    if exchange == "halo" or (exchange == "end" and args.extras):
        for _ in range(min(args.warmup, 5)):
            op.spmv(x, y)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(args.steps):
            op.spmv(x, y)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        ag = torch.tensor([(time.perf_counter() - ta) * 1e3 / args.steps], dtype=torch.float64, device=device)
        dist.all_reduce(ag, op=dist.ReduceOp.MAX)
        allgather_ms = float(ag.item())

    # ---- steady-state cost of every exchange step by itself (HIP events on the current stream, which each
    # collective blocks until it is done; median over repetitions after warm-up; MAX over ranks), so that
    # the end-to-end time can be recomputed for any K: total = x_distribution + K * compute + y_collection
    comm_ms = None
    if world > 1:
        def ev_median(fn, reps=10):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                dist.barrier()
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record()
                fn()
                a1.record()
                torch.cuda.synchronize()
                ts.append(a0.elapsed_time(a1))
            ts.sort()
            t = torch.tensor([ts[len(ts) // 2]], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return round(float(t.item()), 4)

        comm_ms = {}
        if x_mode == "scatter_windows":
            comm_ms["x_distribution"] = ev_median(lambda: op.distribute_x(x, ncols, x_needs))
            comm_ms["x_broadcast_whole_vector"] = ev_median(lambda: op.broadcast_x(x))
        else:
            comm_ms["x_distribution"] = ev_median(lambda: op.broadcast_x(x))
        op.local_only(x)
        if y_mode == "gather_root":
            comm_ms["y_collection"] = ev_median(lambda: op.gather_y_root(y))
            comm_ms["y_allgather"] = ev_median(lambda: op.gather_y(y))
        else:
            comm_ms["y_collection"] = ev_median(lambda: op.gather_y(y))
        if nrows == ncols:
            if getattr(op, "halo_recv", None) is None:
                op.plan_halo(int(ci.min()), int(ci.max()) + 1)     # collective: every rank is here
            ybuf = torch.empty_like(x)
            comm_ms["halo_step_incl_kernel"] = ev_median(lambda: op.spmv_halo(x, ybuf))
            comm_ms["halo_bytes_received"] = int(op.halo_bytes)
            del ybuf
        comm_ms["note"] = ("steady state, per call, max over ranks; x_distribution and y_collection are what the "
                           "timed region contains once each")

    # ---- the dominant kernel alone (HIP events on the launch stream = torch's current stream)
    if world == 1:
        # the timed region above IS K launches of this kernel and nothing else:
        # quote the roofline on the very same launches
        kern_ms = ms_per_step
    else:
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        y_loc = op.y_local[: r1 - r0]
        for _ in range(3):
            dev.spmv_torch(x, out=y_loc)
        torch.cuda.synchronize()
        k0.record()
        for _ in range(args.steps):
            dev.spmv_torch(x, out=y_loc)
        k1.record()
        torch.cuda.synchronize()
        kern_ms = k0.elapsed_time(k1) / args.steps
    kmax = torch.tensor([kern_ms], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    kern_ms_max = float(kmax.item())

    local_nnz = int(rp[-1])
    local_bytes = synth.spmv_bytes(local_nnz, r1 - r0, r1 - r0, ncols, esz)
    whole_bytes = synth.spmv_bytes(nnz, nrows, nrows, ncols, esz)
    achieved = local_bytes / (kern_ms * 1e-3) / 1e9          # GB/s, this rank's launch, ALGORITHMIC bytes (32-bit indices)
    peak = synth.HBM_PEAK_BYTES_PER_S / 1e9
    gflops = synth.spmv_flops(nnz) / (ms_per_step * 1e-3) / 1e9
    # what the planned kernel actually has to move: values, columns at the plan's index
    # width, the row pointers of the super-tiles that read them, the rank's window of x once, y once
    plan = dev.describe()
    idx_bytes = plan.get("index_bits", 32) // 8
    x_read = min(ncols, (r1 - r0) + (window if args.dist == "banded" else ncols))
    moved_bytes = (local_nnz * (esz + idx_bytes) + int(4 * (r1 - r0 + 1) * (1.0 - plan.get("uniform_row_fraction", 0.0)))
                   + esz * x_read + esz * (r1 - r0))
    moved = moved_bytes / (kern_ms * 1e-3) / 1e9

    # ---- a spot check so a wrong kernel / exchange cannot post a number: every rank evaluates the
    # first and last two rows of ITS shard in numpy against its slice; rank 0 also regenerates the
    # first and last row of EVERY shard and looks them up in the gathered y
    xh = synth.vector(ncols, dtype=np_dt)
    tol = 1e-10 if esz == 8 else 1e-4
    bad = 0
    nloc = r1 - r0
    y_own = op.y_local[:nloc] if world > 1 else y
    for r in sorted({0, 1, max(nloc - 2, 0), nloc - 1}):
        lo, hi = int(rp[r]), int(rp[r + 1])
        ref = float(np.dot(va[lo:hi].astype(np.float64), xh[ci[lo:hi].astype(np.int64)].astype(np.float64)))
        got = float(y_own[r].item())
        if abs(got - ref) > tol * max(1.0, abs(ref)):
            print(f"[rank {rank}] spot check failed at row {r0 + r}: {got} vs {ref}", file=sys.stderr)
            bad = 1
    if rank == 0 and world > 1:
        for g in range(world):
            for r in (int(bounds[g]), int(bounds[g + 1]) - 1):
                _, c1, v1 = gen_rows(r, r + 1)
                ref = float(np.dot(v1.astype(np.float64), xh[c1.astype(np.int64)].astype(np.float64)))
                got = float(y[r].item())
                if abs(got - ref) > tol * max(1.0, abs(ref)):
                    print(f"[rank 0] gathered y wrong at row {r} (shard {g}): {got} vs {ref}", file=sys.stderr)
                    bad = 1
    flag = torch.tensor([bad], dtype=torch.int32, device=device)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        sys.exit("spot check failed (see stderr)")

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            key = f"config{args.config}_{args.dist}_{args.dtype}_n{world}"
            if plan.get("persistent"):      # the form the autotune kept has its own counter run
                key += "_persistent"
            traffic = t.get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "CSR SpMV GFLOP/s (f64, 10Mx10M, 140M nnz)" if args.config == 3 and esz == 8 and args.dist == "banded"
                  else f"CSR SpMV GFLOP/s ({args.dtype}, config {args.config}, {args.dist})",
        "value": round(gflops, 3),
        "unit": "GFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 6),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"CsrMatrix {args.dtype} SpMV y=A*x, {nrows}x{ncols}, "
                        + (f"{per_row} nnz/row" if args.dist != "ragged" else "1 + next() % 27 nnz/row (SURVEY 8d robustness row)")
                        + f" ({nnz} nnz), {args.dist} columns"
                        + (f" W={window}" if args.dist != "uniform" else "")
                        + f" (BASELINE configs[{2 if args.config == 3 else 1}]), "
                        + ((f"single GPU" + (f"; launches rotate over {copies} independent copies of (A, x, y), "
                                             f"together larger than the 256 MB Infinity Cache" if copies > 1 else ""))
                           if world == 1 else
                           f"rows partitioned over {world} GPUs, x distributed once from rank 0 (RCCL), "
                           + ("y all-gather after every step (RCCL)" if exchange == "allgather" else
                              ("x windows scattered from rank 0 and y slices gathered on rank 0 (each GPU receives "
                               "only the columns its rows read)" if x_mode == "scatter_windows" else
                               "y slices all-gathered once at the end") +
                              "; the x distribution, the K local SpMVs and the gather are all inside the timed "
                              "region" if exchange == "end" else
                              f"per step a halo exchange (RCCL send/recv, {getattr(op, 'halo_bytes', 0)} B received "
                              f"per rank), y slices all-gathered once at the end of the timed region")),
            "nrows": nrows, "ncols": ncols, "nnz": nnz, "algorithmic_index_bits": 32,
            "partition": "none" if world == 1 else f"rows/{world}",
            "partition_rows": None if world == 1 else [int(b) for b in bounds],   # contiguous ranges balanced by stored entries
            "exchange": exchange,
            "x_distribution": x_mode if world > 1 else "none",
            "y_collection": y_mode if world > 1 else "none",
            "plan": plan,
        },
        # achieved_hbm_pct: the bytes the kernel really moves (16-bit columns ...) over the time, against 8 TB/s;
        # algorithmic_hbm_pct: the SURVEY 8d convention (32-bit indices, every pointer read) over the same time
        "achieved_hbm_pct": round(100.0 * moved / peak, 2) if world == 1 else None,
        "algorithmic_hbm_pct": round(100.0 * whole_bytes / (ms_per_step * 1e-3) / (world * synth.HBM_PEAK_BYTES_PER_S), 2),
        "algorithmic_bytes_per_step": whole_bytes,
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": peak,
            "unit": "GB/s",
            "frac": round(achieved / peak, 4),
            "traffic": traffic,
            "traffic_source": "profiles/traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)",
            "kernel": ("csr_spmv_slide" if plan.get("slide") and plan["kernel"] == "stream" else
                       "csr_spmv_cblock_rows" if plan["kernel"] == "cblock" and plan.get("cblock_form") == "rows" else "csr_spmv_" + plan["kernel"]),
            "kernel_ms": round(kern_ms, 6),
            "kernel_ms_max_over_ranks": round(kern_ms_max, 6),
            "algorithmic_bytes_per_launch": local_bytes,
            "note": "achieved / frac use the ALGORITHMIC bytes (32-bit indices, SURVEY 8d); moved_* are the bytes the "
                    "planned kernel has to move and the rate the HBM actually sustained",
            "moved_bytes_per_launch": moved_bytes,
            "moved_gbps": round(moved, 2),
            "moved_frac": round(moved / peak, 4),
        },
        "compute_only": {
            "ms_per_step": round(kern_ms_max, 6),
            "value": round(synth.spmv_flops(nnz) / (kern_ms_max * 1e-3) / 1e9, 3),
            "unit": "GFLOP/s",
        },
        # N > 1: K steps + one x distribution + one y collection, barrier to barrier, max over ranks -- in both forms
        # (`value` / `ms_per_step` are the region with `config.x_distribution` / `config.y_collection`)
        "end_to_end_windows_ms": None if end_to_end is None else end_to_end["windows_scatter_gather_root_ms"],
        "end_to_end_broadcast_allgather_ms": None if end_to_end is None else end_to_end["broadcast_allgather_ms"],
        "end_to_end_broadcast_allgather_value": None if end_to_end is None or not end_to_end["broadcast_allgather_ms"] else
            round(synth.spmv_flops(nnz) / (end_to_end["broadcast_allgather_ms"] / args.steps * 1e-3) / 1e9, 3),
        "comm_ms": comm_ms,
        "x_first_call_ms": round(x_bcast_ms, 4),   # (the first collective of the process: includes communicator warm-up)
        "allgather_every_step": None if allgather_ms is None else {
            "ms_per_step": round(allgather_ms, 6),
            "value": round(synth.spmv_flops(nnz) / (allgather_ms * 1e-3) / 1e9, 3), "unit": "GFLOP/s"},
        "setup_s": {"generate": round(t_gen, 2), "validate_narrow_upload_plan": round(t_upload, 2), "autotune": round(t_autotune, 3),
                    "place_vectors": round(t_vectors, 3)},
    }

    add_hbm_frac(out["roofline"], peak)
    if world == 1:
        # the ceiling next to which `frac` is read: what this box's HBM delivers for the same footprint
        ceil = None if args.no_ceiling else copy_ceiling(nrows, max(1, nnz // nrows))
        if ceil:
            out["roofline"]["copy_ceiling_gbs"] = ceil["footprint_gbs"]
            out["roofline"]["moved_frac_of_ceiling"] = round(moved / ceil["footprint_gbs"], 4)
            out["roofline"]["ceiling"] = dict(ceil, note="child process tools/micro/stream_ceiling --quick, same run, same GPU: the stream "
                                                         "kernel's loads and stores without gather or arithmetic; copy16 / read16 = plain "
                                                         "16-byte copy (read + write bytes) / read of the values array")
    if world == 1 and not args.no_cpu_baseline:
        import oracle  # CPU baseline leg only: the oracle is the thing timed here, never the product
        rp32, ci32 = rp.astype(np.uint32), ci.astype(np.uint32)
        yh = np.empty(nrows, dtype=np_dt)
        oracle.csr_spmv_idx32(rp32, ci32, va, xh, yh)  # warm the pages
        passes, t0 = 0, time.perf_counter()
        while True:
            oracle.csr_spmv_idx32(rp32, ci32, va, xh, yh)
            passes += 1
            el = time.perf_counter() - t0
            if el >= args.cpu_seconds or passes >= 1000:
                break
        out["cpu_baseline"] = {
            "value": round(synth.spmv_flops(nnz) * passes / el / 1e9, 4),
            "unit": "GFLOP/s",
            "cores": 1,
            "kind": "port",
            "sample": f"{passes} full passes over the same {nrows}x{ncols} / {nnz}-nnz matrix in {el:.1f} s, "
                      f"1 thread (the reference is single-threaded), 32-bit indices, gcc -O2 -ffp-contract=off; "
                      f"host has {os.cpu_count()} logical cores",
        }
        # full-size parity against the oracle: the stream path sums every row in the reference's order
        # (src/csr/ops/mul.rs:31-38), so its y must equal the CPU's bit for bit; rows computed by the vector /
        # overflow kernels (tree sums) are held to the componentwise bound of SURVEY 8d
        y_gpu = y.cpu().numpy()
        bits = np.uint64 if esz == 8 else np.uint32
        bit_identical = bool(np.array_equal(y_gpu.view(bits), yh.view(bits)))
        exact_expected = plan["kernel"] == "stream" and plan.get("stream_row_fraction", 0.0) == 1.0 and not plan.get("overflow_tiles")
        if bit_identical:
            agrees = True
        else:
            bound = oracle.csr_abs_bound(rp, ci, va, xh)
            tol = 1e-10 if esz == 8 else 1e-4
            err = np.abs(y_gpu.astype(np.float64) - yh.astype(np.float64))
            agrees = bool(not exact_expected and np.all(err <= tol * bound + 1e-300)
                          and err.max() <= tol * np.abs(yh).max())
        out["cpu_baseline"]["gpu_agrees_with_cpu"] = agrees
        out["cpu_baseline"]["gpu_equals_cpu_bit_for_bit"] = bit_identical
        out["cpu_baseline"]["criterion"] = ("np.array_equal on the bit patterns of all rows" if bit_identical or exact_expected
                                            else "componentwise |y - y_cpu| <= tol * sum|a||x| and normwise (SURVEY 8d)")
        if not agrees:
            print(json.dumps(out))
            sys.exit("the GPU result differs from the CPU oracle")
        # for information: what the crate's only existing route to A*x costs -- `&a * &x_as_matrix`,
        # i.e. transpose, transpose, Gustavson, transpose (src/csr/ops/mul.rs:8-59, restated in the oracle) -- on
        # the first 1M rows of the same matrix (the whole of it for config 2), x as an ncols x 1 CsrMatrix
        try:
            ns = min(nrows, 1_000_000)
            e1 = int(rp[ns])
            xm = (np.arange(ncols + 1, dtype=np.uint64), np.zeros(ncols, dtype=np.uint64), xh)
            t0 = time.perf_counter()
            pr, ir, vr = oracle.csr_mul((ns, ncols), (rp[:ns + 1], ci[:e1], va[:e1]), (ncols, 1), xm)
            el = time.perf_counter() - t0
            y_route = np.zeros(ns, dtype=np.float64 if esz == 8 else np.float32)
            y_route[np.repeat(np.arange(ns), np.diff(pr.astype(np.int64))) if pr[-1] != ns else slice(None)] = vr
            out["cpu_baseline_reference_route"] = {
                "value": round(2.0 * e1 / el / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": "port",
                "sample": f"one product of the first {ns} rows ({e1} entries) with x as a {ncols}x1 CsrMatrix in {el:.2f} s: "
                          f"the reference's literal `&a * &x` (3 counting-sort transposes + Gustavson)",
                "same_result_as_direct_loop": bool(np.array_equal(y_route.view(bits), yh[:ns].view(bits)))}
        except Exception as exc:  # noqa: BLE001  (informational only)
            out["cpu_baseline_reference_route"] = {"error": str(exc)}
        # for information: the same loop row-parallel on the host cores this box gives us
        try:
            from concurrent.futures import ThreadPoolExecutor
            threads = max(1, min(16, len(os.sched_getaffinity(0))))
            y_mt = np.empty(nrows, dtype=np_dt)
            with ThreadPoolExecutor(max_workers=threads) as pool:
                oracle.csr_spmv_idx32_threads(rp32, ci32, va, xh, y_mt, threads, pool)   # warm
                passes, t0 = 0, time.perf_counter()
                while True:
                    oracle.csr_spmv_idx32_threads(rp32, ci32, va, xh, y_mt, threads, pool)
                    passes += 1
                    el = time.perf_counter() - t0
                    if el >= min(args.cpu_seconds, 4.0) or passes >= 50:
                        break
            out["cpu_baseline_threads"] = {
                "value": round(synth.spmv_flops(nnz) * passes / el / 1e9, 4), "unit": "GFLOP/s",
                "cores": threads, "host_threads_available": len(os.sched_getaffinity(0)), "kind": "port",
                "sample": f"{passes} full passes in {el:.1f} s, {threads} threads, one row range each "
                          f"(informational: the reference has no threads)",
                "same_result_as_1_thread": bool(np.array_equal(y_mt, yh))}
        except Exception as exc:  # noqa: BLE001  (informational only)
            out["cpu_baseline_threads"] = {"error": str(exc)}
    if (world == 1 and args.config == 3 and args.dist == "banded" and args.dtype == "f64" and not args.opt
            and not args.no_other_configs):
        # the driver runs only this default line: every other BASELINE config rides along as a compact record
        out["other_configs"] = other_configs(args)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
