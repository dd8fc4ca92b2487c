#!/usr/bin/env python3
"""Placement classes against the walking order (development tool): several handles of config 3 in one process; for each,
the sliding kernel fully persistent (512 fronts), with runs dealt round-robin (compact fronts per XCD), with half the
grid, and the one-super-tile-per-workgroup kernel.  Does the slower placement class hurt the scattered fronts more?"""
import os, sys, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
