#!/usr/bin/env python3
"""Writes tests/golden/generated_cases.json: seeded synthetic inputs (regenerated
from the seeds by the library's generators, so no large arrays are committed)
and SHA-256 checksums of the CPU oracle's outputs for them.

The oracle itself is pinned by the reference's known-answer tests
(reference_kats.json); these cases extend the pin to sizes the reference's
tests do not reach, and let the GPU paths that are bit-identical by design
(CSR stream kernel, COO->CSR/CSC assembly, CSR<->CSC) be checked against a
committed value rather than only against a same-run oracle call.

    python tests/golden/make_generated.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import spalinalg_amd as sp  # noqa: E402  (host generators only: no GPU needed)
import spal_synth as synth  # noqa: E402


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    cases = {}
    # BASELINE config 1: 10k x 10k, 100k random triplets -> CSR -> SpMV
    cfg = synth.CONFIGS[1]
    r, c, v = synth.coo(cfg["nrows"], cfg["ncols"], cfg["length"], synth.matrix_seed(1))
    p, i, w = oracle.coo_to_csr(cfg["nrows"], cfg["ncols"], r, c, v)
    x = synth.vector(cfg["ncols"])
    y = oracle.csr_spmv(p, i, w, x)
    pc, ic, wc = oracle.coo_to_csc(cfg["nrows"], cfg["ncols"], r, c, v)
    cases["config1_coo_10k"] = dict(
        gen="coo", nrows=cfg["nrows"], ncols=cfg["ncols"], length=cfg["length"], seed=synth.matrix_seed(1),
        dup_permille=0, cancel_permille=0, nnz=int(w.size),
        csr_sha256=sha(p, i, w), csc_sha256=sha(pc, ic, wc), x_seed=synth.SEED_X, y_sha256=sha(y))
    # config-5 style injection at 200k entries
    r, c, v = synth.coo(20_000, 20_000, 200_000, synth.matrix_seed(5), 10, 1)
    p, i, w = oracle.coo_to_csr(20_000, 20_000, r, c, v)
    cases["config5_style_200k"] = dict(gen="coo", nrows=20_000, ncols=20_000, length=200_000,
                                       seed=synth.matrix_seed(5), dup_permille=10, cancel_permille=1,
                                       nnz=int(w.size), csr_sha256=sha(p, i, w))
    # banded CSR (config 2/3 generator) at 50k rows, f64 and f32
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        n = 50_000
        rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(3), dtype=dt)
        x = synth.vector(n, dtype=dt)
        y = oracle.csr_spmv(rp, ci, va, x)
        cp, ri, cv = oracle.transpose(n, n, rp, ci, va)
        cases[f"banded_50k_{name}"] = dict(gen="banded", nrows=n, ncols=n, per_row=14, window=4096,
                                           seed=synth.matrix_seed(3), dtype=name, input_sha256=sha(rp, ci, va),
                                           x_seed=synth.SEED_X, y_sha256=sha(y), csc_sha256=sha(cp, ri, cv))
    out = os.path.join(ROOT, "tests", "golden", "generated_cases.json")
    json.dump({"_comment": "made by tests/golden/make_generated.py from the CPU oracle (itself pinned by "
                           "reference_kats.json); inputs are regenerated from the seeds",
               "cases": cases}, open(out, "w"), indent=1)
    print("wrote", out, list(cases))


if __name__ == "__main__":
    main()
