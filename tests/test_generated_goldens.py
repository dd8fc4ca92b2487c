"""Committed generated cases (tests/golden/generated_cases.json): the oracle
must keep reproducing them (CPU), and the GPU paths that are bit-identical by
design must hit the same checksums."""
import hashlib
import json
import os

import numpy as np
import pytest

import spalinalg_amd as sp
import spal_synth as synth

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "generated_cases.json")))["cases"]


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def coo_case(c):
    return synth.coo(c["nrows"], c["ncols"], c["length"], c["seed"], c["dup_permille"], c["cancel_permille"])


def test_oracle_reproduces_generated_goldens(oracle):
    for name, c in CASES.items():
        if c["gen"] == "coo":
            r, cc, v = coo_case(c)
            p, i, w = oracle.coo_to_csr(c["nrows"], c["ncols"], r, cc, v)
            assert sha(p, i, w) == c["csr_sha256"] and w.size == c["nnz"], name
            if "csc_sha256" in c:
                assert sha(*oracle.coo_to_csc(c["nrows"], c["ncols"], r, cc, v)) == c["csc_sha256"], name
            if "y_sha256" in c:
                assert sha(oracle.csr_spmv(p, i, w, synth.vector(c["ncols"], c["x_seed"]))) == c["y_sha256"], name
        else:
            dt = np.float64 if c["dtype"] == "f64" else np.float32
            rp, ci, va = synth.banded_csr(c["nrows"], c["ncols"], c["per_row"], c["window"], c["seed"], dtype=dt)
            assert sha(rp, ci, va) == c["input_sha256"], name        # the generator is pinned too
            x = synth.vector(c["ncols"], c["x_seed"], dtype=dt)
            assert sha(oracle.csr_spmv(rp, ci, va, x)) == c["y_sha256"], name
            assert sha(*oracle.transpose(c["nrows"], c["ncols"], rp, ci, va)) == c["csc_sha256"], name


@pytest.mark.gpu
def test_gpu_hits_generated_goldens():
    """no oracle call here: the device results are compared with committed checksums"""
    for name, c in CASES.items():
        if c["gen"] == "coo":
            r, cc, v = coo_case(c)
            coo = sp.CooMatrix.with_triplets(c["nrows"], c["ncols"], r, cc, v)
            csr = sp.CsrMatrix.from_coo(coo)
            assert sha(csr.rowptr(), csr.colind(), csr.values()) == c["csr_sha256"], name
            if "csc_sha256" in c:
                csc = sp.CscMatrix.from_coo(coo)
                assert sha(csc.colptr(), csc.rowind(), csc.values()) == c["csc_sha256"], name
            if "y_sha256" in c:
                dev = csr.device()
                dev.set_option("kernel", 2)       # lane-per-row sums: the reference's order
                y = dev.spmv(synth.vector(c["ncols"], c["x_seed"]))
                if dev.describe()["stream_row_fraction"] == 1.0:
                    assert sha(y) == c["y_sha256"], name
        else:
            dt = np.float64 if c["dtype"] == "f64" else np.float32
            rp, ci, va = synth.banded_csr(c["nrows"], c["ncols"], c["per_row"], c["window"], c["seed"], dtype=dt)
            a = sp.CsrMatrix(c["nrows"], c["ncols"], rp, ci, va)
            x = synth.vector(c["ncols"], c["x_seed"], dtype=dt)
            assert a.device().describe()["kernel"] == "stream"
            assert sha(a * x) == c["y_sha256"], name
            csc = sp.CscMatrix.from_csr(a)
            assert sha(csc.colptr(), csc.rowind(), csc.values()) == c["csc_sha256"], name
            assert sha(csc * x) == c["y_sha256"], name          # transposed route: same bits
