"""GPU parity: CSC scatter SpMV and device COO -> CSR assembly vs the oracle."""
import os

import numpy as np
import pytest

import spalinalg_amd as sp
import spal_synth as synth
from tests.util import assert_spmv_close, random_csr

pytestmark = pytest.mark.gpu


# ---- CSR <-> CSC on the device ---------------------------------
def test_conversion_kats_g3_g4_g6(kats):
    g = kats["G3_csc_to_csr"]
    csc = sp.CscMatrix(g["nrows"], g["ncols"], g["colptr"], g["rowind"], np.array(g["csc_values"]))
    csr = sp.CsrMatrix.from_csc(csc)
    assert (csr.rowptr().tolist(), csr.colind().tolist(), csr.values().tolist()) == (
        g["rowptr"], g["colind"], g["csr_values"])
    g = kats["G4_csr_to_csc"]
    csr = sp.CsrMatrix(g["nrows"], g["ncols"], g["rowptr"], g["colind"], np.array(g["csr_values"]))
    csc = sp.CscMatrix.from_csr(csr)
    assert (csc.colptr().tolist(), csc.rowind().tolist(), csc.values().tolist()) == (
        g["colptr"], g["rowind"], g["csc_values"])
    # transpose doc-tests: the CSC arrays of M are the CSR arrays of M^T
    g = kats["G6_transpose"]["csr"]
    m = sp.CsrMatrix(g["n"], g["n"], g["rowptr"], g["colind"], np.array(g["values"]))
    t = sp.CscMatrix.from_csr(m)
    assert (t.colptr().tolist(), t.rowind().tolist(), t.values().tolist()) == (
        g["t_rowptr"], g["t_colind"], g["t_values"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_conversions_random_exact(oracle, dtype):
    """entries are only moved: indices AND values equal the oracle's bit for bit"""
    rng = np.random.default_rng(31)
    for nr, nc in [(1, 1), (1, 300), (300, 1), (257, 300), (40_000, 30_000), (3000, 200_000)]:
        rp, ci, va = random_csr(rng, nr, nc, density=min(0.3, 6.0 / nc), dtype=dtype)
        cp, ri, cv = oracle.transpose(nr, nc, rp, ci, va)
        csr = sp.CsrMatrix(nr, nc, rp, ci, va)
        csc = sp.CscMatrix.from_csr(csr)
        assert np.array_equal(csc.colptr(), cp) and np.array_equal(csc.rowind(), ri)
        assert np.array_equal(csc.values(), cv)
        back = sp.CsrMatrix.from_csc(csc)
        assert np.array_equal(back.rowptr(), rp) and np.array_equal(back.colind(), ci)
        assert np.array_equal(back.values(), va)
        # the converted matrices are valid in the reference's sense
        sp.CscMatrix(nr, nc, csc.colptr(), csc.rowind(), csc.values())
    # one row holding 100k entries (longer than any LDS tile)
    nr, nc = 3, 200_000
    rp = np.array([0, 0, 100_000, 100_000], dtype=np.uint64)
    ci = np.sort(rng.choice(nc, 100_000, replace=False)).astype(np.uint64)
    va = rng.uniform(-1, 1, 100_000).astype(dtype)
    cp, ri, cv = oracle.transpose(nr, nc, rp, ci, va)
    csc = sp.CscMatrix.from_csr(sp.CsrMatrix(nr, nc, rp, ci, va))
    assert np.array_equal(csc.colptr(), cp) and np.array_equal(csc.rowind(), ri) and np.array_equal(csc.values(), cv)


def test_csc_transposed_kernel_is_bit_identical(oracle):
    """kernel 2 (the default): CSC -> CSR once on the device, then the stream
    kernel: y equals the reference's k-ascending sums bit for bit."""
    n = 200_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 3)
    cp, ri, cv = oracle.transpose(n, n, rp, ci, va)
    x = synth.vector(n)
    dev = sp.CscMatrix(n, n, cp, ri, cv).device()
    assert dev.describe()["kernel"] == "transposed_csr"
    y = dev.spmv(x)
    assert np.array_equal(y, oracle.csc_spmv(n, cp, ri, cv, x))
    dev.set_option("kernel", 1)
    d = dev.describe()
    assert d["kernel"] == "lds_privatised_scatter" and d["row_tiles"] == 1 and d["row_tile_rows"] == 4096, d
    np.testing.assert_allclose(dev.spmv(x), y, rtol=1e-10, atol=1e-13)
    dev.set_option("row_tiles", 0)
    assert dev.describe()["row_tiles"] == 0
    np.testing.assert_allclose(dev.spmv(x), y, rtol=1e-10, atol=1e-13)


# ---- CSC ---------------------------------------------------------------------
def test_csc_kat_g5(kats):
    g = kats["G5_csc_mul"]
    a = g["lhs"]
    m = sp.CscMatrix(a["nrows"], a["ncols"], a["colptr"], a["rowind"], np.array(a["values"]))
    for case in g["spmv"]:
        assert (m * np.array(case["x"])).tolist() == case["y"]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_csc_random(oracle, dtype):
    rng = np.random.default_rng(17)
    for nr, nc in [(1, 1), (40, 30), (3000, 2000), (2000, 3000)]:
        rp, ci, va = random_csr(rng, nr, nc, density=min(0.3, 10.0 / nc), dtype=dtype)
        cp, ri, cv = oracle.transpose(nr, nc, rp, ci, va)
        x = rng.uniform(-1, 1, nc).astype(dtype)
        m = sp.CscMatrix(nr, nc, cp, ri, cv)
        y_ref = oracle.csc_spmv(nr, cp, ri, cv, x)
        bound = oracle.csr_abs_bound(rp, ci, va.astype(np.float64), x.astype(np.float64))
        for kernel in (2, 1):
            m.device().set_option("kernel", kernel)
            assert_spmv_close(m * x, y_ref, bound, 1e-10 if dtype == np.float64 else 1e-4)


def test_csc_lds_and_global_paths(oracle):
    """banded columns use the LDS-privatised scatter; a few full-height columns
    force single super-tiles onto the global-atomic path; both must agree with
    the oracle, as must the forced global path."""
    rng = np.random.default_rng(8)
    n = 30_000
    rp, ci, va = synth.banded_csr(n, n, 14, 2048, 21)
    cp, ri, cv = oracle.transpose(n, n, rp, ci, va)        # CSC of a banded matrix
    cols = [ri[int(cp[k]):int(cp[k + 1])] for k in range(n)]
    vals = [cv[int(cp[k]):int(cp[k + 1])] for k in range(n)]
    for k in (5, 12_000, 29_999):                           # tall columns: window too large for LDS
        cols[k] = np.sort(rng.choice(n, 3000, replace=False)).astype(np.uint64)
        vals[k] = rng.uniform(-1, 1, 3000)
    cp = np.concatenate([[0], np.cumsum([c.size for c in cols])]).astype(np.uint64)
    ri, cv = np.concatenate(cols), np.concatenate(vals)
    x = synth.vector(n)
    x[12_000] = np.inf
    m = sp.CscMatrix(n, n, cp, ri, cv)
    rp2, ci2, va2 = oracle.transpose(n, n, cp, ri, cv)      # rows again, for the error bound
    y_ref = oracle.csc_spmv(n, cp, ri, cv, x)
    bound = oracle.csr_abs_bound(rp2, ci2, va2, np.nan_to_num(x, posinf=0.0))
    dev = m.device()
    assert_spmv_close(dev.spmv(x), y_ref, bound, 1e-10)        # default: transposed
    dev.set_option("kernel", 1)
    dev.set_option("row_tiles", 0)                           # (this test is about the COLUMN tiles)
    d = dev.describe()
    assert d["kernel"] == "lds_privatised_scatter" and 0.8 < d["lds_col_fraction"] < 1.0
    assert d["flush"] == "global_atomics"                    # default: window rows flushed with atomics
    assert_spmv_close(dev.spmv(x), y_ref, bound, 1e-10)
    dev.set_option("flush", 1)                               # windows stored, then an ordered reduce
    assert dev.describe()["flush"] == "windows_then_reduce"
    assert_spmv_close(dev.spmv(x), y_ref, bound, 1e-10)
    dev.set_option("flush", 0)
    dev.set_option("lds", 0)
    assert dev.describe()["kernel"] == "atomic_scatter"
    assert_spmv_close(dev.spmv(x), y_ref, bound, 1e-10)
    dev.set_option("lds", 1)
    for lanes in (2, 64):
        dev.set_option("lanes_per_col", lanes)
        assert_spmv_close(dev.spmv(x), y_ref, bound, 1e-10)
    # every width of the scatter kernel's column super-tiles (wider ones leave more of them to the global path here)
    for cols in (1024, 2048, 4096, 0):
        dev.set_option("cols_per_block", cols)
        d = dev.describe()
        assert cols == 0 or d["cols_per_block"] == cols
        assert_spmv_close(dev.spmv(x), y_ref, bound, 1e-10)
        dev.set_option("flush", 1)
        assert_spmv_close(dev.spmv(x), y_ref, bound, 1e-10)
        dev.set_option("flush", 0)
    with pytest.raises(sp.Panic):
        dev.set_option("cols_per_block", 512)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_csc_neighbour_handoff_flush(oracle, dtype):
    """Banded columns: row windows ascend and only neighbours overlap, so the scatter kernel stores every row of y once
    and hands the shared rows to the next super-tile behind a flag -- no memset, no global atomics.  Against the oracle
    and against the two other flush forms; y pre-filled with NaN (every row must be written, also rows that no column
    touches); launches back to back and on several streams (they chain on the handle's event); under graph capture the
    atomics form runs."""
    import torch
    tol = 1e-10 if dtype == np.float64 else 1e-4
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    n = 300_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 33, dtype=dtype)
    # empty rows at the head, in the middle (wider than a window overlap) and at the tail; a stretch of empty columns
    keep = np.ones(n, bool)
    keep[:700] = False
    keep[150_000:153_000] = False
    keep[-1200:] = False
    lens = np.diff(rp.astype(np.int64)) * keep
    sel = np.repeat(keep, np.diff(rp.astype(np.int64)))
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ci, va = ci[sel], va[sel]
    colsel = (ci < 40_000) | (ci >= 49_000)            # columns 40 000 ... 48 999 hold nothing: super-tiles without entries
    lens = np.bincount(np.repeat(np.arange(n), np.diff(rp.astype(np.int64)))[colsel], minlength=n)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ci, va = ci[colsel], va[colsel]
    cp, ri, cv = oracle.transpose(n, n, rp, ci, va)
    x = synth.vector(n, dtype=dtype)
    y_ref = oracle.csc_spmv(n, cp, ri, cv, x)
    bound = oracle.csr_abs_bound(rp, ci, va, x)
    dev = sp.CscMatrix(n, n, cp, ri, cv).device()
    dev.set_option("kernel", 1)
    dev.set_option("row_tiles", 0)                           # (this test is about the COLUMN tiles)
    xt = torch.from_numpy(x).cuda()
    for cols in (0, 1024, 2048, 4096):
        dev.set_option("cols_per_block", cols)
        d = dev.describe()
        # (windows of 1024 / 2048 columns of this band reach beyond their neighbours: those widths keep the atomics)
        handoff = cols in (0, 4096)
        assert d["flush"] == ("neighbour_handoff" if handoff else "global_atomics") and d["lds_col_fraction"] > 0.95, d
        yt = torch.full((n,), float("nan"), dtype=tdt, device="cuda")
        for _ in range(3):                                   # back to back: launch numbers 1, 2, 3 in the flags
            dev.spmv_torch(xt, yt)
        torch.cuda.synchronize()
        assert_spmv_close(yt.cpu().numpy(), y_ref, bound, tol)
        assert np.all(yt.cpu().numpy()[:700] == 0) and np.all(yt.cpu().numpy()[-1200:] == 0)
        for flush, name in ((2, "global_atomics"), (1, "windows_then_reduce"),
                            (0, "neighbour_handoff" if handoff else "global_atomics")):
            dev.set_option("flush", flush)
            assert dev.describe()["flush"] == name
            assert_spmv_close(dev.spmv(x), y_ref, bound, tol)
    dev.set_option("cols_per_block", 0)
    assert dev.describe()["flush"] == "neighbour_handoff"
    # several streams at once on one handle
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = [torch.full((n,), float("nan"), dtype=tdt, device="cuda") for _ in streams]
    torch.cuda.synchronize()
    for _ in range(5):
        for st, out in zip(streams, outs):
            with torch.cuda.stream(st):
                dev.spmv_torch(xt, out)
    torch.cuda.synchronize()
    for out in outs:
        assert_spmv_close(out.cpu().numpy(), y_ref, bound, tol)
    # captured into a graph: the hand-off's event chain cannot be captured, the atomics form runs (and replays)
    g = torch.cuda.CUDAGraph()
    yg = torch.zeros(n, dtype=tdt, device="cuda")
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        dev.spmv_torch(xt, yg)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            dev.spmv_torch(xt, yg)
    for _ in range(2):
        yg.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert_spmv_close(yg.cpu().numpy(), y_ref, bound, tol)
    dev.spmv_torch(xt, yg)            # and the hand-off again afterwards
    torch.cuda.synchronize()
    assert_spmv_close(yg.cpu().numpy(), y_ref, bound, tol)


def _band_csc(oracle, n, dtype=np.float64, seed=41):
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, seed, dtype=dtype)
    cp, ri, cv = oracle.transpose(n, n, rp, ci, va)
    x = synth.vector(n, dtype=dtype)
    return cp, ri, cv, x, oracle.csc_spmv(n, cp, ri, cv, x), oracle.csr_abs_bound(rp, ci, va, x)


@pytest.mark.parametrize("ticket", [1, 0])
def test_csc_handoff_tile_ids_by_ticket_or_block_index(oracle, ticket):
    """Neighbour hand-off with the super-tile taken from the start-order ticket or from blockIdx: 600 products
    back to back on one handle (the ticket counter and the flags run on from launch to launch), more super-tiles than
    the device holds at once (two rounds of workgroups), y pre-filled with NaN each time."""
    import torch
    n = 2_200_000                                   # 538 super-tiles of 4096 columns: more than 256 CUs x 1
    cp, ri, cv, x, y_ref, bound = _band_csc(oracle, n)
    dev = sp.CscMatrix(n, n, cp, ri, cv).device()
    dev.set_option("kernel", 1)
    dev.set_option("row_tiles", 0)                           # (this test is about the COLUMN tiles)
    dev.set_option("ticket", ticket)
    d = dev.describe()
    assert d["flush"] == "neighbour_handoff" and d["ticket"] == ticket and d["blocks"] > 512, d
    xt = torch.from_numpy(x).cuda()
    yt = torch.empty(n, dtype=torch.float64, device="cuda")
    ref_t, bound_t = torch.from_numpy(y_ref).cuda(), torch.from_numpy(bound).cuda()
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    for it in range(600):                            # every product is checked (on the device: no host round trip)
        yt.fill_(float("nan"))
        dev.spmv_torch(xt, yt)
        bad += (~((yt - ref_t).abs() <= 1e-10 * bound_t + 1e-300)).sum()      # (NaN counts as bad)
        if it in (0, 1, 599):
            assert_spmv_close(yt.cpu().numpy(), y_ref, bound, 1e-10)
    assert int(bad.item()) == 0
    assert dev.describe()["handoff_timeouts"] == 0


def test_csc_handoff_backstop_is_reported(oracle, monkeypatch):
    """The hand-off's spin bound forced to 0 (SPAL_CSC_HANDOFF_SPINS, read when the plan is built): a super-tile that finds
    its predecessor's flag missing gives up at once.  Host-vector path: the library sees the report after its own
    synchronisation and repeats the product with atomics -- the caller gets the right y.  Device path: the NEXT call on
    the handle fails loudly (that earlier y was invalid), the handle then flushes with atomics and is right again."""
    import torch
    n = 1_200_000
    cp, ri, cv, x, y_ref, bound = _band_csc(oracle, n, seed=43)
    monkeypatch.setenv("SPAL_CSC_HANDOFF_SPINS", "0")
    # host-vector path
    dev = sp.CscMatrix(n, n, cp, ri, cv).device()
    dev.set_option("kernel", 1)
    dev.set_option("row_tiles", 0)                           # (this test is about the COLUMN tiles)
    assert dev.describe()["flush"] == "neighbour_handoff"
    for _ in range(30):                              # (all super-tiles finish together: some find a flag missing soon)
        assert_spmv_close(dev.spmv(x), y_ref, bound, 1e-10)
        if dev.describe()["handoff_timeouts"]:
            break
    d = dev.describe()
    assert d["handoff_timeouts"] == 1 and d["flush"] == "global_atomics", d
    assert_spmv_close(dev.spmv(x), y_ref, bound, 1e-10)
    # device path
    dev = sp.CscMatrix(n, n, cp, ri, cv).device()
    dev.set_option("kernel", 1)
    dev.set_option("row_tiles", 0)                           # (this test is about the COLUMN tiles)
    xt = torch.from_numpy(x).cuda()
    yt = torch.empty(n, dtype=torch.float64, device="cuda")
    raised = False
    assert dev.invalid_products() == 0
    for _ in range(60):
        try:
            dev.spmv_torch(xt, yt)
        except sp.SpalError as e:
            raised = True
            assert "hand-off" in str(e), e
            break
        torch.cuda.synchronize()
        if dev.invalid_products():                    # spal_csc_status: visible right after the synchronisation,
            assert dev.invalid_products() == 1         # before the next product fails loudly
    assert raised, "no super-tile ever took the backstop"
    assert dev.invalid_products() == 1
    d = dev.describe()
    assert d["handoff_timeouts"] == 1 and d["flush"] == "global_atomics", d
    yt.fill_(float("nan"))
    dev.spmv_torch(xt, yt)
    torch.cuda.synchronize()
    assert_spmv_close(yt.cpu().numpy(), y_ref, bound, 1e-10)


def test_csc_handoff_not_taken_when_windows_interleave(oracle):
    """rows anywhere: every super-tile's window covers most rows -- atomics (or the transposed route), as before"""
    rng = np.random.default_rng(5)
    n = 20_000
    rp, ci, va = random_csr(rng, n, n, density=6.0 / n)
    cp, ri, cv = oracle.transpose(n, n, rp, ci, va)
    x = synth.vector(n)
    dev = sp.CscMatrix(n, n, cp, ri, cv).device()
    dev.set_option("kernel", 1)
    dev.set_option("row_tiles", 0)                           # (this test is about the COLUMN tiles)
    assert dev.describe()["flush"] == "global_atomics"
    assert_spmv_close(dev.spmv(x), oracle.csc_spmv(n, cp, ri, cv, x), oracle.csr_abs_bound(rp, ci, va, x), 1e-10)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SPAL_FUZZ_SEEDS", "12"))))
def test_csc_scatter_randomised_bands(oracle, seed):
    """Random band-like CSC matrices (width, column fill, empty column stretches, sections whose rows jump back): whether
    the plan hands rows from super-tile to super-tile or falls back to atomics, y (pre-filled with NaN) must be the
    oracle's, for every super-tile width."""
    import torch
    rng = np.random.default_rng(4000 + seed)
    dtype = np.float64 if seed % 3 else np.float32
    nrows = int(rng.choice([6_000, 40_000, 150_000]))
    ncols = int(nrows * rng.choice([0.5, 1.0, 1.7]))
    width = int(rng.choice([64, 700, 3000, 9000]))
    per_col = rng.integers(0, int(rng.choice([3, 12, 30])) + 1, ncols)
    for _ in range(int(rng.integers(0, 4))):          # stretches of empty columns
        a = int(rng.integers(0, ncols))
        per_col[a:a + int(rng.integers(1, 6000))] = 0
    centre = (np.arange(ncols) * (nrows / ncols)).astype(np.int64)
    if seed % 4 == 3:                                 # a section whose rows jump back: windows no longer ascend
        a = ncols // 2
        centre[a:] = np.maximum(centre[a:] - nrows // 3, 0)
    cols = np.repeat(np.arange(ncols, dtype=np.int64), per_col)
    rows = np.clip(centre[cols] - width // 2 + rng.integers(0, width, cols.size), 0, nrows - 1)
    key = np.unique(cols * nrows + rows)              # sorted by (column, row), duplicates dropped
    cols, rows = key // nrows, key % nrows
    cp = np.concatenate([[0], np.cumsum(np.bincount(cols, minlength=ncols))]).astype(np.uint64)
    ri = rows.astype(np.uint64)
    cv = rng.uniform(-1, 1, ri.size).astype(dtype)
    x = rng.uniform(-1, 1, ncols).astype(dtype)
    y_ref = oracle.csc_spmv(nrows, cp, ri, cv, x)
    rp, ci, va = oracle.transpose(ncols, nrows, cp, ri, cv)     # rows of A, for the error bound
    bound = oracle.csr_abs_bound(rp, ci, va.astype(np.float64), x.astype(np.float64))
    tol = 1e-10 if dtype == np.float64 else 1e-4
    dev = sp.CscMatrix(nrows, ncols, cp, ri, cv).device()
    dev.set_option("kernel", 1)
    xt = torch.from_numpy(x).cuda()
    # the row tiles (the scatter path's default where every tile's window of x fits LDS) ...
    if dev.describe()["row_tiles"]:
        for _ in range(2):
            yt = torch.full((nrows,), float("nan"), dtype=xt.dtype, device="cuda")
            dev.spmv_torch(xt, yt)
        torch.cuda.synchronize()
        assert_spmv_close(yt.cpu().numpy(), y_ref, bound, tol)
    if width <= 3000 and seed % 4 != 3:
        assert dev.describe()["row_tiles"] == 1, dev.describe()
    # ... and the column tiles
    dev.set_option("row_tiles", 0)
    seen = set()
    for cols_opt in (0, 1024, 2048, 4096):
        dev.set_option("cols_per_block", cols_opt)
        seen.add(dev.describe()["flush"])
        for _ in range(2):
            yt = torch.full((nrows,), float("nan"), dtype=xt.dtype, device="cuda")
            dev.spmv_torch(xt, yt)
        torch.cuda.synchronize()
        assert_spmv_close(yt.cpu().numpy(), y_ref, bound, tol)
    assert seen <= {"neighbour_handoff", "global_atomics"}
    if seed % 4 != 3 and width <= 700 and ncols >= 20_000:
        assert "neighbour_handoff" in seen       # a narrow ascending band: some super-tile width hands rows over
    dev.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_csc_row_tiles(oracle, dtype):
    """The scatter path over ROW tiles (a workgroup owns rows of y: no hand-off, no memset, no global atomics): a band with
    empty rows at the head, in the middle and at the tail and a stretch of empty columns; y pre-filled with NaN; several
    streams at once and a captured graph (nothing is shared between launches); tall columns disqualify the tiling."""
    import torch
    tol = 1e-10 if dtype == np.float64 else 1e-4
    n = 300_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 34, dtype=dtype)
    keep = np.ones(n, bool)
    keep[:700] = False
    keep[150_000:159_000] = False                      # more than two whole row tiles without entries
    keep[-1200:] = False
    lens = np.diff(rp.astype(np.int64)) * keep
    sel = np.repeat(keep, np.diff(rp.astype(np.int64)))
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ci, va = ci[sel], va[sel]
    cp, ri, cv = oracle.transpose(n, n, rp, ci, va)
    x = synth.vector(n, dtype=dtype)
    y_ref = oracle.csc_spmv(n, cp, ri, cv, x)
    bound = oracle.csr_abs_bound(rp, ci, va, x)
    dev = sp.CscMatrix(n, n, cp, ri, cv).device()
    dev.set_option("kernel", 1)
    d = dev.describe()
    assert d["row_tiles"] == 1 and d["row_tile_rows"] == 4096 and d["row_tile_x_window"] <= 8194, d
    xt = torch.from_numpy(x).cuda()
    yt = torch.full((n,), float("nan"), dtype=xt.dtype, device="cuda")
    for _ in range(3):
        dev.spmv_torch(xt, yt)
    torch.cuda.synchronize()
    y = yt.cpu().numpy()
    assert_spmv_close(y, y_ref, bound, tol)
    assert np.all(y[:700] == 0) and np.all(y[150_000:159_000] == 0) and np.all(y[-1200:] == 0)
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = [torch.full((n,), float("nan"), dtype=xt.dtype, device="cuda") for _ in streams]
    torch.cuda.synchronize()
    for _ in range(5):
        for st, out in zip(streams, outs):
            with torch.cuda.stream(st):
                dev.spmv_torch(xt, out)
    torch.cuda.synchronize()
    for out in outs:
        assert_spmv_close(out.cpu().numpy(), y_ref, bound, tol)
    g = torch.cuda.CUDAGraph()
    yg = torch.full((n,), float("nan"), dtype=xt.dtype, device="cuda")
    with torch.cuda.graph(g):
        dev.spmv_torch(xt, yg)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert_spmv_close(yg.cpu().numpy(), y_ref, bound, tol)
    # smaller tiles where 4096 rows + their columns do not fit: a band of 14 000 columns (f64: 4096 + 18 000 > 20 352 elements)
    rp2, ci2, va2 = synth.banded_csr(60_000, 60_000, 10, 14_000, 35, dtype=dtype)
    cp2, ri2, cv2 = oracle.transpose(60_000, 60_000, rp2, ci2, va2)
    x2 = synth.vector(60_000, dtype=dtype)
    dev2 = sp.CscMatrix(60_000, 60_000, cp2, ri2, cv2).device()
    dev2.set_option("kernel", 1)
    d2 = dev2.describe()
    assert d2["row_tiles"] == 1 and d2["row_tile_rows"] == (2048 if dtype == np.float64 else 4096), d2
    assert_spmv_close(dev2.spmv(x2), oracle.csc_spmv(60_000, cp2, ri2, cv2, x2), oracle.csr_abs_bound(rp2, ci2, va2, x2), tol)
    for rows in (1024, 2048, 0):                      # the height as an option; 0 = the tallest that fits again
        dev2.set_option("row_tile_rows", rows)
        d2 = dev2.describe()
        assert d2["row_tiles"] == 1 and (rows == 0 or d2["row_tile_rows"] == rows), d2
        assert_spmv_close(dev2.spmv(x2), oracle.csc_spmv(60_000, cp2, ri2, cv2, x2), oracle.csr_abs_bound(rp2, ci2, va2, x2), tol)
    with pytest.raises(sp.Panic):
        dev2.set_option("row_tile_rows", 3000)
    # columns anywhere: no tiling of the rows keeps its columns inside LDS
    rp3, ci3, va3 = synth.banded_csr(100_000, 100_000, 8, 100_000, 36, dtype=dtype)
    cp3, ri3, cv3 = oracle.transpose(100_000, 100_000, rp3, ci3, va3)
    dev3 = sp.CscMatrix(100_000, 100_000, cp3, ri3, cv3).device()
    dev3.set_option("kernel", 1)
    assert dev3.describe()["row_tiles"] == 0
    x3 = synth.vector(100_000, dtype=dtype)
    assert_spmv_close(dev3.spmv(x3), oracle.csc_spmv(100_000, cp3, ri3, cv3, x3), oracle.csr_abs_bound(rp3, ci3, va3, x3), tol)


def test_csc_config4(oracle):
    """BASELINE config 4: CSC of the config-2 matrix, 1M x 1M."""
    n = 1_000_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(2))
    cp, ri, cv = oracle.transpose(n, n, rp, ci, va)
    x = synth.vector(n)
    m = sp.CscMatrix(n, n, cp, ri, cv)
    m.device().set_option("kernel", 1)          # the atomic scatter path config 4 names
    d = m.device().describe()
    assert d["row_tiles"] == 1 and d["row_tile_rows"] == 4096 and d["row_tile_count"] == 245, d   # rows of y owned by one workgroup each
    y = m * x
    y_ref = oracle.csc_spmv(n, cp, ri, cv, x)
    assert_spmv_close(y, y_ref, oracle.csr_abs_bound(rp, ci, va, x), 1e-10)
    m.device().set_option("row_tiles", 0)       # the column tiles with their neighbour hand-off
    assert m.device().describe()["flush"] == "neighbour_handoff"
    assert_spmv_close(m * x, y_ref, oracle.csr_abs_bound(rp, ci, va, x), 1e-10)
    assert m.device().describe()["lds_col_fraction"] > 0.99
    assert m.device().describe()["cols_per_block"] == 4096      # the band's windows fit beside 4096 columns of x
    m.device().set_option("flush", 1)           # windows + ordered reduce (writes y without a memset here)
    assert_spmv_close(m * x, y_ref, oracle.csr_abs_bound(rp, ci, va, x), 1e-10)
    m.device().set_option("flush", 0)
    m32 = sp.CscMatrix(n, n, cp, ri, cv.astype(np.float32))
    m32.device().set_option("kernel", 1)
    assert m32.device().describe()["row_tiles"] == 1
    y32 = m32 * x.astype(np.float32)
    assert_spmv_close(y32, y_ref, oracle.csr_abs_bound(rp, ci, va, x), 1e-4)
    m32.device().set_option("row_tiles", 0)
    assert_spmv_close(m32 * x.astype(np.float32), y_ref, oracle.csr_abs_bound(rp, ci, va, x), 1e-4)
    with pytest.raises(sp.Panic):
        m * np.ones(n - 1)


# ---- COO -> CSR ----------------------------------------------------------------
def assemble_and_compare(oracle, nrows, ncols, r, c, v):
    coo = sp.CooMatrix.with_triplets(nrows, ncols, r, c, v)
    csr = sp.CsrMatrix.from_coo(coo)
    p, i, w = oracle.coo_to_csr(nrows, ncols, r, c, v)
    assert np.array_equal(csr.rowptr(), p)
    assert np.array_equal(csr.colind(), i)
    # bit-exact: same order of additions as the reference
    assert np.array_equal(csr.values().view(np.uint64 if v.dtype == np.float64 else np.uint32),
                          w.view(np.uint64 if v.dtype == np.float64 else np.uint32))
    return csr


def test_coo_to_csc_kat_g2_and_random(kats, oracle):
    """`CscMatrix::from(&coo)` (src/csc/conv/coo.rs): G2 and random inputs, bit-exact."""
    g = kats["G2_coo_to_csc"]
    coo = sp.CooMatrix.with_triplets(g["nrows"], g["ncols"], g["rows"], g["cols"], np.array(g["vals"]))
    csc = sp.CscMatrix.from_coo(coo)
    assert (csc.colptr().tolist(), csc.rowind().tolist(), csc.values().tolist()) == (
        g["colptr"], g["rowind"], g["values"])
    rng = np.random.default_rng(41)
    for nr, nc, n in [(1, 1, 9), (90, 100, 5000), (3, 70_000, 100_000), (5000, 4000, 250_000)]:
        r = rng.integers(0, nr, n).astype(np.uint64)
        c = rng.integers(0, nc, n).astype(np.uint64)
        v = rng.integers(-2, 3, n).astype(np.float64) * rng.uniform(0.5, 1.5)
        csc = sp.CscMatrix.from_coo(sp.CooMatrix.with_triplets(nr, nc, r, c, v))
        p, i, w = oracle.coo_to_csc(nr, nc, r, c, v)
        assert np.array_equal(csc.colptr(), p) and np.array_equal(csc.rowind(), i)
        assert np.array_equal(csc.values().view(np.uint64), w.view(np.uint64))
        x = rng.uniform(-1, 1, nc)
        np.testing.assert_allclose(csc * x, oracle.csc_spmv(nr, p, i, w, x), rtol=1e-10, atol=1e-12)


def test_coo_kat_g1(kats, oracle):
    g = kats["G1_coo_to_csr"]
    coo = sp.CooMatrix(g["nrows"], g["ncols"])
    for r, c, v in zip(g["rows"], g["cols"], g["vals"]):
        coo.push(r, c, v)
    csr = sp.CsrMatrix.from_coo(coo)
    assert csr.rowptr().tolist() == g["rowptr"]
    assert csr.colind().tolist() == g["colind"]
    assert csr.values().tolist() == g["values"]
    # and the assembled matrix multiplies
    assert (csr * np.array([1.0, 1.0, 1.0])).tolist() == [10.0, 5.0]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_coo_random_with_duplicates(oracle, dtype):
    rng = np.random.default_rng(23)
    for nr, nc, n in [(1, 1, 1), (1, 1, 50), (5, 7, 400), (100, 90, 5000), (70_000, 3, 200_000),
                      (3, 70_000, 200_000), (5000, 5000, 300_000)]:
        r = rng.integers(0, nr, n).astype(np.uint64)
        c = rng.integers(0, nc, n).astype(np.uint64)
        v = rng.uniform(-1, 1, n).astype(dtype)
        v[rng.random(n) < 0.05] = 0.0
        assemble_and_compare(oracle, nr, nc, r, c, v)


def test_coo_order_of_duplicate_sums_and_zero_drop(oracle):
    big = 2.0 ** 53
    r = np.array([0, 0, 0, 1, 1, 2, 2, 2], dtype=np.uint64)
    c = np.array([1, 1, 1, 0, 0, 2, 2, 2], dtype=np.uint64)
    v = np.array([big, 1.0, 1.0, -0.0, 0.0, np.nan, 1.0, 2.0])
    csr = assemble_and_compare(oracle, 3, 3, r, c, v)
    assert csr.values()[0] == big           # (big + 1) + 1, not big + (1 + 1)
    assert csr.nnz() == 2 and np.isnan(csr.values()[1])


def test_coo_empty_and_all_cancelled(oracle):
    coo = sp.CooMatrix(4, 4)
    csr = sp.CsrMatrix.from_coo(coo)
    assert csr.nnz() == 0 and csr.rowptr().tolist() == [0] * 5
    assert (csr * np.ones(4)).tolist() == [0.0] * 4
    r = np.array([1, 1, 3, 3], dtype=np.uint64)
    c = np.array([2, 2, 0, 0], dtype=np.uint64)
    v = np.array([1.5, -1.5, 2.0, -2.0])
    csr = assemble_and_compare(oracle, 4, 4, r, c, v)
    assert csr.nnz() == 0


def test_coo_long_runs_and_skew(oracle):
    """one (row, col) repeated 20000 times, and one row holding half of all
    entries: the sort and the run sums must not depend on balance."""
    rng = np.random.default_rng(5)
    n = 60_000
    r = rng.integers(0, 50, n).astype(np.uint64)
    c = rng.integers(0, 4000, n).astype(np.uint64)
    r[:30_000] = 7
    r[30_000:50_000] = 9
    c[30_000:50_000] = 11
    v = rng.uniform(-1, 1, n)
    assemble_and_compare(oracle, 50, 4000, r, c, v)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SPAL_FUZZ_SEEDS", "12"))))
def test_coo_assembly_randomised(oracle, seed):
    """shapes, lengths, row skew, input order, duplicate and cancellation rates at random: arrays bit-exact
    (whatever route -- group sort of some geometry or the general one -- the assembly takes), then y = A*x."""
    rng = np.random.default_rng(7000 + seed)
    dtype = np.float32 if seed % 4 == 3 else np.float64
    nr = int(rng.choice([7, 300, 5000, 200_000, 1_500_000]))
    nc = int(rng.choice([5, 4000, 300_000, 1_500_000]))
    n = int(rng.choice([1, 1000, 200_000, 1_500_000]))
    kind = seed % 4
    if kind == 0:
        r = rng.integers(0, nr, n)
    elif kind == 1:      # power-law rows
        r = np.minimum((rng.pareto(1.2, n) * nr / 50).astype(np.int64), nr - 1)
    elif kind == 2:      # a handful of rows hold almost everything
        r = np.where(rng.random(n) < 0.9, rng.integers(0, min(nr, 5), n), rng.integers(0, nr, n))
    else:                # banded: columns near the row
        r = rng.integers(0, nr, n)
    c = rng.integers(0, nc, n) if kind != 3 else np.clip(r * nc // nr + rng.integers(-30, 30, n), 0, nc - 1)
    v = rng.uniform(-1, 1, n).astype(dtype)
    dup = rng.random(n) < rng.choice([0.0, 0.05, 0.5])          # duplicates of an earlier triplet ...
    src = rng.integers(0, n, n)
    r, c = np.where(dup, r[src], r), np.where(dup, c[src], c)
    cancel = dup & (rng.random(n) < 0.3)                         # ... some of which cancel it exactly
    v = np.where(cancel, -v[src], v).astype(dtype)
    v[rng.random(n) < 0.02] = 0.0
    order = rng.choice(3)
    if order == 1:
        o = np.lexsort((c, r))
        r, c, v = r[o], c[o], v[o]
    elif order == 2:
        o = np.lexsort((c, r))[::-1]
        r, c, v = r[o], c[o], v[o]
    csr = assemble_and_compare(oracle, nr, nc, r.astype(np.uint64), c.astype(np.uint64), v)
    x = rng.uniform(-1, 1, nc).astype(dtype)
    rp, ci, va = csr.rowptr(), csr.colind(), csr.values()
    bound = oracle.csr_abs_bound(rp, ci, va.astype(np.float64), x.astype(np.float64))
    assert_spmv_close(csr * x, oracle.csr_spmv(rp, ci, va, x), bound, 1e-10 if dtype == np.float64 else 1e-4)


def test_coo_wide_columns_take_the_unpacked_forms(oracle):
    """2^28 columns: column | row-in-group does not fit one word and column << 5 leaves no room for the place in the row, so
    the second pass writes key + column and the group kernel runs its loop form (rounds 1-3) -- offsets still from the two
    passes' counts.  Same bits as the oracle."""
    rng = np.random.default_rng(41)
    nr, nc, n = 300_000, (1 << 28) + 5, 2_500_000
    r = rng.integers(0, nr, n).astype(np.uint64)
    c = rng.integers(0, nc, n).astype(np.uint64)
    c[: n // 50] = c[n // 50: 2 * (n // 50)]            # duplicates (insertion-order sums) ...
    r[: n // 50] = r[n // 50: 2 * (n // 50)]
    v = rng.integers(-3, 4, n).astype(np.float64) * 0.37  # ... some cancelling
    dev = sp.CooMatrix.with_triplets(nr, nc, r, c, v).upload()
    got = dev.assemble_csr()
    d = dev.describe()
    assert d["last_route"] == "local_sort" and d["packed_payload"] == 0 and d["row_sort"] == 0 and d["offsets_from_counts"] == 1, d
    p, i, w = oracle.coo_to_csr(nr, nc, r, c, v)
    gp, gi, gw = got.download()
    assert np.array_equal(gp, p) and np.array_equal(gi, i) and np.array_equal(gw.view(np.uint64), w.view(np.uint64))


def test_coo_out_of_bounds_panics():
    with pytest.raises(sp.Panic):
        sp.CooMatrix.with_triplets(2, 2, [0, 2], [0, 0], np.array([1.0, 2.0]))
    coo = sp.CooMatrix(2, 2)
    with pytest.raises(sp.Panic):
        coo.push(0, 2, 1.0)


def test_config1_coo_to_csr_then_spmv(oracle):
    """BASELINE config 1: 10k x 10k, 100k random triplets -> CSR -> SpMV."""
    cfg = synth.CONFIGS[1]
    r, c, v = synth.coo(cfg["nrows"], cfg["ncols"], cfg["length"], synth.matrix_seed(1))
    csr = assemble_and_compare(oracle, cfg["nrows"], cfg["ncols"], r, c, v)
    assert 99_000 < csr.nnz() <= 100_000
    x = synth.vector(cfg["ncols"])
    y = csr * x
    y_ref = oracle.csr_spmv(csr.rowptr(), csr.colind(), csr.values(), x)
    assert_spmv_close(y, y_ref, oracle.csr_abs_bound(csr.rowptr(), csr.colind(), csr.values(), x), 1e-10)


def test_config5_scaled_assembly(oracle):
    """BASELINE config 5 generator (1 % duplicates, 0.1 % cancelling pairs) at
    2M entries, bit-exact against the oracle."""
    nr = nc = 200_000
    r, c, v = synth.coo(nr, nc, 2_000_000, synth.matrix_seed(5), 10, 1)
    csr = assemble_and_compare(oracle, nr, nc, r, c, v)
    assert csr.nnz() < 2_000_000


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_coo_local_sort_geometries(oracle, dtype):
    """Every geometry of the LDS local sort (groups of 256 ... 8 rows, capacities
    512 / 1024 / 1536 / 2048 entries), the general route (a group above 2048 entries)
    and very sparse rows (group starts by binary search), each bit-exact against
    the oracle.  Empty rows, duplicates (same (row, col) several times) and exact
    zeros in every case."""
    rng = np.random.default_rng(77)
    seen = set()
    relaunched = 0
    for nr, per_row in [(20_000, 1), (20_000, 3), (20_000, 6), (20_000, 12), (12_000, 24),
                        (8_000, 45), (4_000, 100), (300_000, 0.01), (20_001, 7), (20_000, 10),
                        (200, 600), (64, 1500), (300, 850)]:   # the last three: groups of 2 rows / 1 row, long rows
        n = max(1, int(nr * per_row))
        nc = 50 if per_row < 1 else 4 * max(1, int(per_row))     # few columns: many duplicates
        r = rng.integers(0, nr, n).astype(np.uint64)
        r[r % 11 == 3] = r[r % 11 == 3] // 11 * 11                 # empty rows, some fuller rows
        if per_row == 100:
            r[: 3000] = 17                                        # a row above the group capacity
        c = rng.integers(0, nc, n).astype(np.uint64)
        v = rng.integers(-3, 4, n).astype(dtype) * dtype(0.37)    # sums that cancel exactly do occur
        coo = sp.CooMatrix.with_triplets(nr, nc, r, c, v)
        dev = coo.upload()
        p, i, w = oracle.coo_to_csr(nr, nc, r, c, v)
        bits = np.uint64 if dtype == np.float64 else np.uint32
        # twice: the first assembly GUESSES the group kernel's capacity (mean + 6 sigma; too small a guess is caught
        # on the device and the kernel launched again), the second one knows the fullest group of the first
        for attempt in range(2):
            got = dev.assemble_csr()
            d = dev.describe()
            seen.add((d["last_route"], d["group_rows"], d["group_cap"]))
            relaunched += d["group_relaunches"]
            assert attempt == 0 or d["group_relaunches"] == 0, d
            gp, gi, gw = got.download()
            assert np.array_equal(gp, p) and np.array_equal(gi, i), d
            assert np.array_equal(gw.view(bits), w.view(bits)), d
            got.close()
        dev.close()
    assert relaunched >= 1, "no case took the too-small-guess path"
    assert ("general", 0, 0) in seen
    assert len({g[1] for g in seen}) >= 5, seen                  # 256, 128, 64, 32, ... rows per group
    assert {g[2] for g in seen} >= {512, 1024, 1536, 2048}, seen



def _coo_case(seed, n=1_500_000, nr=150_000, nc=90_000):
    rng = np.random.default_rng(seed)
    r = rng.integers(0, nr, n).astype(np.uint64)
    c = rng.integers(0, nc, n).astype(np.uint64)
    v = rng.integers(-3, 4, n).astype(np.float64) * 0.37
    src = rng.integers(0, n, n)
    dup = rng.random(n) < 0.05
    return nr, nc, np.where(dup, r[src], r), np.where(dup, c[src], c), v


def test_coo_lookback_backstop_takes_the_general_route(oracle, monkeypatch):
    """The look-back's spin bound forced to 0 (SPAL_COO_LOOKBACK_SPINS): the first group that has to wait gives up, the
    error flag comes back and the host repeats the assembly on the general route -- the result must still equal the
    oracle's bit for bit, and describe() must say which route produced it."""
    nr, nc, r, c, v = _coo_case(123)
    p, i, w = oracle.coo_to_csr(nr, nc, r, c, v)
    dev = sp.CooMatrix.with_triplets(nr, nc, r, c, v).upload()
    monkeypatch.setenv("SPAL_COO_LOOKBACK_SPINS", "0")
    got = dev.assemble_csr()
    d = dev.describe()
    assert d["last_route"] == "general" and d["lookback_gave_up"] == 1, d
    gp, gi, gw = got.download()
    assert np.array_equal(gp, p) and np.array_equal(gi, i) and np.array_equal(gw.view(np.uint64), w.view(np.uint64))
    got.close()
    monkeypatch.delenv("SPAL_COO_LOOKBACK_SPINS")
    got = dev.assemble_csr()                                   # the same handle, bound restored: the local sort again
    d = dev.describe()
    assert d["last_route"] == "local_sort" and d["lookback_gave_up"] == 1, d
    gp, gi, gw = got.download()
    assert np.array_equal(gp, p) and np.array_equal(gi, i) and np.array_equal(gw.view(np.uint64), w.view(np.uint64))
    got.close()
    dev.close()


@pytest.mark.parametrize("ticket", ["8", "1", "0"])
def test_coo_group_ids_by_ticket_or_by_block_index(oracle, monkeypatch, ticket):
    """Groups are handed out by device tickets in start order: by default from 8 class counters (round 4), with
    SPAL_COO_TICKET=1 from the single counter of round 3; SPAL_COO_TICKET=0 takes blockIdx (the dispatch-order assumption
    of round 2).  All must give the oracle's arrays; 20 assemblies of one handle agree."""
    monkeypatch.setenv("SPAL_COO_TICKET", ticket)
    nr, nc, r, c, v = _coo_case(321, n=3_000_000, nr=400_000)
    p, i, w = oracle.coo_to_csr(nr, nc, r, c, v)
    dev = sp.CooMatrix.with_triplets(nr, nc, r, c, v).upload()
    for _ in range(20):
        got = dev.assemble_csr()
        d = dev.describe()
        assert d["last_route"] == "local_sort" and d["ticket_mode"] == int(ticket), d
        gp, gi, gw = got.download()
        assert np.array_equal(gp, p) and np.array_equal(gi, i) and np.array_equal(gw.view(np.uint64), w.view(np.uint64))
        got.close()
    dev.close()


@pytest.mark.parametrize("env", [{}, {"SPAL_COO_NO_PACK": "1"}, {"SPAL_COO_NO_OFFSETS": "1"}])
@pytest.mark.parametrize("by_cols", [False, True])
def test_coo_two_pass_forms_agree(oracle, monkeypatch, env, by_cols):
    """Round 4: with exactly two radix passes the groups' offsets come from the passes' scanned counts (group_offsets) and
    the second pass writes column | row-in-group in one word.  Both switched off one by one (the round-3 forms: offsets
    from the sorted keys, key + column) must give the same bits, by rows and by columns."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    nr, nc, r, c, v = _coo_case(77, n=2_500_000, nr=300_000)
    dev = sp.CooMatrix.with_triplets(nr, nc, r, c, v).upload()
    if by_cols:
        p, i, w = oracle.coo_to_csc(nr, nc, r, c, v)
        got = dev.assemble_csc()
    else:
        p, i, w = oracle.coo_to_csr(nr, nc, r, c, v)
        got = dev.assemble_csr()
    d = dev.describe()
    assert d["last_route"] == "local_sort", d
    assert d["offsets_from_counts"] == (0 if "SPAL_COO_NO_OFFSETS" in env else 1), d
    assert d["packed_payload"] == (0 if env else 1), d
    gp, gi, gw = got.download()
    assert np.array_equal(gp, p) and np.array_equal(gi, i) and np.array_equal(gw.view(np.uint64), w.view(np.uint64))


def test_coo_rows_beyond_the_network_form(oracle):
    """Step 2 of the group kernel sorts rows of up to 16 entries in one thread's registers and takes rows of up to 256 by a
    wave each; a group with a longer row raises a flag and the host runs the kernel's other form (every entry counts its
    place for itself).  Rows of 17 ... 256 entries with duplicates inside them, then one row of 400: same bits as the
    oracle, the relaunch happens once and is remembered on the handle."""
    rng = np.random.default_rng(5)
    nr, nc, r, c, v = _coo_case(31, n=2_000_000, nr=250_000)
    extra_r, extra_c, extra_v = [], [], []
    for row, cnt in ((7, 17), (100_000, 56), (100_001, 120), (200_000, 200), (249_999, 236)):   # (+ the ~8 entries each row holds anyway)
        cols = rng.integers(0, nc, cnt)
        cols[cnt // 2:cnt // 2 + 5] = cols[:5]                  # duplicates inside the long row (insertion-order sums)
        extra_r.append(np.full(cnt, row)); extra_c.append(cols); extra_v.append(rng.integers(-3, 4, cnt) * 0.37)
    r1 = np.concatenate([r] + [e.astype(np.uint64) for e in extra_r])
    c1 = np.concatenate([c] + [e.astype(np.uint64) for e in extra_c])
    v1 = np.concatenate([v] + extra_v)
    perm = rng.permutation(r1.size)
    r1, c1, v1 = r1[perm], c1[perm], v1[perm]
    dev = sp.CooMatrix.with_triplets(nr, nc, r1, c1, v1).upload()
    got = dev.assemble_csr()
    d = dev.describe()
    assert d["last_route"] == "local_sort" and d["row_sort"] == 1, d   # (the first capacity guess may cost a relaunch; not the form)
    p, i, w = oracle.coo_to_csr(nr, nc, r1, c1, v1)
    gp, gi, gw = got.download()
    assert np.array_equal(gp, p) and np.array_equal(gi, i) and np.array_equal(gw.view(np.uint64), w.view(np.uint64))
    # one row of 400 entries: beyond the wave pass
    r2 = np.concatenate([r1, np.full(400, 123_456, dtype=np.uint64)])
    c2 = np.concatenate([c1, rng.integers(0, nc, 400).astype(np.uint64)])
    v2 = np.concatenate([v1, rng.integers(-3, 4, 400) * 0.37])
    dev2 = sp.CooMatrix.with_triplets(nr, nc, r2, c2, v2).upload()
    p, i, w = oracle.coo_to_csr(nr, nc, r2, c2, v2)
    for attempt in range(2):
        got = dev2.assemble_csr()
        d = dev2.describe()
        assert d["last_route"] == "local_sort" and d["row_sort"] == 0, d
        assert d["group_relaunches"] >= 1 if attempt == 0 else d["group_relaunches"] == 0, d
        gp, gi, gw = got.download()
        assert np.array_equal(gp, p) and np.array_equal(gi, i) and np.array_equal(gw.view(np.uint64), w.view(np.uint64))


def test_assembled_handle_plans_like_an_uploaded_one(oracle):
    """The assembly hands the CSR planner the column windows it saw on the way;
    the resulting plan (LDS windows, stream fractions) must be the plan the same
    matrix gets when it is uploaded from the host, and multiply bit-identically."""
    rng = np.random.default_rng(91)
    for n, per_row, window in [(40_000, 14, 2048), (100_000, 5, 600), (3_000, 40, 3_000), (70_001, 9, 70_001)]:
        rp, ci, va = synth.banded_csr(n, n, per_row, window, 17)
        rows = np.repeat(np.arange(n, dtype=np.uint64), np.diff(rp).astype(np.int64))
        perm = rng.permutation(rows.size)                       # insertion order: shuffled
        coo = sp.CooMatrix.with_triplets(n, n, rows[perm], ci[perm], va[perm])
        dcoo = coo.upload()
        got = dcoo.assemble_csr()
        assert dcoo.describe()["last_route"] == "local_sort"
        ref = sp.CsrMatrix(n, n, rp, ci, va).device()
        dg, dr = got.describe(), ref.describe()
        for k in ("kernel", "rows_per_tile", "blocks", "lds_x", "lds_window_bytes", "lds_row_fraction",
                  "stream_row_fraction", "index_bits"):
            assert dg[k] == dr[k], (k, dg, dr)
        gp, gi, gv = got.download()
        assert np.array_equal(gp, rp) and np.array_equal(gi, ci) and np.array_equal(gv, va)
        x = synth.vector(n)
        assert np.array_equal(got.spmv(x), ref.spmv(x))
        got.close()
        dcoo.close()


def test_full_size_config5_properties():
    """BASELINE config 5 at full size (50M triplets into 5M x 5M), checked through
    size-independent properties (the bit-exact oracle comparison lives in the
    smaller tests): CSR invariants, nothing stored that is zero, idempotence
    (assembling the result's own triplets reproduces it exactly), the product on
    the result equals the product summed straight from the triplets, and two
    assemblies give identical arrays."""
    torch = pytest.importorskip("torch")
    cfg = synth.CONFIGS[5]
    nr, length = cfg["nrows"], cfg["length"]
    r, c, v = synth.coo(nr, nr, length, synth.matrix_seed(5), cfg["dup_permille"], cfg["cancel_permille"])
    dcoo = sp.CooMatrix.with_triplets(nr, nr, r, c, v).upload()
    a = dcoo.assemble_csr()
    assert dcoo.describe()["last_route"] == "local_sort"
    rp, ci, va = a.download()
    nnz = int(rp[-1])
    assert rp[0] == 0 and nnz == ci.size == va.size and nnz < length       # duplicates merged / cancelled
    lens = np.diff(rp.astype(np.int64))
    assert lens.min() >= 0
    # columns strictly increasing inside every row (src/csr.rs:152-156)
    d = np.diff(ci.astype(np.int64))
    row_start = np.zeros(nnz, dtype=bool)
    row_start[rp[:-1][lens > 0].astype(np.int64)] = True
    assert np.all((d > 0) | row_start[1:])
    assert ci.max() < nr and not np.any(va == 0.0)                          # coo.rs:64
    # the product: straight from the triplets (float64 index_add, another order) vs the kernel
    x = torch.from_numpy(synth.vector(nr)).cuda()
    y = a.spmv_torch(x)
    rt, ct, vt = (torch.from_numpy(t.astype(np.int64) if t.dtype != np.float64 else t).cuda() for t in (r, c, v))
    y_direct = torch.zeros(nr, dtype=torch.float64, device="cuda").index_add_(0, rt, vt * x[ct])
    bound = torch.zeros(nr, dtype=torch.float64, device="cuda").index_add_(0, rt, (vt * x[ct]).abs())
    assert bool(torch.all((y - y_direct).abs() <= 1e-10 * bound + 1e-300))
    del rt, ct, vt, y_direct, bound
    # idempotence and determinism
    rows = np.repeat(np.arange(nr, dtype=np.uint64), lens)
    d2 = sp.CooMatrix.with_triplets(nr, nr, rows, ci, va).upload()
    b = d2.assemble_csr()
    rp2, ci2, va2 = b.download()
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(va2.view(np.uint64), va.view(np.uint64))
    a2 = dcoo.assemble_csr()
    rp3, ci3, va3 = a2.download()
    assert np.array_equal(rp3, rp) and np.array_equal(ci3, ci) and np.array_equal(va3.view(np.uint64), va.view(np.uint64))


def test_csc_handle_converted_on_the_device_scatters_over_row_tiles(oracle):
    """CSR -> CSC on the device (`DeviceCsr.to_csc`: the handle adopts the converted arrays) and the scatter path on the
    result: the row tiles are planned for adopted handles as for uploaded ones."""
    n = 200_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 41)
    x = synth.vector(n)
    dcsc = sp.CsrMatrix(n, n, rp, ci, va).device().to_csc()
    dcsc.set_option("kernel", 1)
    d = dcsc.describe()
    assert d["row_tiles"] == 1 and d["row_tile_rows"] == 4096, d
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    assert_spmv_close(dcsc.spmv(x), y_ref, oracle.csr_abs_bound(rp, ci, va, x), 1e-10)
    dcsc.set_option("row_tiles", 0)
    assert_spmv_close(dcsc.spmv(x), y_ref, oracle.csr_abs_bound(rp, ci, va, x), 1e-10)
