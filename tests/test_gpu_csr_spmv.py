"""GPU parity: CSR y = A*x through the C ABI vs the CPU oracle.

Tolerance: f64 1e-10 relative (north_star), f32 1e-4; both normwise (inf
norm) and componentwise against sum|a||x|."""
import numpy as np
import pytest

import spalinalg_amd as sp
import spal_synth as synth
from tests.util import assert_spmv_close, random_csr

pytestmark = pytest.mark.gpu
TOL = {np.dtype(np.float64): 1e-10, np.dtype(np.float32): 1e-4}


def check(oracle, rp, ci, va, x, ncols, **opts):
    a = sp.CsrMatrix(rp.size - 1, ncols, rp, ci, va)
    dev = a.device()
    for k, v in opts.items():
        dev.set_option(k, v)
    y = dev.spmv(x)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    bound = oracle.csr_abs_bound(rp, ci, va.astype(np.float64), np.nan_to_num(x.astype(np.float64), posinf=0, neginf=0))
    assert y.dtype == va.dtype
    assert_spmv_close(y, y_ref, bound, TOL[va.dtype])
    return dev


def test_reference_kat_vectors_g5(kats, oracle):
    g = kats["G5_csc_mul"]
    a = g["lhs"]
    vals = np.array(a["values"])
    rp, ci, rv = oracle.transpose(a["ncols"], a["nrows"], a["colptr"], a["rowind"], vals)
    m = sp.CsrMatrix(a["nrows"], a["ncols"], rp, ci, rv)
    for case in g["spmv"]:
        y = m * np.array(case["x"])
        assert y.tolist() == case["y"]          # small integers: exact
    g8 = kats["G8_dok_matrix"]
    m = sp.CsrMatrix(g8["nrows"], g8["ncols"], g8["rowptr"], g8["colind"], np.array(g8["values"]))
    assert (m @ np.array([1.0, 2.0, 3.0])).tolist() == [21.0, 15.0]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(1, 1), (1, 50), (50, 1), (37, 41), (300, 200), (5000, 3000)])
def test_random_small(oracle, dtype, shape):
    rng = np.random.default_rng(hash(shape) % 2**32)
    nr, nc = shape
    rp, ci, va = random_csr(rng, nr, nc, density=min(0.2, 8.0 / nc), dtype=dtype)
    x = rng.uniform(-1, 1, nc).astype(dtype)
    check(oracle, rp, ci, va, x, nc)


def test_empty_matrix_and_empty_rows(oracle):
    rp = np.zeros(101, dtype=np.uint64)
    a = sp.CsrMatrix(100, 7, rp, np.empty(0, dtype=np.uint64), np.empty(0))
    y = a * np.ones(7)
    assert y.tolist() == [0.0] * 100
    # only the last row stores something
    rp[-1] = 2
    a = sp.CsrMatrix(100, 7, rp, [1, 5], np.array([2.0, 3.0]))
    y = a * np.arange(7.0)
    assert y[:-1].tolist() == [0.0] * 99 and y[-1] == 2.0 + 15.0


def test_stream_kernel_is_bit_identical_to_the_reference_order(oracle):
    """Rows handled by the stream kernel are summed left to right with
    separately rounded mul and add, exactly like the reference
    (src/csr/ops/mul.rs:31-38): results are bit-for-bit the oracle's."""
    rng = np.random.default_rng(2)
    for dtype in (np.float64, np.float32):
        n = 300_000
        rp, ci, va = synth.banded_csr(n, n, 14, 4096, 11, dtype=dtype)
        x = synth.vector(n, dtype=dtype)
        dev = sp.CsrMatrix(n, n, rp, ci, va).device()
        d = dev.describe()
        assert d["kernel"] == "stream" and d["stream_row_fraction"] == 1.0 and d["index_bits"] == 16
        y = dev.spmv(x)
        assert np.array_equal(y, oracle.csr_spmv(rp, ci, va, x))
        for persistent, blocks in ((1, 512), (1, 8), (1, 1024), (0, 512)):
            dev.set_option("persistent", persistent)
            dev.set_option("nt_store", blocks == 8)
            dev.set_option("persistent_blocks", blocks)
            assert dev.describe()["persistent"] == persistent
            assert np.array_equal(dev.spmv(x), y)
        # ragged rows (0..24 entries), odd tile starts, a short last super-tile
        nr, nc = 70_001, 3000
        rp, ci, va = random_csr(rng, nr, nc, row_len=lambda r: r.integers(0, 25), dtype=dtype)
        x = rng.uniform(-1, 1, nc).astype(dtype)
        x[7] = 0.0
        dev = sp.CsrMatrix(nr, nc, rp, ci, va).device()
        d = dev.describe()
        assert d["kernel"] == "stream" and d["stream_row_fraction"] > 0.99
        y = dev.spmv(x)
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        assert np.array_equal(y, y_ref)
        assert np.array_equal(np.signbit(y), np.signbit(y_ref))     # -0.0 from a lone -v * 0.0 included
        dev.set_option("persistent", 1)
        for blocks in (512, 16, 40):
            dev.set_option("persistent_blocks", blocks)
            assert np.array_equal(dev.spmv(x), y_ref)


@pytest.mark.parametrize("maxlen,rpt", [(25, 64), (41, 32), (90, 16)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_stream_kernel_narrow_tiles_bit_identical(oracle, maxlen, rpt, dtype):
    """rows of up to ~64 entries stream too, in tiles of 32 or 16 rows; still
    bit-identical to the sequential reference order, in every kernel form."""
    rng = np.random.default_rng(maxlen)
    nr, nc = 50_003, 4000
    rp, ci, va = random_csr(rng, nr, nc, row_len=lambda r: r.integers(0, maxlen), dtype=dtype, empty_rows=0.02)
    x = rng.uniform(-1, 1, nc).astype(dtype)
    dev = sp.CsrMatrix(nr, nc, rp, ci, va).device()
    d = dev.describe()
    assert d["kernel"] == "stream" and d["rows_per_tile"] == rpt and d["stream_row_fraction"] > 0.95, d
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    tol = 1e-10 if dtype == np.float64 else 1e-4

    def same(y, frac):
        # rows of streamable super-tiles are bit-identical; the few super-tiles that fell
        # back to the vector path (a tile over 1024 entries) agree to rounding
        np.testing.assert_allclose(y, y_ref, rtol=tol, atol=tol)
        assert int((y != y_ref).sum()) <= int(round((1.0 - frac) * nr))
        if frac == 1.0:
            assert np.array_equal(y, y_ref)

    same(dev.spmv(x), d["stream_row_fraction"])
    for persistent, nt in ((1, 0), (1, 1), (0, 1)):
        dev.set_option("persistent", persistent)
        dev.set_option("nt_store", nt)
        same(dev.spmv(x), d["stream_row_fraction"])
    # forcing a narrower tile than needed is still exact
    if rpt > 16:
        dev.set_option("rows_per_tile", rpt // 2)
        d2 = dev.describe()
        assert d2["rows_per_tile"] == rpt // 2
        same(dev.spmv(x), d2["stream_row_fraction"])


@pytest.mark.parametrize("per_row,dtype,rpt,skew", [(16, np.float64, 64, 1), (32, np.float64, 32, 1), (64, np.float64, 16, 1),
                                                    (96, np.float64, 8, 1), (100, np.float64, 8, 0), (81, np.float32, 12, 0), (40, np.float64, 24, 0),
                                                    (32, np.float32, 32, 1), (15, np.float64, 64, 0)])
def test_stream_kernel_bank_skew_and_8_row_tiles(oracle, per_row, dtype, rpt, skew):
    """Rows whose length is a multiple of 128 bytes get skewed product strips (their lanes would sum through one
    LDS bank otherwise); rows of 65 ... 120 entries stream in tiles of 8 rows.  Both are layout / geometry only:
    every form stays bit-identical to the reference order."""
    n = 40_003
    rp, ci, va = synth.banded_csr(n, n, per_row, 2048, 40 + per_row, dtype=dtype)
    x = synth.vector(n, dtype=dtype)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    d = dev.describe()
    assert d["kernel"] == "stream" and d["rows_per_tile"] == rpt and d["skew"] == skew, d
    assert d["stream_row_fraction"] == 1.0 and d["overflow_tiles"] == 0, d
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    for sk in (skew, 1 - skew):
        dev.set_option("skew", sk)
        assert dev.describe()["skew"] == sk
        for persistent in (0, 1):
            dev.set_option("persistent", persistent)
            assert np.array_equal(dev.spmv(x), y_ref)
    dev.set_option("skew", -1)
    assert dev.describe()["skew"] == skew
    # the same rows with columns all over the matrix: x through L2 (32-bit columns), same strips
    rp, ci, va = synth.banded_csr(n, n, per_row, n, 41 + per_row, dtype=dtype)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    d = dev.describe()
    assert d["kernel"] == "stream" and d["skew"] == skew and d["lds_row_fraction"] == 0.0, d
    assert np.array_equal(dev.spmv(x), oracle.csr_spmv(rp, ci, va, x))


@pytest.mark.parametrize("lo,hi,rpt", [(0, 4, 256), (3, 9, 128)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_stream_kernel_tall_tiles_several_rows_per_lane(oracle, dtype, lo, hi, rpt):
    """rows of at most 4 / 8 entries (diagonal, tridiagonal, 5- and 7-point stencils) stream in tiles of 256 / 128
    rows, a lane summing four / two adjacent rows: bit-identical; a heavy row takes its tile to the overflow
    kernel in pieces of 64 rows."""
    rng = np.random.default_rng(8 + rpt)
    n = 70_001                                                   # odd: the last tile ends inside a lane's rows
    lens = rng.integers(lo, hi, n)                               # (empty rows included when lo == 0)
    lens[::1000] = 7
    rows = np.repeat(np.arange(n), lens)
    cols = np.clip(rows + rng.integers(-300, 300, rows.size), 0, n - 1)
    key = np.unique(rows * n + cols)
    rp = np.concatenate([[0], np.cumsum(np.bincount(key // n, minlength=n))]).astype(np.uint64)
    ci, va = (key % n).astype(np.uint64), rng.uniform(-1, 1, key.size).astype(dtype)
    x = synth.vector(n, dtype=dtype)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    d = dev.describe()
    assert d["kernel"] == "stream" and d["rows_per_tile"] == rpt and d["overflow_tiles"] == 0, d
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    for persistent, nt, skew in ((0, 0, 0), (1, 0, 0), (0, 1, 1), (1, 1, 1)):
        dev.set_option("persistent", persistent)
        dev.set_option("nt_store", nt)
        dev.set_option("skew", skew)
        assert np.array_equal(dev.spmv(x), y_ref)
    dev.set_option("rows_per_tile", 64)                          # the same matrix, a row per lane
    assert np.array_equal(dev.spmv(x), y_ref)
    # heavy rows: first row, a row in the middle of a lane's rows, last row
    rp = rp.astype(np.int64)
    ci, va = list(np.split(ci, rp[1:-1])), list(np.split(va, rp[1:-1]))
    heavy = ((0, 1500), (4001, 300), (n - 1, 1100))
    for r, k in heavy:
        ci[r] = np.sort(rng.choice(n, k, replace=False)).astype(np.uint64)
        va[r] = rng.uniform(-1, 1, k).astype(dtype)
    rp = np.concatenate([[0], np.cumsum([c.size for c in ci])]).astype(np.uint64)
    ci, va = np.concatenate(ci), np.concatenate(va)
    dev = check(oracle, rp, ci, va, x, n)
    d = dev.describe()
    pieces = sum(-(-(min(r // rpt * rpt + rpt, n) - r // rpt * rpt) // 64) for r, _ in heavy)
    assert d["rows_per_tile"] == rpt and d["overflow_tiles"] == pieces, d
    y, y_ref = dev.spmv(x), oracle.csr_spmv(rp, ci, va, x)
    light = np.ones(n, dtype=bool)
    for r, _ in heavy:
        light[r // rpt * rpt:r // rpt * rpt + rpt] = False
    assert np.array_equal(y[light], y_ref[light])


def test_stream_global_mode_bit_identical(oracle):
    """column windows too wide for LDS: the stream kernel gathers x through L2
    instead (32-bit columns), still summing each row in the reference order."""
    n = 300_000
    for window, dtype in ((20_000, np.float64), (None, np.float64), (30_000, np.float32)):
        rp, ci, va = synth.banded_csr(n, n, 14, window or n, 5, dtype=dtype)
        x = synth.vector(n, dtype=dtype)
        dev = sp.CsrMatrix(n, n, rp, ci, va).device()
        d = dev.describe()
        assert d["kernel"] == "stream" and d["stream_row_fraction"] == 1.0, d
        assert d["lds_row_fraction"] < (0.05 if window else 1e-9), d     # (banded: only the clamped edges fit LDS)
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        assert np.array_equal(dev.spmv(x), y_ref)
        dev.set_option("persistent", 1)
        assert np.array_equal(dev.spmv(x), y_ref)
        dev.set_option("stream_global", 0)          # the old behaviour: vector kernel, global gathers
        assert dev.describe()["kernel"] == "vector"
        np.testing.assert_allclose(dev.spmv(x), y_ref, rtol=1e-10 if dtype == np.float64 else 1e-4, atol=1e-6)


def test_long_rows_go_to_the_vector_kernel(oracle):
    rng = np.random.default_rng(140)
    nr, nc = 20_000, 5000
    rp, ci, va = random_csr(rng, nr, nc, row_len=lambda r: r.integers(40, 140), empty_rows=0.0)
    x = rng.uniform(-1, 1, nc)
    dev = check(oracle, rp, ci, va, x, nc)
    assert dev.describe()["kernel"] == "vector"


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_long_rows_read_16_bit_columns(oracle, dtype):
    """rows of hundreds of entries: the vector kernel, a wave per row, with 16-bit window-relative columns where the
    block's x window is in LDS -- the index width changes nothing in the arithmetic."""
    n = 30_011
    rp, ci, va = synth.banded_csr(n, n, 250, 4096, 77, dtype=dtype)
    rp = rp.astype(np.int64)
    ci, va = list(np.split(ci, rp[1:-1])), list(np.split(va, rp[1:-1]))
    rng = np.random.default_rng(3)
    for r in (5, 12_000, n - 1):          # rows whose columns are all over the matrix: their blocks gather from global memory
        ci[r] = np.sort(rng.choice(n, 250, replace=False)).astype(np.uint64)
    rp = np.concatenate([[0], np.cumsum([c.size for c in ci])]).astype(np.uint64)
    ci, va = np.concatenate(ci), np.concatenate(va)
    x = synth.vector(n, dtype=dtype)
    dev = check(oracle, rp, ci, va, x, n)
    d = dev.describe()
    assert d["kernel"] == "vector" and d["index_bits"] == 16 and 0.9 < d["lds_row_fraction"] < 1.0, d
    y16 = dev.spmv(x)
    dev.set_option("col16", 0)
    assert dev.describe()["index_bits"] == 32
    assert np.array_equal(dev.spmv(x), y16)
    dev.set_option("col16", 1)
    for threads in (512, 1024):
        dev.set_option("threads", threads)
        assert np.array_equal(dev.spmv(x), y16)


def test_stream_kernel_mixed_supertiles(oracle):
    """a few heavy rows take their tiles out of the stream path (overflow kernel); one wide row forces
    the global-gather mode."""
    rng = np.random.default_rng(4)
    n = 20_000
    rp, ci, va = synth.banded_csr(n, n, 14, 2048, 13)
    rp, ci, va = rp.astype(np.int64), list(np.split(ci, rp[1:-1].astype(np.int64))), list(np.split(va, rp[1:-1].astype(np.int64)))
    for r, k, span in [(5000, 3000, 4000), (5001, 1500, 4000), (12_345, 200, n), (19_999, 1100, 3000)]:
        lo = max(0, min(r - span // 2, n - span))
        ci[r] = (lo + np.sort(rng.choice(span, k, replace=False))).astype(np.uint64)
        va[r] = rng.uniform(-1, 1, k)
    lens = np.array([c.size for c in ci])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ci, va = np.concatenate(ci), np.concatenate(va)
    x = synth.vector(n)
    dev = check(oracle, rp, ci, va, x, n, kernel=2)
    d = dev.describe()
    assert d["kernel"] == "stream" and 0.5 < d["stream_row_fraction"] < 1.0 and d["overflow_tiles"] >= 3
    check(oracle, rp, ci, va, x, n, kernel=2, persistent=1)
    check(oracle, rp, ci, va, x, n, kernel=2, persistent=1, persistent_blocks=8)
    check(oracle, rp, ci, va, x, n, kernel=1)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("rpt", [64, 32, 16])
def test_oversized_tiles_go_to_the_overflow_kernel(oracle, dtype, rpt):
    """heavy rows among light ones: only their TILE leaves the stream path (a second kernel computes its
    rows); every other row keeps the reference's summation order bit for bit.  A tile leaves when it holds
    more than 1024 entries or a row of more than 128 (option stream_row_max)."""
    rng = np.random.default_rng(77 + rpt)
    n = 20_011                                   # ragged last super-tile and last tile
    rp, ci, va = synth.banded_csr(n, n, 14, 2048, 21, dtype=dtype)
    rp = rp.astype(np.int64)
    ci, va = list(np.split(ci, rp[1:-1])), list(np.split(va, rp[1:-1]))
    heavy = [(0, 1200), (63, 1500), (64, 1100), (700, 600), (701, 600), (4096, 3000), (4097 + rpt, 2500),
             (9000, 5000), (15_000, 129), (16_000, 128), (n - 1, 1300)]
    for r, k in heavy:
        span = max(4000, 2 * k)
        lo = max(0, min(r - span // 2, n - span))
        ci[r] = (lo + np.sort(rng.choice(span, k, replace=False))).astype(np.uint64)
        va[r] = rng.uniform(-1, 1, k).astype(dtype)
    lens = np.array([c.size for c in ci])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ci, va = np.concatenate(ci), np.concatenate(va)
    x = synth.vector(n, dtype=dtype)
    starts = np.arange(0, n, rpt)
    ends = np.minimum(starts + rpt, n)
    lens = np.diff(rp.astype(np.int64))
    longest = np.array([lens[s0:e0].max() for s0, e0 in zip(starts, ends)])
    big = ((rp[ends] - (rp[starts] & ~np.uint64(1))) > 1024) | (longest > 128)   # the plan's two criteria
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    light = np.ones(n, dtype=bool)
    for s0, e0 in zip(starts[big], ends[big]):
        light[s0:e0] = False
    for pers in (0, 1):
        dev = check(oracle, rp, ci, va, x, n, kernel=2, rows_per_tile=rpt, persistent=pers)
        d = dev.describe()
        assert d["kernel"] == "stream" and d["rows_per_tile"] == rpt
        assert d["overflow_tiles"] == int(big.sum()) > 0
        y = dev.spmv(x)
        assert np.array_equal(y[light], y_ref[light])
    # the row-length criterion is an option: at 1024 only the tiles that cannot fit the strip are left
    dev.set_option("stream_row_max", 1024)
    only_size = (rp[ends] - (rp[starts] & ~np.uint64(1))) > 1024
    assert dev.describe()["overflow_tiles"] == int(only_size.sum()) < int(big.sum())
    bound = oracle.csr_abs_bound(rp, ci, va.astype(np.float64), x.astype(np.float64))
    assert_spmv_close(dev.spmv(x), y_ref, bound, TOL[va.dtype])
    with pytest.raises(Exception):
        dev.set_option("stream_row_max", 0)


@pytest.mark.parametrize("lanes", [2, 4, 8, 16, 32, 64])
@pytest.mark.parametrize("unroll", [1, 2, 4])
def test_every_lane_width_and_unroll(oracle, lanes, unroll):
    """rows shorter, equal to and longer than L, incl. a 5000-entry row."""
    rng = np.random.default_rng(lanes * 10 + unroll)
    nr, nc = 3000, 6000
    lens = lambda r: r.choice([0, 1, 2, 7, 14, 16, 17, 31, 33, 64, 65, 130, 700])
    rp, ci, va = random_csr(rng, nr, nc, row_len=lens, empty_rows=0.0)
    # append one very long row
    long_cols = np.sort(rng.choice(nc, 5000, replace=False)).astype(np.uint64)
    rp = np.concatenate([rp, [rp[-1] + 5000]]).astype(np.uint64)
    ci = np.concatenate([ci, long_cols])
    va = np.concatenate([va, rng.uniform(-1, 1, 5000)])
    x = rng.uniform(-1, 1, nc)
    for threads in (512, 1024):
        for lds in (0, 1):
            check(oracle, rp, ci, va, x, nc, kernel=1, lanes_per_row=lanes, unroll=unroll, threads=threads, lds_x=lds)
    if unroll == 1:
        check(oracle, rp, ci, va, x, nc, kernel=2, lanes_per_row=lanes)
        check(oracle, rp, ci, va, x, nc, kernel=2, lanes_per_row=lanes, persistent=1)


@pytest.mark.parametrize("rows_per_block", [16, 64, 80, 512, 1008, 2048, 4096, 16384])
def test_rows_per_block_and_window_fallback(oracle, rows_per_block):
    """banded blocks use the LDS window; a few wide rows force the per-block
    global-gather fallback inside the same launch."""
    nr = nc = 40_000
    rp, ci, va = synth.banded_csr(nr, nc, 14, 1024, 99)
    rng = np.random.default_rng(3)
    # overwrite the column pattern of rows 10000..10009 with full-width rows
    for r in range(10_000, 10_010):
        lo = int(rp[r])
        ci[lo:lo + 14] = np.sort(rng.choice(nc, 14, replace=False))
    x = synth.vector(nc)
    dev = check(oracle, rp, ci, va, x, nc, kernel=1, rows_per_block=rows_per_block)
    d = dev.describe()
    assert d["rows_per_block"] == rows_per_block and d["kernel"] == "vector"


def test_nan_inf_in_x_stay_local(oracle):
    """NaN / Inf in x reach exactly the rows that reference them (idle lanes
    and clamped loads must not leak them)."""
    nr = nc = 5000
    rp, ci, va = synth.banded_csr(nr, nc, 14, 256, 5)
    x = synth.vector(nc)
    x[0] = np.nan
    x[2500] = np.inf
    x[4999] = -np.inf
    a = sp.CsrMatrix(nr, nc, rp, ci, va)
    for kernel, lanes in ((1, 4), (1, 16), (1, 64), (2, 16)):
        dev = a.device()
        dev.set_option("kernel", kernel)
        dev.set_option("lanes_per_row", lanes)
        y = dev.spmv(x)
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        assert np.array_equal(np.isnan(y), np.isnan(y_ref))
        assert np.array_equal(np.isinf(y), np.isinf(y_ref))
        ok = np.isfinite(y_ref)
        np.testing.assert_allclose(y[ok], y_ref[ok], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("window", [4096, None])
def test_config2_banded_and_uniform(oracle, dtype, window):
    """BASELINE config 2: 1M x 1M, 14 per row; banded (headline) and uniform."""
    n = 1_000_000
    rp, ci, va = synth.banded_csr(n, n, 14, window or n, synth.matrix_seed(2), dtype=dtype)
    x = synth.vector(n, dtype=dtype)
    dev = check(oracle, rp, ci, va, x, n)
    d = dev.describe()
    assert d["lds_x"] == (1 if window else 0)
    # banded: stream tiles out of the LDS window; uniform columns: the column-blocked kernel when x is beyond what an
    # XCD's L2 keeps (f64: 8 MB), else stream tiles with x through L2 (f32: 4 MB)
    assert d["kernel"] == ("cblock" if (window is None and dtype == np.float64) else "stream"), d
    check(oracle, rp, ci, va, x, n, kernel=1)
    dev = check(oracle, rp, ci, va, x, n, cblock=0)
    assert dev.describe()["kernel"] == "stream"          # ... and the stream kernels on request


def test_dimension_mismatch_panics():
    a = sp.CsrMatrix(2, 3, [0, 1, 3], [0, 1, 2], np.array([1.0, 2.0, 3.0]))
    with pytest.raises(sp.Panic):
        a * np.ones(2)
    with pytest.raises(sp.Panic):
        a.device().spmv(np.ones(4))


def test_autotune_keeps_results(oracle):
    torch = pytest.importorskip("torch")
    n = 400_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 9)
    x = synth.vector(n)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    d = dev.autotune(xd, yd, iters=5)
    assert all(t > 0 for t in d["autotune_us"]) and d["persistent"] in (0, 1) and d["nt_store"] in (0, 1)
    assert np.array_equal(dev.spmv_torch(xd).cpu().numpy(), oracle.csr_spmv(rp, ci, va, x))
    # a matrix the vector kernel handles: nothing to tune, still fine
    rp, ci, va = synth.banded_csr(n, n, 14, n, 9)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    dev.set_option("kernel", 1)
    assert dev.autotune(xd, yd, iters=2)["autotune_us"] == [0.0] * 4


def test_device_path_with_torch_stream(oracle):
    """the timed entry point: device pointers + torch's current stream."""
    torch = pytest.importorskip("torch")
    n = 200_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 1234)
    x = synth.vector(n)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    xd = torch.from_numpy(x).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        yd = dev.spmv_torch(xd)
        yd2 = dev.spmv_torch(xd)
    s.synchronize()
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    np.testing.assert_allclose(yd.cpu().numpy(), y_ref, rtol=1e-10, atol=1e-13)
    assert torch.equal(yd, yd2)        # deterministic run to run


def test_concurrent_callers_on_one_handle(oracle):
    """`&CsrMatrix` is Sync in the reference; the handle's device entry point may be
    called from several threads at once (each on its own stream), the host-vector
    call serialises internally.  Also: many create / assemble / destroy cycles
    (pooled streams, cached device blocks) keep giving the same results."""
    torch = pytest.importorskip("torch")
    import threading
    n = 120_000
    rp, ci, va = synth.banded_csr(n, n, 14, 2048, 77)
    x = synth.vector(n)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    xd = torch.from_numpy(x).cuda()
    results, errors = {}, []

    def worker(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(40):
                    yd = dev.spmv_torch(xd)
            s.synchronize()
            results[k] = yd.cpu().numpy()
            for _ in range(5):
                results[("host", k)] = dev.spmv(x)          # host-vector convenience call
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k, y in results.items():
        assert np.array_equal(y, y_ref), k                    # stream kernel: bit-identical
    # handle churn
    r, c, v = synth.coo(5_000, 5_000, 60_000, 9)
    p0, i0, w0 = oracle.coo_to_csr(5_000, 5_000, r, c, v)
    for _ in range(60):
        d = sp.CooMatrix.with_triplets(5_000, 5_000, r, c, v).upload()
        got = d.assemble_csr()
        gp, gi, gw = got.download()
        assert np.array_equal(gp, p0) and np.array_equal(gi, i0) and np.array_equal(gw.view(np.uint64), w0.view(np.uint64))
        got.close()
        d.close()


def test_full_size_config3_properties():
    """BASELINE config 3 (10M x 10M, 140M nnz) at full size: checked through
    size-independent properties (the oracle comparison lives in the smaller
    tests): y for x = e (row sums), linearity, and determinism."""
    torch = pytest.importorskip("torch")
    n = 10_000_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(3))
    dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    ones = torch.ones(n, dtype=torch.float64, device="cuda")
    y1 = dev.spmv_torch(ones).cpu().numpy()
    rowsum = va.reshape(n, 14).sum(axis=1)
    np.testing.assert_allclose(y1, rowsum, rtol=0, atol=1e-13)
    x = torch.from_numpy(synth.vector(n)).cuda()
    z = torch.from_numpy(synth.vector(n, seed=77)).cuda()
    ax, az = dev.spmv_torch(x), dev.spmv_torch(z)
    comb = dev.spmv_torch(2.0 * x - 0.5 * z)
    assert torch.allclose(comb, 2.0 * ax - 0.5 * az, rtol=0, atol=1e-12)
    assert torch.equal(dev.spmv_torch(x), ax)
    # spot rows against a direct numpy evaluation
    xs = x.cpu().numpy()
    for r in (0, 1, 4_999_999, 9_999_999):
        lo, hi = int(rp[r]), int(rp[r + 1])
        assert abs(float(ax[r]) - float(np.dot(va[lo:hi], xs[ci[lo:hi].astype(np.int64)]))) < 1e-13


def test_single_process_multi_gpu_context(oracle):
    """the spal_mg_* exports (one process driving the node's GPUs): on a 1-GPU
    box only ngpus = 1 can run, which still exercises partition, shard upload,
    the resident buffers and the gather layout; more GPUs than visible is refused."""
    n = 150_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 17)
    x = synth.vector(n)
    a = sp.CsrMatrix(n, n, rp, ci, va)
    mg = sp.MultiGpuCsr(a, 1)
    assert mg.partition().tolist() == [0, n]
    y = mg.spmv(x)
    assert np.array_equal(y, oracle.csr_spmv(rp, ci, va, x))       # same kernel, same bits
    with pytest.raises(sp.Panic):
        mg.spmv(x[:-1])
    mg.close()
    if sp.device_count() == 1:
        with pytest.raises(sp.Panic):
            sp.MultiGpuCsr(a, 2)


def _patchwork(rng, dtype):
    """A matrix stitched from row sections of very different character (empty,
    sparse, 14/row banded, 60/row, a few rows of hundreds or thousands of entries,
    narrow / wide / full-width column windows), so that one launch mixes every
    super-tile mode the planner has."""
    ncols = int(rng.integers(2_000, 120_000))
    rows_c, rows_v = [], []
    for _ in range(int(rng.integers(2, 9))):
        nr = int(rng.integers(1, 6_000))
        kind = rng.integers(0, 7)
        per = [0, int(rng.integers(1, 4)), 14, int(rng.integers(20, 70)), int(rng.integers(150, 400)),
               int(rng.integers(1, 30)), 14][kind]
        span = [1, ncols, 4096, 2048, ncols, 20_000, ncols][kind]
        span = min(max(span, per + 1), ncols)
        if kind == 4:
            nr = int(rng.integers(1, 12))
        for r in range(nr):
            k = per if kind != 5 else int(rng.integers(0, per + 1))
            if kind == 4 and r == 0 and ncols > 3_000:
                k = int(rng.integers(1_100, 3_000))          # a row longer than a whole stream tile
            lo = int(rng.integers(0, ncols - span + 1))
            cols = lo + np.sort(rng.choice(span, min(k, span), replace=False))
            rows_c.append(cols.astype(np.uint64))
            rows_v.append(rng.uniform(-1, 1, cols.size).astype(dtype))
    lens = np.array([c.size for c in rows_c])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ci = np.concatenate(rows_c) if lens.sum() else np.zeros(0, np.uint64)
    va = np.concatenate(rows_v) if lens.sum() else np.zeros(0, dtype)
    return rp, ci, va, ncols


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_patchwork_matrices_all_planner_modes(oracle, dtype):
    """30 seeded patchwork matrices, each through the automatic plan, the forced
    stream kernel (plain and persistent) and the forced vector kernel."""
    rng = np.random.default_rng(2024)
    kernels = set()
    for _ in range(30):
        rp, ci, va, ncols = _patchwork(rng, dtype)
        x = rng.uniform(-1, 1, ncols).astype(dtype)
        dev = check(oracle, rp, ci, va, x, ncols)
        kernels.add(dev.describe()["kernel"])
        check(oracle, rp, ci, va, x, ncols, kernel=2, persistent=0)
        check(oracle, rp, ci, va, x, ncols, kernel=2, persistent=1, stream_global=0)
        check(oracle, rp, ci, va, x, ncols, kernel=1)
    assert "stream" in kernels


def _stencil(m, points, dtype, rng):
    """m^3 grid, 7- or 27-point stencil, row-major numbering: the columns of a row sit in a few
    narrow clusters far apart (offsets +-1, +-m, +-m^2)."""
    n = m ** 3
    idx = np.arange(n, dtype=np.int64)
    i, j, k = idx // (m * m), (idx // m) % m, idx % m
    offs = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)
            if points == 27 or abs(a) + abs(b) + abs(c) <= 1]
    cols = np.stack([idx + a * m * m + b * m + c for a, b, c in offs], 1)
    valid = np.stack([(i + a >= 0) & (i + a < m) & (j + b >= 0) & (j + b < m) & (k + c >= 0) & (k + c < m)
                      for a, b, c in offs], 1)
    rp = np.concatenate([[0], np.cumsum(valid.sum(1))]).astype(np.uint64)
    ci = cols[valid].astype(np.uint64)
    return n, rp, ci, rng.uniform(-1, 1, ci.size).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_paged_x_window_stencils_bit_identical(oracle, dtype):
    """The LDS x window is a set of 256-column pages, not one interval: stencil matrices
    (column span of a super-tile = 2 m^2, far beyond LDS; distinct pages: a dozen) stream out
    of LDS like a band does, in every form, bit-identical to the reference order.  Also: a
    vector x that is not 16-byte aligned, and a column count that ends inside a page."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(31)
    for m, points in ((61, 7), (47, 27)):
        n, rp, ci, va = _stencil(m, points, dtype, rng)
        x = rng.uniform(-1, 1, n).astype(dtype)
        dev = sp.CsrMatrix(n, n, rp, ci, va).device()
        d = dev.describe()
        assert d["kernel"] == "stream" and d["stream_row_fraction"] == 1.0 and d["lds_row_fraction"] == 1.0, d
        assert d["lds_window_bytes"] <= 120 * 1024
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        assert np.array_equal(dev.spmv(x), y_ref)
        if d["lds_window_bytes"] > 48 * 1024:      # the one-workgroup-per-CU budget was chosen: the small one must work too
            dev.set_option("window_pages", 24 if dtype == np.float64 else 48)
            d2 = dev.describe()
            assert d2["lds_window_bytes"] <= 48 * 1024 and d2["stream_row_fraction"] == 1.0, d2
            assert np.array_equal(dev.spmv(x), y_ref)
            dev.set_option("window_pages", 0)
        for persistent, nt in ((1, 0), (1, 1), (0, 1)):
            dev.set_option("persistent", persistent)
            dev.set_option("nt_store", nt)
            assert np.array_equal(dev.spmv(x), y_ref)
        # x at an odd element offset: not 16-byte aligned, the element-wise staging path
        big = torch.zeros(n + 3, dtype=torch.float64 if dtype == np.float64 else torch.float32, device="cuda")
        big[1:n + 1].copy_(torch.from_numpy(x))
        yd = dev.spmv_torch(big[1:n + 1])
        assert np.array_equal(yd.cpu().numpy(), y_ref)
    # a band of 8192 columns: 36 pages, beyond the two-workgroups-per-CU budget of f64 (24 pages) but
    # inside the one-workgroup-per-CU budget (60), which the planner prefers to x through L2
    n = 120_000
    rp, ci, va = synth.banded_csr(n, n, 14, 8192, 5, dtype=dtype)
    x = synth.vector(n, dtype=dtype)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    d = dev.describe()
    assert d["kernel"] == "stream" and d["lds_row_fraction"] == 1.0, d
    assert (d["lds_window_bytes"] > 48 * 1024) == (dtype == np.float64), d
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    assert np.array_equal(dev.spmv(x), y_ref)
    dev.set_option("persistent", 1)
    assert np.array_equal(dev.spmv(x), y_ref)
    # columns too scattered for any page budget: the stream kernel gathers through L2 instead
    n = 150_000
    rp, ci, va = synth.banded_csr(n, n, 14, n, 5, dtype=dtype)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    d = dev.describe()
    assert d["kernel"] == "stream" and d["lds_row_fraction"] == 0.0, d
    x = synth.vector(n, dtype=dtype)
    assert np.array_equal(dev.spmv(x), oracle.csr_spmv(rp, ci, va, x))
    # two bands far apart + a last page that x ends inside of (ncols = 5 * 256 + 3 beyond the band)
    nr, nc = 30_000, 900_003
    lens = rng.integers(0, 20, nr)
    rows_c = []
    for r in range(nr):
        k = int(lens[r])
        left = np.sort(rng.choice(800, k // 2, replace=False)) + r
        right = np.sort(rng.choice(500, k - k // 2, replace=False)) + (nc - 500 if r % 3 == 0 else 600_000 + r)
        rows_c.append(np.concatenate([left, right]).astype(np.uint64))
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ci = np.concatenate(rows_c)
    va = rng.uniform(-1, 1, ci.size).astype(dtype)
    x = rng.uniform(-1, 1, nc).astype(dtype)
    dev = sp.CsrMatrix(nr, nc, rp, ci, va).device()
    d = dev.describe()
    assert d["kernel"] == "stream" and d["lds_row_fraction"] > 0.9, d
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    assert np.array_equal(dev.spmv(x), y_ref)
    dev.set_option("persistent", 1)
    assert np.array_equal(dev.spmv(x), y_ref)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_x_vector_ends_inside_a_page(oracle, dtype):
    """column counts around the 16-byte vector and 256-column page boundaries, every row
    touching the first and the last column; x both aligned and at an odd element offset."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(8)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    for nc in (1, 2, 3, 4, 5, 7, 255, 256, 257, 511, 513, 1023, 1025):
        nr = 300
        rows = []
        for r in range(nr):
            inner = rng.choice(nc, min(nc, int(rng.integers(0, 6))), replace=False) if nc > 2 else np.empty(0, np.int64)
            rows.append(np.unique(np.concatenate([[0, nc - 1], inner])).astype(np.uint64))
        rp = np.concatenate([[0], np.cumsum([c.size for c in rows])]).astype(np.uint64)
        ci = np.concatenate(rows)
        va = rng.uniform(-1, 1, ci.size).astype(dtype)
        x = rng.uniform(-1, 1, nc).astype(dtype)
        dev = sp.CsrMatrix(nr, nc, rp, ci, va).device()
        dev.set_option("kernel", 2)
        assert dev.describe()["kernel"] == "stream"
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        assert np.array_equal(dev.spmv(x), y_ref), nc
        big = torch.full((nc + 2,), float("nan"), dtype=tdt, device="cuda")   # NaN guards either side
        big[1:nc + 1].copy_(torch.from_numpy(x))
        assert np.array_equal(dev.spmv_torch(big[1:nc + 1]).cpu().numpy(), y_ref), nc
        dev.set_option("persistent", 1)
        assert np.array_equal(dev.spmv(x), y_ref), nc


def test_device_entry_point_is_graph_capturable(oracle):
    """`spal_csr_spmv_dev_*` only enqueues work on the caller's stream (no allocation, no
    synchronisation), so a launch-bound loop of products can be captured in a HIP graph and
    replayed: here y2 = A (A x) for a small matrix, 8 pairs per graph launch."""
    torch = pytest.importorskip("torch")
    n = 50_000
    rp, ci, va = synth.banded_csr(n, n, 14, 1024, 19)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    x = torch.from_numpy(synth.vector(n)).cuda()
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dev.spmv_torch(x, out=y1)          # warm-up outside the capture (module load, LDS attribute)
        dev.spmv_torch(y1, out=y2)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(8):
            dev.spmv_torch(x, out=y1)
            dev.spmv_torch(y1, out=y2)
    y1.zero_()
    y2.zero_()
    g.replay()
    torch.cuda.synchronize()
    xh = x.cpu().numpy()
    r1 = oracle.csr_spmv(rp, ci, va, xh)
    r2 = oracle.csr_spmv(rp, ci, va, r1)
    assert np.array_equal(y1.cpu().numpy(), r1) and np.array_equal(y2.cpu().numpy(), r2)
    x.mul_(2.0)                               # new input, same graph
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(y1.cpu().numpy(), oracle.csr_spmv(rp, ci, va, 2.0 * xh))


def test_tiles_per_wave_8_and_rows_per_tile(oracle):
    """tiles_per_wave = 8 exists for 64-row tiles only: alone it gives correct (bit-identical) y, combined with
    another tile height it is refused -- in either order -- instead of launching a plan built for other rows."""
    n = 150_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 31)
    x = synth.vector(n)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    dev.set_option("kernel", 2)
    dev.set_option("tiles_per_wave", 8)
    assert dev.describe()["rows_per_block"] == 2048
    assert np.array_equal(dev.spmv(x), y_ref)
    dev.set_option("rows_per_tile", 64)
    assert np.array_equal(dev.spmv(x), y_ref)
    with pytest.raises(sp.Panic):
        dev.set_option("rows_per_tile", 32)
    assert np.array_equal(dev.spmv(x), y_ref)           # the refused option left the plan alone
    dev.set_option("tiles_per_wave", 4)
    dev.set_option("rows_per_tile", 32)
    with pytest.raises(sp.Panic):
        dev.set_option("tiles_per_wave", 8)
    assert dev.describe()["rows_per_tile"] == 32 and np.array_equal(dev.spmv(x), y_ref)
    for bad in (("prefetch", 3), ("slide", 2), ("place_tries", 99), ("uniform_rows", 2), ("diag", 256)):
        with pytest.raises(sp.Panic):
            dev.set_option(*bad)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("gen,rpt", [("banded14", 0), ("ragged", 0), ("banded30", 32), ("banded60", 16), ("banded100", 8),
                                     ("banded5", 64), ("ragged", 24), ("ragged", 12)])
def test_sliding_window_kernel_bit_identical(oracle, dtype, gen, rpt):
    """csr_spmv_slide (bands: ring x window that slides from step to step, counted load waits) against the oracle,
    bit for bit: uniform rows (rowptr not read) and ragged ones, every tile height, with the knobs that change how
    it walks (runs dealt round-robin, grid size, non-temporal stores), and the same plan launched on the
    one-super-tile-per-workgroup kernels (slide_on = 0: they read the ring-encoded columns too)."""
    n = 180_000 + 333
    if gen == "ragged":
        rp, ci, va = synth.ragged_csr(n, n, 4096, 77, dtype=dtype)
    else:
        rp, ci, va = synth.banded_csr(n, n, int(gen[6:]), 4096, 77, dtype=dtype)
    x = synth.vector(n, dtype=dtype)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    if rpt:
        dev.set_option("rows_per_tile", rpt)
    dev.set_option("slide_on", 1)     # (by name: also where the tiles are too ragged for the plan to pick it by itself, round 4)
    d = dev.describe()
    assert d["kernel"] == "stream" and d["slide"] == 1 and d["ring_pages"] > 0 and d["stream_row_fraction"] > 0.95
    assert (d["uniform_row_fraction"] == 1.0) == (gen != "ragged")
    # (ragged rows at 64 per tile: the few tiles above 1024 entries are taken in halves by the sliding kernel; on the
    #  one-super-tile-per-workgroup kernels of the option sweep below they go to the overflow kernel, whose rows are
    #  tree-summed -- those rows to rounding, all others bit for bit)
    exact = d["stream_row_fraction"] == 1.0
    assert exact or (gen == "ragged" and rpt == 0 and d["split_tiles"] > 0 and d["overflow_tiles"] == 0)
    if not exact:
        assert np.array_equal(dev.spmv(x), y_ref)      # the sliding kernel: every row bit for bit
    bound = None if exact else oracle.csr_abs_bound(rp, ci, va, x)

    def same(y):
        if exact:
            return np.array_equal(y, y_ref)
        diff = y != y_ref
        tol = 1e-10 if dtype == np.float64 else 1e-4
        return diff.mean() < 0.05 and np.all(np.abs(y.astype(np.float64) - y_ref) <= tol * bound + 1e-300)

    assert same(dev.spmv(x))
    for opts in ({"slide_run": 3}, {"slide_run": 0, "persistent_blocks": 64}, {"persistent_blocks": 4096, "nt_store": 1},
                 {"persistent_blocks": 0, "uniform_rows": 0}, {"uniform_rows": 1, "slide_on": 0}, {"slide_on": 1, "slide": 0},
                 {"slide": -1, "prefetch": 2, "slide_on": 0}):
        for k, v in opts.items():
            dev.set_option(k, v)
        assert same(dev.spmv(x)), opts


def test_sliding_window_jumps_and_synchronous_page_loads(oracle):
    """windows that do not slide smoothly: block-diagonal sections whose column ranges jump forwards and BACKWARDS
    between steps (entering pages on both sides, disjoint windows, more pages than a thread can prefetch), a
    matrix whose last page is cut by ncols (odd, so x ends inside a 16-byte vector), empty steps in between."""
    rng = np.random.default_rng(5)
    n, nc = 40_000, 50_001
    rowptr, cols, = [0], []
    starts = [0, 30_000, 2_000, 45_000, 44_000, 10_000, 49_000 - 2048, 0]
    for r in range(n):
        sec = (r // 2048) % len(starts)      # (sections of two super-tiles: a super-tile's pages stay one run)
        if (r // 700) % 9 == 4:          # a stretch of empty rows
            rowptr.append(rowptr[-1])
            continue
        base = min(starts[sec] + (r % 2048), nc - 2049)
        k = int(rng.integers(1, 20))
        c = np.sort(rng.choice(2049, size=k, replace=False)) + base
        c = np.minimum(c, nc - 1)
        c = np.unique(c)
        cols.append(c)
        rowptr.append(rowptr[-1] + c.size)
    ci = np.concatenate(cols).astype(np.uint64)
    rp = np.asarray(rowptr, dtype=np.uint64)
    for dtype in (np.float64, np.float32):
        va = rng.uniform(-1, 1, ci.size).astype(dtype)
        x = rng.uniform(-1, 1, nc).astype(dtype)
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        dev = sp.CsrMatrix(n, nc, rp, ci, va).device()
        d = dev.describe()
        assert d["slide"] == 1, d
        assert np.array_equal(dev.spmv(x), y_ref)
        dev.set_option("slide_run", 5)
        assert np.array_equal(dev.spmv(x), y_ref)
        dev.set_option("slide_on", 0)
        assert np.array_equal(dev.spmv(x), y_ref)


def test_sliding_window_unaligned_x_falls_back(oracle):
    """the sliding kernel loads pages of x as 16-byte vectors: an x that is not 16-byte aligned is served by the
    one-super-tile-per-workgroup kernel on the same (ring-encoded) plan -- same bits."""
    torch = pytest.importorskip("torch")
    n = 120_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 3)
    x = synth.vector(n)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    assert dev.describe()["slide"] == 1
    big = torch.zeros(n + 8, dtype=torch.float64, device="cuda")
    xs = big[1:n + 1]                       # 8 bytes off a 16-byte boundary
    xs.copy_(torch.from_numpy(x))
    assert xs.data_ptr() % 16 == 8
    y = dev.spmv_torch(xs)
    assert np.array_equal(y.cpu().numpy(), oracle.csr_spmv(rp, ci, va, x))
    with pytest.raises(sp.Panic):           # y = A * x is not computed in place
        dev.spmv_torch(xs, out=xs)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("window,gen,rpt", [(16384, "banded14", 0), (40000, "banded14", 0), (16384, "ragged", 0),
                                            (20000, "banded30", 32), (16384, "banded60", 16), (47000, "banded14", 0)])
def test_wide_bands_in_column_panels_bit_identical(oracle, dtype, window, gen, rpt):
    """a band wider than the LDS window: csr_spmv_panel keeps the entries in registers and walks the x window in
    panels; every product is formed once and rows are summed left to right, so the result equals the oracle's
    bit for bit -- as does the same plan with the panel kernel off (x gathered through L2)."""
    n = 150_000 + 17
    if gen == "ragged":
        rp, ci, va = synth.ragged_csr(n, n, window, 5, dtype=dtype)
    else:
        rp, ci, va = synth.banded_csr(n, n, int(gen[6:]), window, 5, dtype=dtype)
    x = synth.vector(n, dtype=dtype)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    if rpt:
        dev.set_option("rows_per_tile", rpt)
    d = dev.describe()
    assert d["kernel"] == "stream" and d["slide"] == 0, d
    if dtype == np.float64 or window >= 20000:
        assert d["panel_tiles"] > 0, d           # (f32 at 16384 columns: 64 pages of f32 still fit LDS)
    exact = d["overflow_tiles"] == 0

    def same(y):
        if exact:
            return np.array_equal(y, y_ref)
        bound = oracle.csr_abs_bound(rp, ci, va, x)
        tol = 1e-10 if dtype == np.float64 else 1e-4
        return (y != y_ref).mean() < 0.05 and np.all(np.abs(y.astype(np.float64) - y_ref) <= tol * bound + 1e-300)

    assert same(dev.spmv(x))
    for opts in ({"panel_on": 0}, {"panel_on": 1, "persistent": 1}, {"persistent": 0, "nt_store": 1}, {"panel_pages": 0}):
        for k, v in opts.items():
            dev.set_option(k, v)
        assert same(dev.spmv(x)), opts
    assert dev.describe()["panel_tiles"] == 0      # panel_pages = 0: no super-tile is flagged


def test_column_panels_at_the_16_bit_limit(oracle):
    """the panel kernel reads 16-bit columns relative to the first column of a super-tile's span: spans up to 255 pages
    (65 280 columns) qualify, wider ones gather x through L2 -- both bit-identical"""
    n = 200_000
    x = synth.vector(n)
    for window, panels in ((60_000, True), (66_000, False)):
        rp, ci, va = synth.banded_csr(n, n, 14, window, 8)
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        dev = sp.CsrMatrix(n, n, rp, ci, va).device()
        dev.set_option("panel_pages", 255)
        d = dev.describe()
        assert (d["panel_tiles"] > 0) == panels, d
        assert np.array_equal(dev.spmv(x), y_ref)
        with pytest.raises(sp.Panic):
            dev.set_option("panel_pages", 256)
        dev.close()


def test_row_blocks_beyond_32_bit_entry_offsets(oracle, monkeypatch):
    """More stored entries than one set of 32-bit device offsets addresses (the reference's offsets are usize,
    src/csr.rs:66-72): the handle keeps the matrix as row blocks.  SPAL_CSR_PART_ENTRIES lowers the limit so the path
    runs on a small matrix: products (host and device entry points), download, options, autotune and the refusal of
    the one conversion that cannot be split.  (The real limit: tools/lab.py huge, 4.5e9 entries.)"""
    import torch
    monkeypatch.setenv("SPAL_CSR_PART_ENTRIES", "300000")
    for dtype, gen in ((np.float64, "banded"), (np.float32, "ragged")):
        n = 200_000 + 37
        if gen == "banded":
            rp, ci, va = synth.banded_csr(n, n, 14, 4096, 3, dtype=dtype)
        else:
            rp, ci, va = synth.ragged_csr(n, n, 4096, 3, dtype=dtype)
        x = synth.vector(n, dtype=dtype)
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        dev = sp.CsrMatrix(n, n, rp, ci, va).device()
        d = dev.describe()
        nparts = -(-int(rp[-1]) // 300_000)
        assert d["kernel"] == "row_blocks" and d["parts"] >= nparts and d["nnz"] == int(rp[-1])
        cuts = d["part_rows"]
        assert cuts[0] == 0 and cuts[-1] == n and all(b > a for a, b in zip(cuts, cuts[1:]))
        assert all(int(rp[b]) - int(rp[a]) <= 300_000 for a, b in zip(cuts, cuts[1:]))
        assert dev.shape() == (n, n, int(rp[-1]))
        bits = np.uint64 if dtype == np.float64 else np.uint32
        if gen == "banded":     # every block streams: bit-identical
            assert np.array_equal(dev.spmv(x).view(bits), y_ref.view(bits))
        else:
            bound = oracle.csr_abs_bound(rp, ci, va, x)
            assert_spmv_close(dev.spmv(x), y_ref, bound, 1e-4)
        xt = torch.from_numpy(x).cuda()
        yt = torch.full((n,), float("nan"), dtype=xt.dtype, device="cuda")
        dev.spmv_torch(xt, yt)
        torch.cuda.synchronize()
        assert np.array_equal(yt.cpu().numpy().view(bits), dev.spmv(x).view(bits))
        rp2, ci2, va2 = dev.download()
        assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(va2.view(bits), va.view(bits))
        dev.set_option("rows_per_tile", 32)
        assert dev.describe()["part0"]["rows_per_tile"] == 32
        assert np.array_equal(dev.spmv(x).view(bits), yt.cpu().numpy().view(bits)) or gen == "ragged"
        dev.autotune(xt, yt, iters=3)
        torch.cuda.synchronize()
        y3 = dev.spmv(x)
        if gen == "banded":
            assert np.array_equal(y3.view(bits), y_ref.view(bits))
        with pytest.raises(sp.SpalError):
            dev.to_csc()
        dev.close()
    monkeypatch.delenv("SPAL_CSR_PART_ENTRIES")
    rp, ci, va = synth.banded_csr(50_000, 50_000, 14, 4096, 3)
    one = sp.CsrMatrix(50_000, 50_000, rp, ci, va).device()
    assert one.describe()["kernel"] == "stream"      # (the limit is back: one block)
    one.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sliding_kernel_takes_oversized_tiles_in_halves(oracle, dtype):
    """Rows of 1 ... 27 entries: 2 % of the 64-row tiles hold more than the strip's 1024 entries.  The sliding kernel
    computes those in two passes of 32 rows (bit-identical, like every streamed row) instead of leaving them to the
    overflow kernel (tree sums); tiles with a very long row still go there."""
    n = 300_000 + 41                      # (the last tile is a partial one)
    rp, ci, va = synth.ragged_csr(n, n, 4096, 12, dtype=dtype)
    x = synth.vector(n, dtype=dtype)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    bits = np.uint64 if dtype == np.float64 else np.uint32
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    d = dev.describe()
    assert d["slide"] == 1 and d["split_tiles"] > 20 and d["overflow_tiles"] == 0, d
    assert np.array_equal(dev.spmv(x).view(bits), y_ref.view(bits))           # every row, bit for bit
    dev.set_option("split_tiles", 0)
    d0 = dev.describe()
    assert d0["split_tiles"] == 0 and d0["overflow_tiles"] == d["split_tiles"], d0
    bound = oracle.csr_abs_bound(rp, ci, va, x)
    assert_spmv_close(dev.spmv(x), y_ref, bound, 1e-10 if dtype == np.float64 else 1e-4)
    dev.set_option("split_tiles", 1)
    # x not 16-byte aligned: the launch falls back to one super-tile per workgroup, and with it to the full overflow list
    import torch
    xpad = torch.zeros(n + 2, dtype=torch.float64 if dtype == np.float64 else torch.float32, device="cuda")
    off = 1 if dtype == np.float64 else 1        # 8 / 4 bytes past a 16-byte boundary
    xpad[off:off + n] = torch.from_numpy(x).cuda()
    assert xpad[off:].data_ptr() % 16 != 0
    y_un = dev.spmv_torch(xpad[off:off + n]).cpu().numpy()
    assert_spmv_close(y_un, y_ref, bound, 1e-10 if dtype == np.float64 else 1e-4)
    dev.set_option("slide_on", 0)         # one super-tile per workgroup: the overflow kernel takes all of them
    assert dev.describe()["overflow_tiles"] == d["split_tiles"]
    assert_spmv_close(dev.spmv(x), y_ref, bound, 1e-10 if dtype == np.float64 else 1e-4)
    dev.close()
    # a few rows of 200 ... 900 entries on top: their tiles cannot be halved (a lane per row would crawl)
    rng = np.random.default_rng(3)
    lens = np.diff(rp.astype(np.int64))
    heavy = np.sort(rng.choice(n, 40, replace=False))
    cols = [ci[int(rp[r]):int(rp[r + 1])] for r in range(n)] if n < 1 else None
    new_cols = {}
    for r in heavy:
        lo, hi = max(0, int(r) - 2000), min(n, int(r) + 2000)
        new_cols[int(r)] = np.unique(np.concatenate([ci[int(rp[r]):int(rp[r + 1])],
                                                     rng.integers(lo, hi, int(rng.integers(200, 900))).astype(np.uint64)]))
        lens[r] = new_cols[int(r)].size
    rp2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ci2 = np.empty(int(rp2[-1]), np.uint64)
    prev = 0
    for r in heavy:
        a0, a1 = int(rp[prev]), int(rp[r])
        ci2[int(rp2[prev]):int(rp2[prev]) + (a1 - a0)] = ci[a0:a1]
        ci2[int(rp2[r]):int(rp2[r + 1])] = new_cols[int(r)]
        prev = int(r) + 1
    ci2[int(rp2[prev]):] = ci[int(rp[prev]):]
    va2 = rng.uniform(-1, 1, int(rp2[-1])).astype(dtype)
    dev = sp.CsrMatrix(n, n, rp2, ci2, va2).device()
    d = dev.describe()
    assert d["slide"] == 1 and d["split_tiles"] > 20 and 0 < d["overflow_tiles"] <= 40, d
    y_ref = oracle.csr_spmv(rp2, ci2, va2, x)
    bound = oracle.csr_abs_bound(rp2, ci2, va2, x)
    y = dev.spmv(x)
    assert_spmv_close(y, y_ref, bound, 1e-10 if dtype == np.float64 else 1e-4)
    differ = np.flatnonzero(y.view(bits) != y_ref.view(bits))
    tiles = set((differ // 64).tolist())
    assert tiles <= set((heavy // 64).tolist())     # only rows of the tiles with a long row may differ (tree sums)
    dev.close()


def test_handle_owned_vectors(oracle):
    """spal_csr_alloc_vectors: x / y owned by the handle (for matrices of 256 MB and more the block is the fastest of a
    walk over the device's memory; small ones take the first block), usable as any device vectors, stable across calls."""
    import torch
    n = 200_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 3)
    dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    x, y = dev.vectors_torch()
    assert x.numel() == n and y.numel() == n and x.is_cuda and x.dtype == torch.float64
    x2, y2 = dev.vectors_torch()
    assert x2.data_ptr() == x.data_ptr() and y2.data_ptr() == y.data_ptr()
    xh = synth.vector(n)
    x.copy_(torch.from_numpy(xh))
    y.fill_(float("nan"))
    dev.spmv_torch(x, out=y)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), oracle.csr_spmv(rp, ci, va, xh))
    d = dev.describe()
    assert d["vectors_walk_blocks"] == 1          # 2.8 MB of matrix: nothing to place
    # a matrix large enough for the walk (330 MB): several blocks probed, the fastest kept
    n = 2_000_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 4)
    dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    dev.set_option("walk_blocks", 4)
    before = dev.describe()["placement_blocks"]
    x, y = dev.vectors_torch()
    d = dev.describe()
    # the first handle of the process that asks walks (4 new blocks here) and keeps one or two of them for everybody; if an
    # earlier test of this process walked already, this one only probes the kept blocks
    if before == 0:
        assert d["vectors_walk_blocks"] == 4 and d["vectors_probes"] == 4, d
    else:
        assert d["vectors_walk_blocks"] == 0 and 1 <= d["vectors_probes"] <= 2, d
    assert 1 <= d["placement_blocks"] <= 2 and 0 < d["vectors_walk_us"][0] <= d["vectors_walk_us"][1], d
    xh = synth.vector(n)
    x.copy_(torch.from_numpy(xh))
    dev.spmv_torch(x, out=y)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), oracle.csr_spmv(rp, ci, va, xh))
    # a SECOND handle of the process: no new block, at most two probes, its vectors a piece of a kept block -- and the
    # first handle's piece goes back to the block when that handle is destroyed
    free_with_both = None
    dev2 = sp.CsrMatrix._trusted(n, n, rp, ci, va).device_copy()
    x2, y2 = dev2.vectors_torch()
    d2 = dev2.describe()
    assert d2["vectors_walk_blocks"] == 0 and 1 <= d2["vectors_probes"] <= 2 and d2["placement_blocks"] == d["placement_blocks"], d2
    assert x2.data_ptr() != x.data_ptr()
    x2.copy_(torch.from_numpy(xh))
    dev2.spmv_torch(x2, out=y2)
    torch.cuda.synchronize()
    assert np.array_equal(y2.cpu().numpy(), oracle.csr_spmv(rp, ci, va, xh))
    free_with_both = d2["placement_free_bytes"]
    del x2, y2
    dev2.close()
    assert dev.describe()["placement_free_bytes"] >= free_with_both + 2 * n * 8
    # cache_trim gives back the kept blocks nobody holds a piece of; the block under this handle's vectors stays
    sp.cache_trim()
    assert dev.describe()["placement_blocks"] == 1
    y.fill_(float("nan"))
    dev.spmv_torch(x, out=y)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), oracle.csr_spmv(rp, ci, va, xh))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_block_window_kernel_on_power_law_rows(oracle, dtype):
    """Skewed row lengths with columns near the rows (round 4, late): entries streamed in passes, one LDS window of x per row
    block, rows of at most 32 entries summed by one thread in the reference's order (bit for bit, also across a pass boundary),
    longer rows by a wave (1e-10 / 1e-4).  Forced by name, then chosen -- or not -- by time against the row split."""
    rng = np.random.default_rng(12)
    n = 300_001                                     # (the last block is a partial one)
    lens = np.minimum((rng.pareto(1.6, n) * 6 + 1).astype(np.int64), 5000)
    lens[:6] = (0, 9000, 33, 32, 0, 1)             # a row across three passes, the two sides of the thread / wave limit, empty rows
    lens[700:1300] = 0                              # more than a block's unit of empty rows in a run
    lens[-3:] = (0, 40, 0)
    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    cols = np.clip(rows - 5000 + rng.integers(0, 10000, rows.size), 0, n - 1)
    key = np.unique(rows * n + cols)
    r2, c2 = key // n, key % n
    rp = np.concatenate([[0], np.cumsum(np.bincount(r2, minlength=n))]).astype(np.uint64)
    ci, va = c2.astype(np.uint64), rng.uniform(-1, 1, c2.size).astype(dtype)
    x = rng.uniform(-1, 1, n).astype(dtype)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    bound = oracle.csr_abs_bound(rp, ci, va.astype(np.float64), x.astype(np.float64))
    tol = 1e-10 if dtype == np.float64 else 1e-4
    bits = np.uint64 if dtype == np.float64 else np.uint32
    rl = np.diff(rp.astype(np.int64))
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    dev.set_option("blockwin", 1)
    d = dev.describe()
    assert d["kernel"] == "blockwin" and d["block_rows"] in (512, 1024, 2048, 4096) and d["window_columns"] % 256 == 0, d
    assert d["blocks"] == -(-n // d["block_rows"]), d
    y = dev.spmv(x)
    assert_spmv_close(y, y_ref, bound, tol)
    thread_rows = rl <= 32
    assert np.array_equal(y[thread_rows].view(bits), y_ref[thread_rows].view(bits))    # the reference's order of additions
    assert np.all(y[rl == 0] == 0)
    # two streams at once: the kernel keeps nothing on the handle
    import torch
    xt = torch.from_numpy(x).cuda()
    ys = [torch.full((n,), float("nan"), dtype=xt.dtype, device="cuda") for _ in range(2)]
    st = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    for _ in range(3):
        for k in range(2):
            dev.spmv_dev(xt.data_ptr(), ys[k].data_ptr(), st[k])
    torch.cuda.synchronize()
    for k in range(2):
        assert np.array_equal(ys[k].cpu().numpy().view(bits), y.view(bits))
    # an x that is only 8-byte aligned takes the scalar staging loop
    if dtype == np.float64:
        xo = torch.empty(n + 1, dtype=torch.float64, device="cuda")[1:]
        xo.copy_(xt)
        yo = torch.empty(n, dtype=torch.float64, device="cuda")
        dev.spmv_dev(xo.data_ptr(), yo.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(yo.cpu().numpy().view(bits), y.view(bits))
    # auto: timed against the row split at setup, the faster of the two stays and says what both took
    dev.set_option("blockwin", -1)
    da = dev.describe()
    assert da["kernel"] in ("blockwin", "split"), da
    if da["kernel"] == "blockwin":
        assert 0 < da["setup_us"][1] <= da["setup_us"][0], da
    assert_spmv_close(dev.spmv(x), y_ref, bound, tol)
    dev.set_option("blockwin", 0)
    assert dev.describe()["kernel"] == "split"
    # columns anywhere: no window fits LDS -- the same passes and row sums with x gathered from memory (window_columns 0)
    lens2 = np.minimum((rng.pareto(1.6, 6000) * 6 + 1).astype(np.int64), 3000)
    rows2 = np.repeat(np.arange(6000, dtype=np.int64), lens2)
    key2 = np.unique(rows2 * (n + 8) + rng.integers(0, n + 8, rows2.size))
    r3, c3 = key2 // (n + 8), key2 % (n + 8)
    rp2 = np.concatenate([[0], np.cumsum(np.bincount(r3, minlength=6000))]).astype(np.uint64)
    va2 = rng.uniform(-1, 1, c3.size).astype(dtype)
    x2 = rng.uniform(-1, 1, n + 8).astype(dtype)
    far = sp.CsrMatrix(6000, n + 8, rp2, c3.astype(np.uint64), va2).device()
    far.set_option("blockwin", 1)
    df = far.describe()
    assert df["kernel"] == "blockwin" and df["window_columns"] == 0, df
    y2 = far.spmv(x2)
    y2_ref = oracle.csr_spmv(rp2, c3.astype(np.uint64), va2, x2)
    assert_spmv_close(y2, y2_ref, oracle.csr_abs_bound(rp2, c3.astype(np.uint64), va2.astype(np.float64), x2.astype(np.float64)), tol)
    rl2 = np.diff(rp2.astype(np.int64))
    assert np.array_equal(y2[rl2 <= 32].view(bits), y2_ref[rl2 <= 32].view(bits))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_block_window_kernel_edge_shapes(oracle, dtype):
    """The block-window kernel on shapes its passes and walks could trip over: a row that spans many passes among rows of one to
    three entries, a dense row in a matrix of an odd number of columns (the window ends inside a 16-byte vector), blocks without
    entries, a last block of a few rows, x 16-byte aligned and not."""
    import torch
    rng = np.random.default_rng(21)
    tol = 1e-10 if dtype == np.float64 else 1e-4
    bits = np.uint64 if dtype == np.float64 else np.uint32
    cases = []
    # (a) 5000 rows, one of 6000 entries (two passes), the rest 1 ... 3 entries near it; 50 001 columns (ten per row: a block of
    # 512 rows spans 11 000)
    lens = rng.integers(1, 4, 5000)
    lens[2500] = 6000
    cases.append((5000, 50_001, lens, 6000))
    # (a') the same with a row of 40 000 entries (14 passes): wider than any window -- the window-less form
    lens = rng.integers(1, 4, 5000)
    lens[2500] = 40_000
    cases.append((5000, 50_001, lens, 40_000))
    # (b) 700 rows x 701 columns: ten empty rows, a dense row, short rows; the second block holds 188 rows
    lens = rng.integers(0, 6, 700)
    lens[:10] = 0
    lens[10] = 701
    cases.append((700, 701, lens, 701))
    # (c) 3000 rows of which only the last 100 hold entries (five blocks of 512 without any)
    lens = np.zeros(3000, np.int64)
    lens[-100:] = rng.integers(1, 200, 100)
    cases.append((3000, 4099, lens, 600))
    for n, nc, lens, spread in cases:
        ci = []
        for r, k in enumerate(lens):
            k = int(min(k, nc))
            lo = max(0, min(nc - min(spread, nc), int(r * nc / n) - spread // 2))
            ci.append(lo + np.sort(rng.choice(min(spread, nc - lo), k, replace=False)))
        lens2 = np.array([c.size for c in ci])
        rp = np.concatenate([[0], np.cumsum(lens2)]).astype(np.uint64)
        ci = np.concatenate(ci).astype(np.uint64)
        va = rng.uniform(-1, 1, ci.size).astype(dtype)
        x = rng.uniform(-1, 1, nc).astype(dtype)
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        bound = oracle.csr_abs_bound(rp, ci, va.astype(np.float64), x.astype(np.float64))
        dev = sp.CsrMatrix(n, nc, rp, ci, va).device()
        dev.set_option("blockwin", 1)
        d = dev.describe()
        assert d["kernel"] == "blockwin" and (d["window_columns"] > 0) == (spread <= 6000), d
        y = dev.spmv(x)
        assert_spmv_close(y, y_ref, bound, tol)
        assert np.array_equal(y[lens2 <= 32].view(bits), y_ref[lens2 <= 32].view(bits)), d
        # the same through the device entry point with x at an address that is not a multiple of 16
        xo = torch.empty(nc + 3, dtype=torch.from_numpy(x).dtype, device="cuda")[(1 if dtype == np.float64 else 3):][:nc]
        assert xo.data_ptr() % 16 != 0
        xo.copy_(torch.from_numpy(x))
        yo = torch.full((n,), float("nan"), dtype=xo.dtype, device="cuda")
        dev.spmv_dev(xo.data_ptr(), yo.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(yo.cpu().numpy().view(bits), y.view(bits))


def test_device_copy_is_a_handle_of_its_own(oracle):
    """`device()` caches ONE handle per matrix and device; `device_copy()` uploads again -- what a benchmark rotates its
    launches over, so that a matrix below the Infinity Cache's 256 MB is not served from it (bench.py, configs 2 and 4)."""
    n = 50_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 3)
    a = sp.CsrMatrix(n, n, rp, ci, va)
    x = synth.vector(n)
    d0, d1, d2 = a.device(), a.device(), a.device_copy()
    assert d0 is d1 and d2 is not d0
    assert d2.describe()["addr"] != d0.describe()["addr"]          # (its arrays lie elsewhere)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    assert np.array_equal(d0.spmv(x), y_ref) and np.array_equal(d2.spmv(x), y_ref)
    d2.close()
    assert np.array_equal(a.device().spmv(x), y_ref)               # the cached handle is untouched


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_row_split_on_power_law_rows(oracle, dtype):
    """Skewed row lengths (round 4): 0.7 % of the rows are longer than a lane may sum, and a third of the
    64-row tiles hold one -- the planner multiplies the long rows apart from the rest (A = A_short + A_long).  The short rows
    go through a plan of their own and keep the reference's bits; the long rows are tree sums (1e-10 / 1e-4).  Switched off
    the handle runs as rounds 1-3 did, and agrees."""
    rng = np.random.default_rng(11)
    n = 400_000
    lens = np.minimum((rng.pareto(1.6, n) * 6 + 1).astype(np.int64), 5000)
    lens[:3] = (0, 5000, 129)
    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    cols = np.clip(rows - 5000 + rng.integers(0, 10000, rows.size), 0, n - 1)
    key = np.unique(rows * n + cols)
    r2, c2 = key // n, key % n
    rp = np.concatenate([[0], np.cumsum(np.bincount(r2, minlength=n))]).astype(np.uint64)
    ci, va = c2.astype(np.uint64), rng.uniform(-1, 1, c2.size).astype(dtype)
    x = rng.uniform(-1, 1, n).astype(dtype)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    bound = oracle.csr_abs_bound(rp, ci, va.astype(np.float64), x.astype(np.float64))
    tol = 1e-10 if dtype == np.float64 else 1e-4
    dev = sp.CsrMatrix(n, n, rp, ci, va).device()
    dev.set_option("blockwin", 0)          # (the block-window kernel may beat the split at setup: its own test below)
    d = dev.describe()
    rl = np.diff(rp.astype(np.int64))
    assert d["kernel"] == "split" and d["split_threshold"] == 128 and d["split_long_rows"] == int((rl > 128).sum()), d
    # (the short part: the stream kernels with column panels, or -- timed against them at setup -- the column-blocked kernels)
    assert d["split_long_entries"] == int(rl[rl > 128].sum()) and d["short_part"]["kernel"] in ("stream", "cblock"), d
    y = dev.spmv(x)
    assert_spmv_close(y, y_ref, bound, tol)
    short = rl <= 128
    bits = np.uint64 if dtype == np.float64 else np.uint32
    if d["short_part"]["kernel"] == "cblock" or (d["short_part"]["stream_row_fraction"] == 1.0 and d["short_part"]["overflow_tiles"] == 0
                                                and d["short_part"]["panel_tiles"] == 0):
        assert np.array_equal(y[short].view(bits), y_ref[short].view(bits))       # every short row: the reference's order
    # the device entry point on two streams at once (no temporaries in the split: concurrent products stay safe)
    import torch
    xt = torch.from_numpy(x).cuda()
    ys = [torch.full((n,), float("nan"), dtype=xt.dtype, device="cuda") for _ in range(2)]
    st = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    for _ in range(3):
        for k in range(2):
            dev.spmv_dev(xt.data_ptr(), ys[k].data_ptr(), st[k])
    torch.cuda.synchronize()
    for k in range(2):
        assert np.array_equal(ys[k].cpu().numpy().view(bits), y.view(bits))
    dev.set_option("row_split", 0)
    d0 = dev.describe()
    assert d0["kernel"] in ("stream", "cblock") and d0["kernel"] != "split", d0
    assert_spmv_close(dev.spmv(x), y_ref, bound, tol)
    dev.set_option("row_split", -1)
    assert dev.describe()["kernel"] == "split"
