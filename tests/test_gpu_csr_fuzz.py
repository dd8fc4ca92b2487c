"""GPU parity, randomised: CSR y = A*x through the C ABI vs the CPU oracle over mixtures of row-length
distributions, column patterns and shapes that steer the planner through all of its paths (tile heights 64 ... 8,
skewed strips, skipped tiles + overflow kernel, LDS pages / x through L2, the vector kernel).

Tolerance as in test_gpu_csr_spmv.py: f64 1e-10, f32 1e-4, normwise and componentwise."""
import os

import numpy as np
import pytest

import spalinalg_amd as sp
from tests.util import assert_spmv_close

pytestmark = pytest.mark.gpu
TOL = {np.dtype(np.float64): 1e-10, np.dtype(np.float32): 1e-4}


def _lengths(rng, kind, n):
    if kind == "const":
        lens = np.full(n, int(rng.choice([1, 3, 7, 16, 24, 32, 64, 65, 96, 100, 128])), np.int64)
    elif kind == "uniform":
        lens = rng.integers(0, int(rng.choice([4, 20, 28, 70, 130, 300])), n)   # (28: 64-row tiles just above the strip: halves)
    elif kind == "pareto":
        lens = np.minimum((rng.pareto(1.5, n) * rng.choice([2, 6, 20]) + 1).astype(np.int64), 6000)
    elif kind == "two_regions":      # half the matrix short rows, half long ones
        lens = np.where(np.arange(n) < n // 2, rng.integers(0, 12, n), rng.integers(40, 110, n))
    else:
        raise AssertionError(kind)
    lens = lens.astype(np.int64)
    # stretches of empty rows, a few very heavy rows, heavy rows at both ends
    for _ in range(int(rng.integers(0, 4))):
        a = int(rng.integers(0, n))
        lens[a:a + int(rng.integers(1, 300))] = 0
    for _ in range(int(rng.integers(0, 6))):
        lens[int(rng.integers(0, n))] = int(rng.choice([129, 500, 1025, 3000, 9000]))
    if rng.random() < 0.5:
        lens[0] = int(rng.choice([0, 1, 2000]))
        lens[-1] = int(rng.choice([0, 1, 1500]))
    return lens


def _matrix(rng, n, ncols, lens, pattern, dtype):
    lens = np.minimum(lens, ncols)
    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    if pattern == "band":
        w = int(rng.choice([300, 3000, 20_000]))
        centre = rows * ncols // max(n, 1)
        cols = np.clip(centre - w // 2 + rng.integers(0, w, rows.size), 0, ncols - 1)
    elif pattern == "clusters":      # stencil-like: a few narrow clusters far apart
        offs = rng.integers(-ncols // 3, ncols // 3, 5)
        cols = np.clip(rows * ncols // max(n, 1) + offs[rng.integers(0, 5, rows.size)] + rng.integers(-40, 40, rows.size), 0, ncols - 1)
    else:                            # anywhere
        cols = rng.integers(0, ncols, rows.size)
    key = np.unique(rows * ncols + cols)          # sorted by (row, column), duplicates dropped
    rows2, cols2 = key // ncols, key % ncols
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows2, minlength=n))]).astype(np.uint64)
    return rp, cols2.astype(np.uint64), rng.uniform(-1, 1, cols2.size).astype(dtype)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SPAL_FUZZ_SEEDS", "24"))))   # (more seeds: a longer soak)
def test_random_matrices_all_planner_paths(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    dtype = np.float64 if seed % 3 else np.float32
    n = [20_011, 50_000, 1000, 120_000, 63][(seed // 2) % 5]
    ncols = [40_000, 300_007, 2049, n][(seed // 3) % 4]
    if seed == 22:
        n = 1
    if seed == 23:
        ncols = 1
    kind = ["const", "uniform", "pareto", "two_regions"][seed % 4]
    pattern = ["band", "clusters", "anywhere"][(seed // 4) % 3]
    rp, ci, va = _matrix(rng, n, ncols, _lengths(rng, kind, n), pattern, dtype)
    x = rng.uniform(-1, 1, ncols).astype(dtype)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    bound = oracle.csr_abs_bound(rp, ci, va.astype(np.float64), x.astype(np.float64))
    dev = sp.CsrMatrix(n, ncols, rp, ci, va).device()
    d = dev.describe()
    assert_spmv_close(dev.spmv(x), y_ref, bound, TOL[va.dtype])
    if d["kernel"] == "blockwin":       # skewed rows near the diagonal: the block-window kernel won against the split at setup
        assert d["setup_us"][1] <= d["setup_us"][0], d          # (printed to 0.1 us)
        dev.set_option("blockwin", 0)
        d = dev.describe()
        assert d["kernel"] == "split", d
        assert_spmv_close(dev.spmv(x), y_ref, bound, TOL[va.dtype])
    dev.set_option("blockwin", 0)       # (the paths below are the split's and the one-handle kernels')
    if d["kernel"] == "split":          # skewed rows: the planner split the long rows off (round 4); the rest of this test
        assert d["split_long_rows"] > 0 and d["short_part"]["nnz"] + d["split_long_entries"] == int(rp[-1]), d
        dev.set_option("row_split", 0)  # drives the one-handle paths
        d = dev.describe()
        assert d["kernel"] != "split"
        assert_spmv_close(dev.spmv(x), y_ref, bound, TOL[va.dtype])
    if d["kernel"] == "stream":
        for key, value in (("persistent", 1), ("skew", 1 - d["skew"]), ("rows_per_tile", 8), ("rows_per_tile", 128), ("rows_per_tile", 256), ("rows_per_tile", 12), ("stream_row_max", 16),
                           ("window_pages", 4)):
            dev.set_option(key, value)
            assert_spmv_close(dev.spmv(x), y_ref, bound, TOL[va.dtype])
    dev.set_option("kernel", 1)
    assert_spmv_close(dev.spmv(x), y_ref, bound, TOL[va.dtype])
    if int(rp[-1]):
        dev.set_option("kernel", 2)      # the stream kernel on whatever this is
        assert dev.describe()["kernel"] == "stream"
        assert_spmv_close(dev.spmv(x), y_ref, bound, TOL[va.dtype])
        # ... and the column-blocked kernel forced on whatever this is (random row-block height and column-block width):
        # where the matrix qualifies (no row with more than 255 entries in one column block, every tile inside the strip)
        # its rows must equal the oracle's BIT FOR BIT, heavy rows, empty stretches and all
        dev.set_option("cblock_rows", int(rng.choice([0, 256, 700, 1024, 3000, 4096, 5555, 8192])))
        dev.set_option("cblock_shift", int(rng.choice([0, 8, 11, 14, 17])))
        dev.set_option("cblock_form", int(rng.choice([-1, 0, 1])))      # (the rows form takes heights of 256 << k only: others fall to the stream kernels)
        dev.set_option("cblock", 1)
        if dev.describe()["kernel"] == "cblock":
            bits = np.uint64 if dtype == np.float64 else np.uint32
            assert np.array_equal(dev.spmv(x).view(bits), y_ref.view(bits)), dev.describe()
        else:
            assert_spmv_close(dev.spmv(x), y_ref, bound, TOL[va.dtype])
        # ... and the row split forced at a random threshold: rows above it by csr_spmv_row_list, the rest by a handle of their own
        dev.set_option("cblock", -1)
        dev.set_option("row_split_threshold", int(rng.choice([1, 4, 16, 128, 1000])))
        dev.set_option("row_split", 1)
        d = dev.describe()
        lens = np.diff(rp.astype(np.int64))
        if (lens > d.get("split_threshold", 1 << 30)).any() and d["kernel"] == "split":
            assert d["split_long_rows"] == int((lens > d["split_threshold"]).sum()), d
        assert_spmv_close(dev.spmv(x), y_ref, bound, TOL[va.dtype])
        # ... and the block-window kernel forced on whatever this is: where the windows fit LDS, rows of at most 32 entries keep
        # the reference's bits (one thread, left to right, across pass boundaries), longer rows are wave sums
        dev.set_option("row_split", -1)
        dev.set_option("blockwin", 1)
        d = dev.describe()
        yb = dev.spmv(x)
        assert_spmv_close(yb, y_ref, bound, TOL[va.dtype])
        if d["kernel"] == "blockwin":
            bits = np.uint64 if dtype == np.float64 else np.uint32
            assert np.array_equal(yb[lens <= 32].view(bits), y_ref[lens <= 32].view(bits)), d
