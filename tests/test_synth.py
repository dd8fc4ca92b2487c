"""The bench support library's (spal_synth) synthetic-input generators against an independent pure-Python
statement of SURVEY.md section 8d (SplitMix64)."""
import numpy as np

import spalinalg_amd as sp
import spal_synth as synth

M64 = (1 << 64) - 1
GAMMA = 0x9E3779B97F4A7C15
ROW_MULT = 0xD1B54A32D192ED03


class SplitMix:
    def __init__(self, seed):
        self.s = seed & M64

    def next(self):
        self.s = (self.s + GAMMA) & M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    def value(self):
        return 2.0 * ((self.next() >> 11) * 2.0 ** -53) - 1.0


def py_banded(nrows, ncols, per_row, window, seed):
    rp, ci, va = [0], [], []
    for row in range(nrows):
        rng = SplitMix(seed ^ ((ROW_MULT * (row + 1)) & M64))
        centre = row * ncols // nrows
        w0 = max(0, centre - window // 2)
        w0 = min(w0, ncols - window)
        cols = []
        while len(cols) < per_row:
            c = w0 + rng.next() % window
            if c not in cols:
                cols.append(c)
        cols.sort()
        ci += cols
        va += [rng.value() for _ in range(per_row)]
        rp.append(len(ci))
    return rp, ci, va


def test_banded_generator_bit_exact():
    for nrows, ncols, per_row, window in [(50, 50, 14, 32), (64, 64, 3, 64), (40, 100, 5, 20), (100, 40, 5, 20)]:
        rp, ci, va = synth.banded_csr(nrows, ncols, per_row, window, synth.matrix_seed(3))
        prp, pci, pva = py_banded(nrows, ncols, per_row, window, synth.matrix_seed(3))
        assert rp.tolist() == prp and ci.tolist() == pci and va.tolist() == pva
        # it is a valid CsrMatrix with exactly per_row entries in every row
        a = sp.CsrMatrix(nrows, ncols, rp, ci, va)
        assert a.nnz() == nrows * per_row
        assert np.all(va >= -1) and np.all(va < 1)


def py_ragged(nrows, ncols, window, seed):
    """SURVEY.md section 8d robustness variant: the row's first draw is its length, 1 + next() % 27."""
    rp, ci, va = [0], [], []
    for row in range(nrows):
        rng = SplitMix(seed ^ ((ROW_MULT * (row + 1)) & M64))
        k = 1 + rng.next() % 27
        centre = row * ncols // nrows
        w0 = min(max(0, centre - window // 2), ncols - window)
        cols = []
        while len(cols) < k:
            c = w0 + rng.next() % window
            if c not in cols:
                cols.append(c)
        cols.sort()
        ci += cols
        va += [rng.value() for _ in range(k)]
        rp.append(len(ci))
    return rp, ci, va


def test_ragged_generator_bit_exact_and_slices():
    for nrows, ncols, window in [(60, 60, 40), (40, 100, 27), (300, 300, 300)]:
        rp, ci, va = synth.ragged_csr(nrows, ncols, window, synth.matrix_seed(3))
        prp, pci, pva = py_ragged(nrows, ncols, window, synth.matrix_seed(3))
        assert rp.tolist() == prp and ci.tolist() == pci and va.tolist() == pva
        sp.CsrMatrix(nrows, ncols, rp, ci, va)   # a valid CsrMatrix (CsrMatrix::new's assertions)
    n = 20000
    rp, ci, va = synth.ragged_csr(n, n, 4096, 5)
    lens = np.diff(rp.astype(np.int64))
    assert lens.min() == 1 and lens.max() == 27 and abs(lens.mean() - 14.0) < 0.2
    srp, sci, sva = synth.ragged_csr(n, n, 4096, 5, rows=(777, 9000))
    a, b = int(rp[777]), int(rp[9000])
    assert np.array_equal(srp, rp[777:9001] - rp[777]) and np.array_equal(sci, ci[a:b]) and np.array_equal(sva, va[a:b])
    f32 = synth.ragged_csr(n, n, 4096, 5, dtype=np.float32)
    assert np.array_equal(f32[0], rp) and np.array_equal(f32[2], va.astype(np.float32))


def test_banded_window_bounds_and_slices():
    n, w = 5000, 256
    rp, ci, va = synth.banded_csr(n, n, 14, w, 1)
    rows = np.repeat(np.arange(n), 14)
    w0 = np.clip(rows - w // 2, 0, n - w)
    assert np.all(ci >= w0) and np.all(ci < w0 + w)
    srp, sci, sva = synth.banded_csr(n, n, 14, w, 1, rows=(1234, 2345))
    assert np.array_equal(sci, ci[1234 * 14:2345 * 14]) and np.array_equal(sva, va[1234 * 14:2345 * 14])
    f32 = synth.banded_csr(n, n, 14, w, 1, dtype=np.float32)
    assert np.array_equal(f32[2], va.astype(np.float32)) and np.array_equal(f32[1], ci)


def test_vector_generator_bit_exact():
    x = synth.vector(1000)
    rng = SplitMix(synth.SEED_X)
    assert x.tolist() == [rng.value() for _ in range(1000)]


def test_coo_generator_bit_exact_and_injection():
    seed = synth.matrix_seed(1)
    r, c, v = synth.coo(100, 70, 500, seed)
    rng = SplitMix(seed)
    want = [(rng.next() % 100, rng.next() % 70, rng.value()) for _ in range(500)]
    assert list(zip(r.tolist(), c.tolist(), v.tolist())) == want
    # injection: deterministic, about the requested rates, copies refer to earlier entries
    n = 200_000
    r0, c0, v0 = synth.coo(5000, 5000, n, 9)
    r1, c1, v1 = synth.coo(5000, 5000, n, 9, 10, 1)
    r2, c2, v2 = synth.coo(5000, 5000, n, 9, 10, 1)
    assert np.array_equal(r1, r2) and np.array_equal(c1, c2) and np.array_equal(v1, v2)
    changed = (r0 != r1) | (c0 != c1) | (v0 != v1)
    assert 0.008 * n < changed.sum() < 0.014 * n
    neg = changed & (v0 != v1)
    assert 0.0005 * n < neg.sum() < 0.002 * n
    key0 = {(int(a), int(b)) for a, b in zip(r0, c0)}
    assert all((int(a), int(b)) in key0 for a, b in zip(r1[changed], c1[changed]))


def test_algorithmic_byte_counts():
    # config 2 / 3
    assert synth.spmv_bytes(14_000_000, 1_000_000, 1_000_000, 1_000_000, 8) == 188_000_004
    assert synth.spmv_bytes(140_000_000, 10_000_000, 10_000_000, 10_000_000, 8) == 1_880_000_004
    assert synth.spmv_flops(140_000_000) == 280_000_000
