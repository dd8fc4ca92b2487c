"""GPU parity: the column-blocked CSR kernels (csr_cblock.hpp) -- matrices whose columns are anywhere.

Both forms (entry-parallel: the thread that heads a row's run in a tile adds the run to the row's running sum in LDS;
rows form: a thread owns rows, their running sums in its registers) add a row's products in ascending column order, one
after the other, from the first column block to the last (reference order: src/csr/ops/mul.rs:31-38), so every row must
equal the oracle's bit for bit (f64 and f32), whatever the tile geometry."""
import numpy as np
import pytest

import spalinalg_amd as sp
import spal_synth as synth

pytestmark = pytest.mark.gpu


def bits(a):
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def random_rows(rng, nrows, ncols, lens, dtype):
    """CSR with the given row lengths, columns uniform over [0, ncols) (sorted, distinct inside a row)."""
    lens = np.minimum(np.asarray(lens, dtype=np.int64), ncols)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    rows = np.repeat(np.arange(nrows, dtype=np.int64), lens)
    cols = rng.integers(0, ncols, rows.size)
    key = np.unique(rows * ncols + cols)              # sorted (row, col), duplicates dropped
    rows, cols = key // ncols, key % ncols
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=nrows))]).astype(np.uint64)
    va = rng.uniform(-1, 1, cols.size).astype(dtype)
    return rp, cols.astype(np.uint64), va


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("form, rows", [(0, 300), (0, 512), (0, 1000), (0, 2048), (0, 3333), (0, 4096), (0, 6001), (0, 8192),
                                        (1, 256), (1, 512), (1, 1024), (1, 2048), (1, 4096)])
def test_forced_geometries_bit_identical(oracle, dtype, form, rows):
    """Row-block heights from 300 to 4096 rows (multiples of the workgroup size or not), small column blocks (many tiles, empty tiles, tiles of one entry),
    rows of 0 ... 40 entries, a row count that is no multiple of anything."""
    rng = np.random.default_rng(100 + rows)
    nrows, ncols = 70_001, 50_000
    lens = rng.integers(0, 41, nrows)
    lens[rng.random(nrows) < 0.2] = 0                  # empty rows
    lens[1000:1300] = 0                                # a run of empty rows
    rp, ci, va = random_rows(rng, nrows, ncols, lens, dtype)
    x = rng.uniform(-1, 1, ncols).astype(dtype)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    dev = sp.CsrMatrix(nrows, ncols, rp, ci, va).device()
    ran = 0
    for shift in (10, 13, 16):                         # 49, 7 and 1 column blocks
        dev.set_option("cblock_shift", shift)
        dev.set_option("cblock_rows", rows)
        dev.set_option("cblock_form", form)
        dev.set_option("cblock", 1)
        d = dev.describe()
        if d["kernel"] != "cblock":                   # (a tile above the strip's 4096 entries: this height does not qualify)
            continue
        ran += 1
        assert d["cblock_rows"] == rows and d["cblock_cols"] == 1 << shift and d["cblock_form"] == ("rows" if form else "entry"), d
        y = dev.spmv(x)
        assert np.array_equal(bits(y), bits(y_ref)), (shift, d)
    assert ran >= 1


def test_first_product_is_assigned_and_signed_zeros(oracle):
    """The first product of a row is taken as it is (mul.rs:34): a row whose only product is -0.0 gives -0.0, a row of
    (+0.0) + (-0.0) gives +0.0, empty rows give +0.0 -- compared on the bit patterns."""
    nrows, ncols = 3000, 9000
    rng = np.random.default_rng(3)
    rp, ci, va = random_rows(rng, nrows, ncols, rng.integers(0, 6, nrows), np.float64)
    x = rng.uniform(-1, 1, ncols)
    x[::3] = 0.0
    x[1::7] = -0.0
    va[::5] = -va[::5]
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    assert np.any(bits(y_ref) == bits(np.array([-0.0]))[0])          # the case is in the data
    dev = sp.CsrMatrix(nrows, ncols, rp, ci, va).device()
    dev.set_option("cblock_shift", 11)
    dev.set_option("cblock", 1)
    for form in (0, 1):
        dev.set_option("cblock_form", form)
        d = dev.describe()
        assert d["kernel"] == "cblock" and d["cblock_form"] == ("rows" if form else "entry"), d
        assert np.array_equal(bits(dev.spmv(x)), bits(y_ref))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_uniform_columns_take_the_column_blocked_kernel(oracle, dtype):
    """SURVEY 8d's stress row (config 2 with uniform columns, 1M x 1M, 14 per row): the plan finds the rows gathering x
    from beyond L2 and builds the tiled copy by itself; the product equals the oracle's bit for bit and agrees with
    the stream kernels' (same order of additions)."""
    import torch
    n = 1_000_000 if dtype == np.float64 else 2_000_000      # x of 8 MB either way (4 MB would sit in an XCD's L2)
    rp, ci, va = synth.banded_csr(n, n, 14, n, synth.matrix_seed(2), dtype=dtype)
    x = synth.vector(n, dtype=dtype)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
    d = dev.describe()
    assert d["kernel"] == "cblock" and d["nonlocal_row_fraction"] > 0.9, d
    assert d["cblock_form"] == "rows" and d["cblock_run"] > 1.6, d     # 14 per row over 4 column blocks of 2 MB: runs of 3.6 entries
    y = dev.spmv(x)
    assert np.array_equal(bits(y), bits(y_ref))
    xt = torch.from_numpy(x).cuda()
    yt = torch.empty(n, dtype=xt.dtype, device="cuda")
    d = dev.autotune(xt, yt, iters=10)               # times the two families, keeps the faster: identical results
    assert d["cblock_us"][0] > 0 and d["cblock_us"][1] > 0, d
    dev.spmv_torch(xt, yt)
    torch.cuda.synchronize()
    assert np.array_equal(bits(yt.cpu().numpy()), bits(y_ref))
    dev.set_option("cblock", 0)
    assert dev.describe()["kernel"] == "stream"
    assert np.array_equal(bits(dev.spmv(x)), bits(y_ref))


def test_banded_matrices_do_not_take_it(oracle):
    n = 300_000
    rp, ci, va = synth.banded_csr(n, n, 14, 4096, 5)
    d = sp.CsrMatrix._trusted(n, n, rp, ci, va).device().describe()
    assert d["kernel"] == "stream" and d["cblock"] == 0 and d["nonlocal_row_fraction"] < 0.1, d


def test_a_crowded_column_block_disqualifies(oracle):
    """more than 255 entries of one row inside one column block: the counts are bytes -- the stream kernels run"""
    rng = np.random.default_rng(8)
    nrows, ncols = 20_000, 3_000_000
    lens = rng.integers(0, 12, nrows)
    rp, ci, va = random_rows(rng, nrows, ncols, lens, np.float64)
    # row 77: 400 entries packed into the first 1000 columns
    r0, r1 = int(rp[77]), int(rp[78])
    extra = np.sort(rng.choice(1000, 400, replace=False)).astype(np.uint64)
    ci = np.concatenate([ci[:r0], extra, ci[r1:]])
    va = np.concatenate([va[:r0], rng.uniform(-1, 1, 400), va[r1:]])
    rp = rp.copy()
    rp[78:] += np.uint64(400 - (r1 - r0))
    x = rng.uniform(-1, 1, ncols)
    dev = sp.CsrMatrix(nrows, ncols, rp, ci, va).device()
    dev.set_option("cblock", 1)
    assert dev.describe()["kernel"] != "cblock"
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    bound = oracle.csr_abs_bound(rp, ci, va, x)
    assert np.all(np.abs(dev.spmv(x) - y_ref) <= 1e-10 * bound + 1e-300)


def test_form_follows_the_entries_per_run(oracle):
    """10 entries per row over 20 column blocks (the shape of the matrix config 5 assembles, at a quarter of its size
    with column blocks a quarter as wide): runs of about one entry -> the entry-parallel form; the same rows with their
    columns inside a window of 2^16: runs of several entries -> the rows form.  Bit-identical either way."""
    n = 1_250_000
    for window, form in ((n, "entry"), (1 << 16, "rows")):
        rp, ci, va = synth.banded_csr(n, n, 10, window, 7)
        x = synth.vector(n)
        dev = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
        dev.set_option("cblock_shift", 16)
        dev.set_option("cblock", 1)
        d = dev.describe()
        assert d["kernel"] == "cblock" and d["cblock_form"] == form, d
        assert np.array_equal(bits(dev.spmv(x)), bits(oracle.csr_spmv(rp, ci, va, x)))


def test_first_products_from_two_threads_on_a_device_assembled_handle(oracle):
    """a handle assembled on the device builds its column-blocked copy with its FIRST product, and
    spal_csr_spmv_dev may be called concurrently on one handle.  Two threads issue the first product together, each on
    its own stream: the build happens once, under the handle's lock, both wait for it, both results carry the oracle's
    bits."""
    import threading
    import torch
    n, per_row = 1_000_000, 10
    r, c, v = synth.coo(n, n, n * per_row, synth.matrix_seed(5), 10, 1)
    x = synth.vector(n)
    p, i, w = oracle.coo_to_csr(n, n, r, c, v)
    y_ref = oracle.csr_spmv(p, i, w, x)
    for attempt in range(3):
        dev = sp.CooMatrix.with_triplets(n, n, r, c, v).upload().assemble_csr()
        d = dev.describe()
        assert d["cblock_pending"] == 1 and d["cblock"] == 0, d
        xt = torch.from_numpy(x).cuda()
        ys = [torch.full((n,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2)]
        streams = [torch.cuda.Stream() for _ in range(2)]
        torch.cuda.synchronize()
        gate, errors = threading.Barrier(2), []

        def work(k):
            try:
                gate.wait()
                for _ in range(3):
                    dev.spmv_dev(xt.data_ptr(), ys[k].data_ptr(), streams[k])
                streams[k].synchronize()
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)

        threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        d = dev.describe()
        assert d["cblock_pending"] == 0 and d["cblock"] == 1 and d["cblock_failed"] == 0, d
        for k in range(2):
            assert np.array_equal(bits(ys[k].cpu().numpy()), bits(y_ref))
