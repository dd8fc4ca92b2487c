"""Shared helpers for the parity tests."""
import numpy as np


def random_csr(rng, nrows, ncols, density=None, row_len=None, dtype=np.float64, empty_rows=0.1):
    """Random valid CSR (strictly increasing columns, some empty rows)."""
    rowptr = [0]
    colind = []
    for r in range(nrows):
        if rng.random() < empty_rows:
            k = 0
        elif row_len is not None:
            k = int(row_len(rng))
        else:
            k = int(rng.binomial(ncols, density))
        k = min(k, ncols)
        cols = np.sort(rng.choice(ncols, size=k, replace=False)) if k else np.empty(0, dtype=np.int64)
        colind.append(cols)
        rowptr.append(rowptr[-1] + k)
    colind = np.concatenate(colind) if colind else np.empty(0)
    values = rng.uniform(-1, 1, size=colind.size).astype(dtype)
    return (np.asarray(rowptr, dtype=np.uint64), colind.astype(np.uint64), values)


def assert_spmv_close(y, y_ref, bound, tol):
    """Parity criterion of SURVEY.md section 8d: normwise inf-norm AND
    componentwise against sum_k |A_ik||x_k|."""
    y = np.asarray(y, dtype=np.float64)
    y_ref = np.asarray(y_ref, dtype=np.float64)
    assert y.shape == y_ref.shape
    assert np.array_equal(np.isnan(y), np.isnan(y_ref))
    fin = np.isfinite(y_ref)
    assert np.array_equal(y[~fin & ~np.isnan(y_ref)], y_ref[~fin & ~np.isnan(y_ref)])
    err = np.abs(y - y_ref)[fin]
    scale = np.max(np.abs(y_ref[fin])) if fin.any() else 0.0
    if scale > 0:
        assert err.max() <= tol * scale, f"normwise {err.max() / scale:.3e} > {tol}"
    b = np.asarray(bound, dtype=np.float64)[fin]
    slack = tol * b + 1e-300
    worst = np.max(err - slack)
    assert worst <= 0, f"componentwise violation {worst:.3e}"
