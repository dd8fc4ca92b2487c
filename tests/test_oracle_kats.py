"""Pins the CPU oracle against every known-answer vector the reference's own
tests hold for the hot path."""
import numpy as np
import pytest

U = np.uint64
F = np.float64


def test_g1_coo_to_csr(kats, oracle):
    g = kats["G1_coo_to_csr"]
    for dt in (np.float64, np.float32):
        p, i, v = oracle.coo_to_csr(g["nrows"], g["ncols"], g["rows"], g["cols"],
                                    np.array(g["vals"], dtype=dt))
        assert p.tolist() == g["rowptr"]
        assert i.tolist() == g["colind"]
        assert v.tolist() == g["values"] and v.dtype == dt


def test_g2_coo_to_csc(kats, oracle):
    g = kats["G2_coo_to_csc"]
    p, i, v = oracle.coo_to_csc(g["nrows"], g["ncols"], g["rows"], g["cols"],
                                np.array(g["vals"], dtype=F))
    assert p.tolist() == g["colptr"]
    assert i.tolist() == g["rowind"]
    assert v.tolist() == g["values"]


def test_g3_g4_csc_csr_conversions(kats, oracle):
    g = kats["G3_csc_to_csr"]
    p, i, v = oracle.transpose(g["ncols"], g["nrows"], g["colptr"], g["rowind"],
                               np.array(g["csc_values"], dtype=F))
    assert (p.tolist(), i.tolist(), v.tolist()) == (g["rowptr"], g["colind"], g["csr_values"])
    g = kats["G4_csr_to_csc"]
    p, i, v = oracle.transpose(g["nrows"], g["ncols"], g["rowptr"], g["colind"],
                               np.array(g["csr_values"], dtype=F))
    assert (p.tolist(), i.tolist(), v.tolist()) == (g["colptr"], g["rowind"], g["csc_values"])


def test_g5_csc_mul_literal_route(kats, oracle):
    g = kats["G5_csc_mul"]
    a, b, o = g["lhs"], g["rhs"], g["out"]
    p, i, v = oracle.csc_mul(
        (a["nrows"], a["ncols"]), (a["colptr"], a["rowind"], np.array(a["values"], dtype=F)),
        (b["nrows"], b["ncols"]), (b["colptr"], b["rowind"], np.array(b["values"], dtype=F)))
    assert p.tolist() == o["colptr"]
    assert i.tolist() == o["rowind"]
    assert v.tolist() == o["values"]


def test_g5_derived_spmv_vectors(kats, oracle):
    """Each rhs column is an x, each out column the y: both SpMV restatements
    (CSC direct, CSR after conversion) must reproduce them exactly."""
    g = kats["G5_csc_mul"]
    a = g["lhs"]
    vals = np.array(a["values"], dtype=F)
    rp, ci, rv = oracle.transpose(a["ncols"], a["nrows"], a["colptr"], a["rowind"], vals)
    for case in g["spmv"]:
        x = np.array(case["x"], dtype=F)
        y_csc = oracle.csc_spmv(a["nrows"], a["colptr"], a["rowind"], vals, x)
        y_csr = oracle.csr_spmv(rp, ci, rv, x)
        assert y_csc.tolist() == case["y"]
        assert y_csr.tolist() == case["y"]


def test_g5_columns_match_fixture_consistency(kats):
    """the derived (x, y) pairs really are the columns of rhs / out."""
    g = kats["G5_csc_mul"]
    for j, case in enumerate(g["spmv"]):
        for name, mat, vec in (("rhs", g["rhs"], case["x"]), ("out", g["out"], case["y"])):
            dense = [0.0] * mat["nrows"]
            for p in range(mat["colptr"][j], mat["colptr"][j + 1]):
                dense[mat["rowind"][p]] = mat["values"][p]
            assert dense == vec, (name, j)


def test_g6_transpose(kats, oracle):
    g = kats["G6_transpose"]["csr"]
    p, i, v = oracle.transpose(g["n"], g["n"], g["rowptr"], g["colind"], np.array(g["values"], dtype=F))
    assert (p.tolist(), i.tolist(), v.tolist()) == (g["t_rowptr"], g["t_colind"], g["t_values"])
    g = kats["G6_transpose"]["csc"]
    p, i, v = oracle.transpose(g["n"], g["n"], g["colptr"], g["rowind"], np.array(g["values"], dtype=F))
    assert (p.tolist(), i.tolist(), v.tolist()) == (g["t_colptr"], g["t_rowind"], g["t_values"])


@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_g7_rejections(kats, oracle, fmt):
    for case in kats["G7_rejections"][fmt]:
        rc = oracle.validate(case["nrows"], case["ncols"], case["ptr"], case["ind"],
                             case["nvalues"], csr=(fmt == "csr"))
        assert rc != 0, case["name"]


def test_g7_expected_reason_codes(kats, oracle):
    """the first failing assertion is the one the test's name says."""
    want = {"new_invalid_nrows": 1, "new_invalid_ncols": 2,
            "new_invalid_colptr_first_not_zero": 4, "new_invalid_colptr_invalid_length": 3,
            "new_invalid_rowind": 8, "new_unsorted_colind": 8, "new_unsorted_rowind": 8,
            "new_invalid_rowind_values": 6}
    # new_unsorted_*: ind [1, 0] with one minor index -> 1 is out of range
    # first (assert :151 precedes the sortedness loop :152-156).
    for fmt in ("csr", "csc"):
        for case in kats["G7_rejections"][fmt]:
            rc = oracle.validate(case["nrows"], case["ncols"], case["ptr"], case["ind"],
                                 case["nvalues"], csr=(fmt == "csr"))
            assert rc == want[case["name"]], (fmt, case["name"], rc)


def test_g7_accepted(kats, oracle):
    for case in kats["G7_rejections"]["accepted"]:
        assert oracle.validate(case["nrows"], case["ncols"], case["ptr"], case["ind"],
                               case["nvalues"], csr=(case["format"] == "csr")) == 0


def test_validate_sortedness_and_monotone(oracle):
    # unsorted inside a row, all indices in range -> reason 9
    assert oracle.validate(1, 3, [0, 2], [2, 1], 2) == 9
    # duplicate column inside a row is "not strictly increasing"
    assert oracle.validate(1, 3, [0, 2], [1, 1], 2) == 9
    # decreasing rowptr
    assert oracle.validate(2, 3, [0, 2, 1], [0], 1) == 7
    # ok with an empty row
    assert oracle.validate(3, 3, [0, 1, 1, 2], [0, 2], 2) == 0


def test_g8_dok_matrix_spmv(kats, oracle):
    g = kats["G8_dok_matrix"]
    x = np.array([1.0, 2.0, 3.0])
    y = oracle.csr_spmv(g["rowptr"], g["colind"], np.array(g["values"], dtype=F), x)
    assert y.tolist() == [3.0 + 6.0 + 12.0, 15.0]


def test_spmv_equals_literal_reference_route(oracle):
    """The direct SpMV is bit-identical to the reference's own route
    (`&A * &X`, X = n x 1 sparse) for random inputs, CSR and CSC."""
    rng = np.random.default_rng(7)
    for trial in range(20):
        m, n = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        dense = rng.standard_normal((m, n)) * (rng.random((m, n)) < 0.3)
        x = rng.standard_normal(n)
        x[rng.random(n) < 0.3] = 0.0
        # CSR arrays of A
        rp = np.zeros(m + 1, dtype=U)
        ci, va = [], []
        for r in range(m):
            nzc = np.nonzero(dense[r])[0]
            ci += nzc.tolist()
            va += dense[r, nzc].tolist()
            rp[r + 1] = len(ci)
        ci, va = np.array(ci, dtype=U), np.array(va, dtype=F)
        # X as n x 1 CSR: one row per k with a single column 0 when x[k] != 0
        nzx = np.nonzero(x)[0]
        xrp = np.concatenate([[0], np.cumsum(x != 0)]).astype(U)
        xci = np.zeros(nzx.size, dtype=U)
        xva = x[nzx]
        p, i, v = oracle.csr_mul((m, n), (rp, ci, va), (n, 1), (xrp, xci, xva))
        y_lit = np.zeros(m)
        for r in range(m):
            for q in range(int(p[r]), int(p[r + 1])):
                y_lit[r] = v[q]
        # direct, but with x's structural zeros removed the same way: the
        # reference never multiplies by an absent x[k]
        keep = np.isin(ci, nzx)
        rp2 = np.zeros(m + 1, dtype=U)
        rows = np.repeat(np.arange(m), np.diff(rp.astype(np.int64)))
        np.add.at(rp2, rows[keep] + 1, 1)
        rp2 = np.cumsum(rp2).astype(U)
        y_dir = oracle.csr_spmv(rp2, ci[keep], va[keep], x)
        assert np.array_equal(y_lit, y_dir)
        # CSC twin: same numbers bit for bit
        cp, ri, cv = oracle.transpose(m, n, rp2, ci[keep], va[keep])
        y_csc = oracle.csc_spmv(m, cp, ri, cv, x)
        assert np.array_equal(y_csc, y_dir)
        # and the CSC literal route
        xcp = np.array([0, nzx.size], dtype=U)
        p2, i2, v2 = oracle.csc_mul((m, n), (cp, ri, cv), (n, 1), (xcp, nzx.astype(U), xva))
        y_lit2 = np.zeros(m)
        y_lit2[i2.astype(np.int64)] = v2
        assert np.array_equal(y_lit2, y_dir)


def test_no_fma_in_oracle(oracle):
    """mul and add are rounded separately (Rust does not contract)."""
    a, b, c = 1.0 + 2.0 ** -30, 1.0 - 2.0 ** -30, -1.0
    # a*b = 1 - 2^-60 rounds to 1.0; fused would give -2^-60
    y = oracle.csr_spmv([0, 2], [0, 1], np.array([c, a]), np.array([1.0, b]))
    assert y[0] == 0.0


def test_empty_rows_and_first_term_assignment(oracle):
    y = oracle.csr_spmv([0, 0, 1, 1], [0], np.array([-1.0]), np.array([0.0]))
    assert y.tolist() == [0.0, -0.0, 0.0]
    assert np.signbit(y[1]) and not np.signbit(y[0])   # first product assigned, not 0.0 + p


def test_from_coo_semantics(oracle):
    # duplicates summed in insertion order; -0.0 dropped; NaN kept
    big = 2.0 ** 53
    rows = [0, 0, 0, 1, 1, 2]
    cols = [1, 1, 1, 0, 0, 2]
    vals = np.array([big, 1.0, 1.0, -0.0, 0.0, np.nan])
    p, i, v = oracle.coo_to_csr(3, 3, rows, cols, vals)
    assert p.tolist() == [0, 1, 1, 2]
    assert i.tolist() == [1, 2]
    assert v[0] == (big + 1.0) + 1.0 == big        # left-to-right, not big + (1+1)
    assert np.isnan(v[1])


def test_from_coo_vs_scipy(oracle):
    sp = pytest.importorskip("scipy.sparse")
    rng = np.random.default_rng(11)
    for nr, nc, n in [(1, 1, 5), (7, 5, 60), (300, 200, 5000), (50, 1000, 3000)]:
        rows = rng.integers(0, nr, n)
        cols = rng.integers(0, nc, n)
        vals = rng.integers(-3, 4, n).astype(F)       # exact sums, many cancellations
        p, i, v = oracle.coo_to_csr(nr, nc, rows, cols, vals)
        ref = sp.coo_matrix((vals, (rows, cols)), shape=(nr, nc)).tocsr()
        ref.sum_duplicates(); ref.eliminate_zeros(); ref.sort_indices()
        assert p.tolist() == ref.indptr.tolist()
        assert i.tolist() == ref.indices.tolist()
        assert v.tolist() == ref.data.tolist()
        assert oracle.validate(nr, nc, p, i, v.size) == 0
        pc, ic, vc = oracle.coo_to_csc(nr, nc, rows, cols, vals)
        refc = ref.tocsc(); refc.sort_indices()
        assert (pc.tolist(), ic.tolist(), vc.tolist()) == (
            refc.indptr.tolist(), refc.indices.tolist(), refc.data.tolist())


def test_spmv_vs_scipy(oracle):
    sp = pytest.importorskip("scipy.sparse")
    rng = np.random.default_rng(5)
    a = sp.random(500, 400, density=0.02, random_state=3, format="csr", dtype=F)
    a.sort_indices()
    x = rng.standard_normal(400)
    y = oracle.csr_spmv(a.indptr, a.indices, a.data, x)
    np.testing.assert_allclose(y, a @ x, rtol=1e-13, atol=1e-13)
    y32 = oracle.csr_spmv(a.indptr, a.indices, a.data.astype(np.float32), x.astype(np.float32))
    assert y32.dtype == np.float32
    np.testing.assert_allclose(y32, a @ x, rtol=1e-4, atol=1e-4)
