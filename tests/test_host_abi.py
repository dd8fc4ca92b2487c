"""CPU-side tests of the product library: the C ABI loads and exports every
symbol include/spal.h declares, host logic (constructor invariants, row
partition, generators) behaves like the reference / the spec, and compute
entry points fail loudly without a GPU (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import spalinalg_amd as sp
from spalinalg_amd import _ffi

U = np.uint64


def test_library_exports_every_declared_symbol():
    lib = _ffi.lib()
    names = _ffi.exported_names()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), n
    assert b"gfx950" in lib.spal_version()


def test_header_is_plain_c():
    """no torch / C++ types in the boundary."""
    text = open(_ffi.HEADER_PATH).read()
    assert 'extern "C"' in text
    body = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for banned in ("std::", "torch", "at::", "template", "class "):
        assert banned not in body


@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_g7_rejections_through_the_abi(kats, fmt):
    cls = sp.CsrMatrix if fmt == "csr" else sp.CscMatrix
    for case in kats["G7_rejections"][fmt]:
        with pytest.raises(sp.Panic):
            cls(case["nrows"], case["ncols"], case["ptr"], case["ind"], np.ones(case["nvalues"]))


def test_validate_reason_codes_match_oracle(kats, oracle):
    lib = _ffi.lib()
    rng = np.random.default_rng(0)
    cases = [(c, "csr") for c in kats["G7_rejections"]["csr"]] + \
            [(c, "csc") for c in kats["G7_rejections"]["csc"]] + \
            [(dict(c, name="ok"), c["format"]) for c in kats["G7_rejections"]["accepted"]]
    # plus fuzzed near-valid inputs
    for _ in range(200):
        nr, nc = int(rng.integers(0, 6)), int(rng.integers(0, 6))
        ptr = np.sort(rng.integers(0, 8, size=int(rng.integers(0, 8)))).tolist()
        if ptr and rng.random() < 0.7:
            ptr[0] = 0
        ind = rng.integers(0, 6, size=int(rng.integers(0, 9))).tolist()
        cases.append((dict(nrows=nr, ncols=nc, ptr=ptr, ind=ind,
                           nvalues=int(rng.integers(0, 9)) if rng.random() < 0.3 else len(ind)),
                      "csr" if rng.random() < 0.5 else "csc"))
    for case, fmt in cases:
        ptr = np.asarray(case["ptr"], dtype=U)
        ind = np.asarray(case["ind"], dtype=U)
        reason = C.c_int(-1)
        st = getattr(lib, f"spal_{fmt}_validate")(
            C.c_uint64(case["nrows"]), C.c_uint64(case["ncols"]),
            ptr.ctypes.data_as(_ffi.u64p), C.c_uint64(ptr.size),
            ind.ctypes.data_as(_ffi.u64p), C.c_uint64(ind.size),
            C.c_uint64(case["nvalues"]), C.byref(reason))
        want = oracle.validate(case["nrows"], case["ncols"], ptr, ind, case["nvalues"], csr=(fmt == "csr"))
        assert reason.value == want, (case, fmt)
        assert st == (0 if want == 0 else _ffi.SPAL_ERR_INVARIANT)
        if want:
            assert b"would panic" in lib.spal_last_error()


def test_accessors_mirror_reference():
    a = sp.CsrMatrix.new(2, 3, [0, 1, 3], [0, 1, 2], np.array([1.0, 2.0, 3.0]))
    assert (a.nrows(), a.ncols(), a.nnz()) == (2, 3, 3)
    assert a.rowptr().tolist() == [0, 1, 3] and a.colind().tolist() == [0, 1, 2]
    assert a.values().tolist() == [1.0, 2.0, 3.0] and a.rowptr().dtype == U
    e = sp.CsrMatrix.eye(3)
    assert e.rowptr().tolist() == [0, 1, 2, 3] and e.values().tolist() == [1.0] * 3
    with pytest.raises(sp.Panic):
        sp.CsrMatrix.eye(0)
    c = sp.CscMatrix.new(1, 2, [0, 1, 1], [0], np.array([1.0], dtype=np.float32))
    assert c.colptr().tolist() == [0, 1, 1] and c.rowind().tolist() == [0] and c.dtype == np.float32
    s = a.row_slice(1, 2)
    assert s.rowptr().tolist() == [0, 2] and s.colind().tolist() == [1, 2] and s.ncols() == 3


def test_coo_container_mirrors_reference():
    coo = sp.CooMatrix.new(2, 3)
    coo.push(1, 2, 5.0)
    coo.push(0, 0, 1.0)
    assert coo.length() == 2 and list(coo.iter()) == [(1, 2, 5.0), (0, 0, 1.0)]
    for bad in [(2, 0), (0, 3)]:
        with pytest.raises(sp.Panic):
            coo.push(bad[0], bad[1], 1.0)
    with pytest.raises(sp.Panic):
        sp.CooMatrix.new(0, 1)
    with pytest.raises(sp.Panic):
        sp.CooMatrix.with_triplets(2, 2, [0], [0, 1], np.array([1.0]))
    t = sp.CooMatrix.with_entries(2, 2, [(0, 0, 1.0), (1, 1, 2.0)])
    assert t.length() == 2


def test_dimension_mismatch_panics_before_touching_a_device():
    a = sp.CsrMatrix(2, 3, [0, 1, 3], [0, 1, 2], np.array([1.0, 2.0, 3.0]))
    with pytest.raises(sp.Panic):
        a * np.ones(2)


def test_no_cpu_fallback():
    """without a GPU every compute entry point raises; nothing silently
    computes on the host."""
    if sp.device_count() > 0:
        pytest.skip("a GPU is present")
    a = sp.CsrMatrix(2, 3, [0, 1, 3], [0, 1, 2], np.array([1.0, 2.0, 3.0]))
    with pytest.raises(sp.SpalError) as e:
        a * np.ones(3)
    assert e.value.status == _ffi.SPAL_ERR_NO_DEVICE
    coo = sp.CooMatrix.with_triplets(2, 2, [0], [1], np.array([1.0]))
    with pytest.raises(sp.SpalError):
        sp.CsrMatrix.from_coo(coo)
    c = sp.CscMatrix(1, 2, [0, 1, 1], [0], np.array([1.0]))
    with pytest.raises(sp.SpalError):
        c * np.ones(2)


def test_product_does_not_import_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for top in ("spalinalg_amd", "include", "tools", "rust_shim"):
        for dirpath, _, files in os.walk(os.path.join(root, top)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".rs", ".sh")):
                    text = open(os.path.join(dirpath, f)).read()
                    assert "import oracle" not in text and "from oracle" not in text, f
                    assert "liboracle" not in text and "spal_oracle" not in text, f
    # bench.py may touch the oracle only inside its cpu_baseline legs
    bench = open(os.path.join(root, "bench.py")).read().split("\n")
    for i, line in enumerate(bench):
        if "import oracle" in line:
            assert "CPU baseline leg only" in line, (i, line)
            window = "\n".join(bench[max(0, i - 4):i + 1])
            assert "no_cpu_baseline" in window, (i, "oracle import outside a cpu_baseline leg")


def test_partition_rows_balances_entries():
    lib = _ffi.lib()
    rng = np.random.default_rng(1)
    for nparts in (1, 2, 3, 8):
        lens = rng.integers(0, 40, size=1000)
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(U)
        b = np.empty(nparts + 1, dtype=U)
        assert lib.spal_partition_rows(rp.ctypes.data_as(_ffi.u64p), C.c_uint64(1000),
                                       C.c_uint32(nparts), b.ctypes.data_as(_ffi.u64p)) == 0
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b.astype(np.int64)) >= 0)
        per = np.diff(rp[b.astype(np.int64)].astype(np.int64))
        assert per.sum() == rp[-1]
        assert per.max() - per.min() <= 2 * lens.max()
    # all-empty matrix: rows split evenly
    rp = np.zeros(11, dtype=U)
    b = np.empty(5, dtype=U)
    lib.spal_partition_rows(rp.ctypes.data_as(_ffi.u64p), C.c_uint64(10), C.c_uint32(4),
                            b.ctypes.data_as(_ffi.u64p))
    assert b.tolist() == [0, 2, 5, 7, 10]


def test_shapes_beyond_32_bit_indices_are_refused():
    """device indices are 32-bit: larger shapes fail loudly with
    SPAL_ERR_UNSUPPORTED before any device is touched."""
    a = sp.CsrMatrix(1, 2 ** 33, [0, 1], [2 ** 33 - 1], np.array([1.0]))
    with pytest.raises(sp.SpalError) as e:
        a.device()
    assert e.value.status == _ffi.SPAL_ERR_UNSUPPORTED
    c = sp.CscMatrix(2 ** 33, 1, [0, 1], [2 ** 33 - 1], np.array([1.0]))
    with pytest.raises(sp.SpalError) as e:
        c.device()
    assert e.value.status == _ffi.SPAL_ERR_UNSUPPORTED
    coo = sp.CooMatrix.with_triplets(2 ** 33, 2, [2 ** 33 - 1], [1], np.array([1.0]))
    with pytest.raises(sp.SpalError) as e:
        coo.upload()
    assert e.value.status == _ffi.SPAL_ERR_UNSUPPORTED


def test_option_validation():
    """unknown keys / values the kernels are not instantiated for are rejected
    (checked before any device work for a handle-less call)."""
    lib = _ffi.lib()
    assert lib.spal_csr_set_option(None, b"kernel", C.c_int64(1)) == _ffi.SPAL_ERR_INVALID_ARGUMENT
    assert lib.spal_csc_set_option(None, b"kernel", C.c_int64(1)) == _ffi.SPAL_ERR_INVALID_ARGUMENT
    assert lib.spal_csr_destroy(None) == 0 and lib.spal_csc_destroy(None) == 0 and lib.spal_coo_destroy(None) == 0


def test_rust_ffi_declares_every_export():
    """rust_shim/src/ffi.rs is generated from include/spal.h (tools/gen_rust_ffi.py): the committed file must be
    what the generator makes of the current header, and must declare every exported function."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "gen_rust_ffi.py"), "--check"]).returncode == 0, \
        "rust_shim/src/ffi.rs is stale: run python tools/gen_rust_ffi.py"
    text = open(os.path.join(root, "rust_shim", "src", "ffi.rs")).read()
    from spalinalg_amd import _ffi
    for name in _ffi.exported_names():
        assert f"pub fn {name}(" in text, name
    # every FFI function the hand-written Rust modules call is one the header declares
    import re
    for f in ("scalar.rs", "device.rs", "ops.rs", "multi.rs"):
        for used in set(re.findall(r"ffi::(spal_[a-z0-9_]+)\(", open(os.path.join(root, "rust_shim", "src", f)).read())):
            assert used in _ffi.exported_names(), (f, used)


def _coo_abi_call(lib, entry, case, dtype=np.float64):
    """Call spal_coo_upload_* / spal_coo_to_csr_* with ctypes directly (not through matrix.py, whose own checks
    would fire first).  The ABI takes ONE len for the three arrays."""
    rows = np.asarray(case["rows"], dtype=U)
    cols = np.asarray(case["cols"], dtype=U)
    vals = np.ones(case["nvalues"], dtype=dtype)
    out = C.c_void_p()
    sfx = "f64" if dtype == np.float64 else "f32"
    fp = C.POINTER(C.c_double if dtype == np.float64 else C.c_float)
    st = getattr(lib, f"spal_coo_{entry}_{sfx}")(
        C.c_int(0), C.c_uint64(case["nrows"]), C.c_uint64(case["ncols"]), C.c_uint64(vals.size),
        rows.ctypes.data_as(_ffi.u64p), cols.ctypes.data_as(_ffi.u64p), vals.ctypes.data_as(fp), C.byref(out))
    return st, out


@pytest.mark.parametrize("entry", ["upload", "to_csr", "to_csc"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_g9_coo_rejections_through_the_c_abi(kats, entry, dtype):
    """a-6: the reference's CooMatrix #[should_panic] fixtures (src/coo.rs:819-1012) against the C ABI's OWN checks.
    They run before any device is touched, so this is a CPU test."""
    lib = _ffi.lib()
    for case in kats["G9_coo_rejections"]["cases"]:
        if case["layer"] != "abi":
            continue
        st, out = _coo_abi_call(lib, entry, case, dtype)
        assert st == getattr(_ffi, case["status"]), (case["name"], st)
        assert not out.value, case["name"]
        text = lib.spal_last_error().decode()
        assert "would panic" in text and case["text"] in text, (case["name"], text)


def test_g9_coo_rejections_through_the_mirror(kats):
    """the same fixtures through the host mirror (the layer that sees three array lengths)."""
    for case in kats["G9_coo_rejections"]["cases"]:
        with pytest.raises(sp.Panic) as e:
            if case.get("via_push"):
                m = sp.CooMatrix.new(case["nrows"], case["ncols"])
                m.push(case["rows"][0], case["cols"][0], 1.0)
            else:
                sp.CooMatrix.with_triplets(case["nrows"], case["ncols"], case["rows"], case["cols"],
                                           np.ones(case["nvalues"]))
        assert e.value.status == getattr(_ffi, case["status"]), case["name"]
        assert case["text"] in str(e.value).replace("self.", ""), (case["name"], str(e.value))
    for case in kats["G9_coo_rejections"]["accepted"]:
        m = sp.CooMatrix.with_triplets(case["nrows"], case["ncols"], case["rows"], case["cols"],
                                       np.ones(case["nvalues"]))
        assert m.length() == case["nvalues"]


def test_coo_abi_reports_the_first_offending_entry_and_shape_limits():
    lib = _ffi.lib()
    # entry 2 is the first out of bounds (its row); entry 3 (a column) comes later
    case = dict(nrows=4, ncols=4, rows=[0, 3, 4, 1], cols=[0, 3, 0, 9], nvalues=4)
    st, _ = _coo_abi_call(lib, "upload", case)
    assert st == _ffi.SPAL_ERR_INDEX_OUT_OF_BOUNDS
    assert "row < nrows (entry 2: row 4)" in lib.spal_last_error().decode()
    case = dict(nrows=4, ncols=4, rows=[0, 3, 3, 1], cols=[0, 3, 0, 9], nvalues=4)
    st, _ = _coo_abi_call(lib, "to_csr", case)
    assert st == _ffi.SPAL_ERR_INDEX_OUT_OF_BOUNDS
    assert "col < ncols (entry 3: col 9)" in lib.spal_last_error().decode()
    # 32-bit device indices: a COO handle may become CSR or CSC, so both dimensions are bounded alike
    for nr, nc in ((0xffffffff, 1), (1, 0xffffffff), (1 << 40, 1 << 40)):
        st, _ = _coo_abi_call(lib, "upload", dict(nrows=nr, ncols=nc, rows=[], cols=[], nvalues=0))
        assert st == _ffi.SPAL_ERR_UNSUPPORTED, (nr, nc)
    # null arrays with len > 0, and a null out pointer
    st = lib.spal_coo_upload_f64(C.c_int(0), C.c_uint64(1), C.c_uint64(1), C.c_uint64(1), None, None, None,
                                 C.byref(C.c_void_p()))
    assert st == _ffi.SPAL_ERR_INVALID_ARGUMENT
    st = lib.spal_coo_upload_f64(C.c_int(0), C.c_uint64(1), C.c_uint64(1), C.c_uint64(0), None, None, None, None)
    assert st == _ffi.SPAL_ERR_INVALID_ARGUMENT
