#!/usr/bin/env python3
"""Generates rust_shim/src/ffi.rs from include/spal.h: one `extern "C"` declaration per exported function,
so that the Rust binding cannot drift from the C ABI (tests/test_host_abi.py regenerates and compares).

    python tools/gen_rust_ffi.py            # rewrites rust_shim/src/ffi.rs
    python tools/gen_rust_ffi.py --check    # exit 1 if the committed file is stale
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "spal.h")
OUT = os.path.join(ROOT, "rust_shim", "src", "ffi.rs")

SCALARS = {"int": "c_int", "uint64_t": "u64", "uint32_t": "u32", "int64_t": "i64", "size_t": "usize", "double": "f64",
           "float": "f32", "void": "c_void", "char": "c_char"}
HANDLES = {"spal_csr_t": "spal_csr", "spal_csc_t": "spal_csc", "spal_coo_t": "spal_coo", "spal_mg_t": "spal_mg",
           "spal_mg_csr_t": "spal_mg_csr"}


def rust_type(ctype: str) -> str:
    """`const uint64_t *` -> `*const u64`, `spal_csr_t *` -> `*mut *mut spal_csr`, `void **` -> `*mut *mut c_void` ..."""
    t = ctype.strip()
    stars = t.count("*")
    base = t.replace("*", " ").split()
    const = "const" in base
    base = [b for b in base if b != "const"]
    assert len(base) == 1, ctype
    b = base[0]
    if b in HANDLES:
        inner, stars = HANDLES[b], stars + 1      # the handle typedefs are pointers themselves
    else:
        inner = SCALARS[b]
    out = inner
    for level in range(stars):
        # only the innermost pointer of `const T *` is const; handles and out-parameters are *mut
        out = ("*const " if (const and level == 0 and b not in HANDLES) else "*mut ") + out
    return out


def parse(header_text: str):
    text = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    fns = []
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(spal_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = []
        if args not in ("void", ""):
            for a in args.split(","):
                a = a.strip()
                pm = re.match(r"(.*?[\s\*])([A-Za-z_][A-Za-z0-9_]*)$", a)
                assert pm, a
                params.append((pm.group(2), rust_type(pm.group(1))))
        fns.append((name, params, rust_type(ret)))
    return fns


def render(fns) -> str:
    lines = ['//! Raw bindings to libspal_hip.so -- GENERATED from include/spal.h by tools/gen_rust_ffi.py; do not edit.',
             '//!',
             '//! NOT COMPILED IN THIS REPOSITORY\'S PIPELINE: the build image has no rustc/cargo (SURVEY.md F7).  This is',
             '//! the source a spalinalg maintainer adds to the crate as `src/hip/ffi.rs`; see INTEGRATION.md.  Every',
             '//! function of the C ABI is declared (tests/test_host_abi.py keeps this file in step with the header).',
             '#![allow(non_camel_case_types)]',
             'use std::os::raw::{c_char, c_int, c_void};',
             '']
    for h in HANDLES.values():
        lines.append(f"#[repr(C)] pub struct {h} {{ _private: [u8; 0] }}")
    lines += ['',
              'pub const SPAL_OK: c_int = 0;',
              'pub const SPAL_ERR_INVALID_ARGUMENT: c_int = 1;',
              'pub const SPAL_ERR_INVARIANT: c_int = 2;',
              'pub const SPAL_ERR_HIP: c_int = 3;',
              'pub const SPAL_ERR_OUT_OF_MEMORY: c_int = 4;',
              'pub const SPAL_ERR_UNSUPPORTED: c_int = 5;',
              'pub const SPAL_ERR_NO_DEVICE: c_int = 6;',
              'pub const SPAL_ERR_INDEX_OUT_OF_BOUNDS: c_int = 7;',
              '',
              '#[link(name = "spal_hip")]',
              'extern "C" {']
    for name, params, ret in fns:
        sig = ", ".join(f"{'r#' + p if p in ('type', 'ref', 'in', 'fn') else p}: {t}" for p, t in params)
        lines.append(f"    pub fn {name}({sig}) -> {ret};")
    lines += ['}',
              '',
              '/// The reference panics on contract violations (`assert!`, src/csr.rs:144-156; `assert_eq!`,',
              '/// src/csr/ops/mul.rs:9); every non-zero status keeps that convention.',
              'pub fn check(status: c_int) {',
              '    if status != SPAL_OK {',
              '        let msg = unsafe { std::ffi::CStr::from_ptr(spal_last_error()) }.to_string_lossy().into_owned();',
              '        panic!("spal_hip status {}: {}", status, msg);',
              '    }',
              '}',
              '']
    return "\n".join(lines)


def main():
    fns = parse(open(HEADER).read())
    text = render(fns)
    if "--check" in sys.argv[1:]:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"{OUT}: {len(fns)} functions")


if __name__ == "__main__":
    main()
