// spal_host.cpp -- host-only parts of libspal_hip.so: error state, the
// constructor invariants of the reference and the row partitioner.  Nothing here touches a device.
#include "spal_internal.hpp"

namespace spal {

std::string &last_error_ref() {
    static thread_local std::string msg;
    return msg;
}

int fail(int status, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return status;
}

unsigned host_threads() {
    static unsigned n = [] {
        unsigned h = std::thread::hardware_concurrency();
        if (const char *e = getenv("SPAL_HOST_THREADS")) {
            int v = atoi(e);
            if (v > 0) return (unsigned)v;
        }
        if (h == 0) h = 1;
        return std::min(h, 32u);
    }();
    return n;
}

void parallel_for(uint64_t n, const std::function<void(uint64_t, uint64_t, unsigned)> &fn,
                  uint64_t min_chunk) {
    if (n == 0) return;
    unsigned nt = host_threads();
    uint64_t max_by_size = (n + min_chunk - 1) / min_chunk;
    if (max_by_size < nt) nt = (unsigned)max_by_size;
    if (nt <= 1) {
        fn(0, n, 0);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(nt);
    uint64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        uint64_t b = std::min<uint64_t>(n, (uint64_t)t * per);
        uint64_t e = std::min<uint64_t>(n, b + per);
        if (b >= e) break;
        th.emplace_back([&fn, b, e, t] { fn(b, e, t); });
    }
    for (auto &t : th) t.join();
}

// ---------------------------------------------------------------------------
// CsrMatrix::new / CscMatrix::new  (reference src/csr.rs:144-156,
// src/csc.rs:144-156).  The reference asserts in sequence, each assertion over
// the whole array, so "the first assertion that fails" is the lowest ordinal
// that fails anywhere -- which lets the three O(nnz) scans run in parallel.
// ---------------------------------------------------------------------------
int compressed_validate(uint64_t nrows, uint64_t ncols, bool major_is_rows,
                        const uint64_t *ptr, uint64_t ptr_len, const uint64_t *ind,
                        uint64_t ind_len, uint64_t val_len) {
    const uint64_t nmajor = major_is_rows ? nrows : ncols;
    const uint64_t nminor = major_is_rows ? ncols : nrows;
    if (!(nrows > 0)) return 1;
    if (!(ncols > 0)) return 2;
    if (!(ptr_len == nmajor + 1)) return 3;
    if (!(ptr[0] == 0)) return 4;
    if (!(ind_len == ptr[nmajor])) return 5;
    if (!(val_len == ptr[nmajor])) return 6;
    const unsigned nt = host_threads();
    std::vector<int> bad7(nt, 0), bad8(nt, 0), bad9(nt, 0);
    parallel_for(nmajor, [&](uint64_t b, uint64_t e, unsigned t) {
        for (uint64_t i = b; i < e; ++i)
            if (!(ptr[i] <= ptr[i + 1])) { bad7[t] = 1; break; }
    });
    for (int f : bad7) if (f) return 7;
    parallel_for(ind_len, [&](uint64_t b, uint64_t e, unsigned t) {
        for (uint64_t p = b; p < e; ++p)
            if (!(ind[p] < nminor)) { bad8[t] = 1; break; }
    });
    for (int f : bad8) if (f) return 8;
    // ptr is now known monotone with ptr[nmajor] == ind_len: slices are in range
    parallel_for(nmajor, [&](uint64_t b, uint64_t e, unsigned t) {
        for (uint64_t m = b; m < e && !bad9[t]; ++m)
            for (uint64_t p = ptr[m]; p + 1 < ptr[m + 1]; ++p)
                if (!(ind[p] < ind[p + 1])) { bad9[t] = 1; break; }
    });
    for (int f : bad9) if (f) return 9;
    return 0;
}

const char *invariant_text(int reason, bool csr) {
    switch (reason) {
        case 1: return "nrows > 0";
        case 2: return "ncols > 0";
        case 3: return csr ? "rowptr.len() == nrows + 1" : "colptr.len() == ncols + 1";
        case 4: return csr ? "rowptr[0] == 0" : "colptr[0] == 0";
        case 5: return csr ? "colind.len() == rowptr[nrows]" : "rowind.len() == colptr[ncols]";
        case 6: return csr ? "values.len() == rowptr[nrows]" : "values.len() == colptr[ncols]";
        case 7: return csr ? "rowptr is sorted" : "colptr is sorted";
        case 8: return csr ? "every colind < ncols" : "every rowind < nrows";
        case 9: return csr ? "colind strictly increasing inside each row"
                           : "rowind strictly increasing inside each column";
        default: return "ok";
    }
}

}  // namespace spal

using namespace spal;

extern "C" {

const char *spal_last_error(void) { return last_error_ref().c_str(); }

const char *spal_version(void) { return "spalinalg_amd 0.1.0 (gfx950)"; }

int spal_csr_validate(uint64_t nrows, uint64_t ncols, const uint64_t *rowptr,
                      uint64_t rowptr_len, const uint64_t *colind, uint64_t colind_len,
                      uint64_t values_len, int *reason) {
    if (reason) *reason = 0;
    if ((!rowptr && rowptr_len) || (!colind && colind_len))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_validate: null array");
    int r = compressed_validate(nrows, ncols, true, rowptr, rowptr_len, colind, colind_len,
                                values_len);
    if (reason) *reason = r;
    if (r) return fail(SPAL_ERR_INVARIANT, "CsrMatrix::new would panic: assertion failed: %s",
                       invariant_text(r, true));
    return SPAL_OK;
}

int spal_csc_validate(uint64_t nrows, uint64_t ncols, const uint64_t *colptr,
                      uint64_t colptr_len, const uint64_t *rowind, uint64_t rowind_len,
                      uint64_t values_len, int *reason) {
    if (reason) *reason = 0;
    if ((!colptr && colptr_len) || (!rowind && rowind_len))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_validate: null array");
    int r = compressed_validate(nrows, ncols, false, colptr, colptr_len, rowind, rowind_len,
                                values_len);
    if (reason) *reason = r;
    if (r) return fail(SPAL_ERR_INVARIANT, "CscMatrix::new would panic: assertion failed: %s",
                       invariant_text(r, false));
    return SPAL_OK;
}

int spal_partition_rows(const uint64_t *rowptr, uint64_t nrows, uint32_t nparts,
                        uint64_t *bounds) {
    if (!rowptr || !bounds || nparts == 0)
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_partition_rows: null array or nparts == 0");
    const uint64_t nnz = rowptr[nrows];
    bounds[0] = 0;
    for (uint32_t g = 1; g < nparts; ++g) {
        // first row whose starting offset reaches g/nparts of the entries;
        // rows (not entries) are split evenly when the matrix is empty
        uint64_t cut;
        if (nnz == 0) {
            cut = (uint64_t)(((unsigned __int128)nrows * g) / nparts);
        } else {
            const uint64_t target = (uint64_t)(((unsigned __int128)nnz * g) / nparts);
            cut = (uint64_t)(std::lower_bound(rowptr, rowptr + nrows + 1, target) - rowptr);
            if (cut > nrows) cut = nrows;
        }
        bounds[g] = std::max(cut, bounds[g - 1]);
    }
    bounds[nparts] = nrows;
    return SPAL_OK;
}

}  // extern "C"
