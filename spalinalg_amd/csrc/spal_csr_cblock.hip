// spal_csr_cblock.hip -- plan and launch of the column-blocked CSR kernel (csr_cblock.hpp).
#include "csr_cblock.hpp"
#include "spal_internal.hpp"

namespace spal {

void cblock_free(spal_csr *a) {
    (void)dev_free(a->d_cb_val); a->d_cb_val = nullptr;
    (void)dev_free(a->d_cb_col); a->d_cb_col = nullptr;
    (void)dev_free(a->d_cb_tile); a->d_cb_tile = nullptr;
    (void)dev_free(a->d_cb_cnt); a->d_cb_cnt = nullptr;
    a->plan.cblock = 0;
}

template <typename T, int RPT>
static void fill_t(spal_csr *a, uint32_t nrb, uint32_t nbc) {
    hipLaunchKernelGGL((cb_fill<T, RPT>), dim3(nrb), dim3(kCbThreads), 0, a->stream, a->d_rowptr, a->d_colind,
                       (const T *)a->d_values, a->d_cb_tile, a->d_cb_cnt, (uint32_t)a->nrows, nbc, a->d_cb_col,
                       (T *)a->d_cb_val);
}

// Builds the tiled copy when the matrix qualifies: x larger than an XCD's L2 keeps warm by itself, at most
// kCbMaxBlocks column blocks of 2 MB of x, at most 255 entries of a row per column block, and a row-block height
// (4096 ... 512 rows) whose fullest tile fits the product strip.  `force`: option "cblock" = 1 (tests: small matrices).
// Returns SPAL_OK with a->plan.cblock = 0 when the matrix does not qualify.
int cblock_plan(spal_csr *a, bool force) {
    CsrPlan &p = a->plan;
    cblock_free(a);
    if (a->nnz == 0 || !a->parts.empty()) return SPAL_OK;
    const size_t esz = (size_t)a->elem_size;
    uint32_t shift = esz == 8 ? 18 : 19;                       // 2 MB of x per column block
    if (p.cblock_shift_user > 0) shift = (uint32_t)p.cblock_shift_user;
    const uint64_t nbc64 = (a->ncols + (1ull << shift) - 1) >> shift;
    if (nbc64 > kCbMaxBlocks) return SPAL_OK;
    if (!force && a->ncols * esz <= (size_t)4 << 20) return SPAL_OK;   // x of 4 MB: the L2s hold it anyway
    const uint32_t nbc = (uint32_t)nbc64;
    const double mean = (double)a->nnz / (double)a->nrows;
    if (!force && mean > 64.0) return SPAL_OK;                 // long rows: the vector kernels' business
    // row-block heights, tallest first (fewest bytes of counts per entry); the fullest tile decides
    uint32_t *d_flag = nullptr, *d_tile_n = nullptr;
    int chosen = 0;
    std::vector<uint32_t> tile_n;
    SPAL_HIP_TRY(dev_alloc((void **)&d_flag, 4));
    for (int rpt : {16, 8, 4, 2}) {
        if (p.cblock_rpt_user > 0 && rpt != p.cblock_rpt_user) continue;
        const uint32_t RB = (uint32_t)kCbThreads * (uint32_t)rpt;
        if ((double)RB * mean / (double)nbc > 0.9 * (double)kCbStrip && rpt > 2 && p.cblock_rpt_user <= 0) continue;   // average tile too full: do not even count
        const uint32_t nrb = (uint32_t)((a->nrows + RB - 1) / RB);
        const size_t cnt_bytes = (size_t)nrb * nbc * RB;
        if (cnt_bytes > ((size_t)a->nnz * 12) && !force) break;                  // counts heavier than the entries: not this kernel's matrix
        (void)dev_free(a->d_cb_cnt); a->d_cb_cnt = nullptr;
        (void)dev_free(d_tile_n); d_tile_n = nullptr;
        SPAL_HIP_TRY(dev_alloc((void **)&a->d_cb_cnt, cnt_bytes));
        SPAL_HIP_TRY(dev_alloc((void **)&d_tile_n, (size_t)nrb * nbc * 4));
        SPAL_HIP_TRY(hipMemsetAsync(a->d_cb_cnt, 0, cnt_bytes, a->stream));
        SPAL_HIP_TRY(hipMemsetAsync(d_flag, 0, 4, a->stream));
        hipLaunchKernelGGL(cb_count, dim3((uint32_t)((a->nrows + 255) / 256)), dim3(256), 0, a->stream, a->d_rowptr,
                           a->d_colind, (uint32_t)a->nrows, RB, nbc, shift, a->d_cb_cnt, d_flag);
        hipLaunchKernelGGL(cb_tile_totals, dim3(nrb * nbc), dim3(256), 0, a->stream, a->d_cb_cnt, RB, d_tile_n);
        SPAL_HIP_TRY(hipGetLastError());
        uint32_t flag = 0;
        tile_n.resize((size_t)nrb * nbc);
        SPAL_HIP_TRY(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, a->stream));
        SPAL_HIP_TRY(hipMemcpyAsync(tile_n.data(), d_tile_n, tile_n.size() * 4, hipMemcpyDeviceToHost, a->stream));
        SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
        if (flag) break;                                       // a row with more than 255 entries in one column block
        uint32_t fullest = 0;
        for (uint32_t n : tile_n) fullest = std::max(fullest, n);
        if (fullest <= kCbStrip) { chosen = rpt; break; }
    }
    (void)dev_free(d_flag);
    (void)dev_free(d_tile_n);
    if (!chosen) { cblock_free(a); return SPAL_OK; }
    const uint32_t RB = (uint32_t)kCbThreads * (uint32_t)chosen;
    const uint32_t nrb = (uint32_t)((a->nrows + RB - 1) / RB);
    std::vector<uint32_t> tp((size_t)nrb * nbc + 1);
    uint64_t run = 0;
    for (size_t i = 0; i < tile_n.size(); ++i) { tp[i] = (uint32_t)run; run += tile_n[i]; }
    tp[tile_n.size()] = (uint32_t)run;
    if (run != a->nnz) { cblock_free(a); return fail(SPAL_ERR_HIP, "cblock plan: %llu entries counted, %llu stored", (unsigned long long)run, (unsigned long long)a->nnz); }
    SPAL_HIP_TRY(dev_alloc((void **)&a->d_cb_tile, tp.size() * 4));
    SPAL_HIP_TRY(dev_alloc((void **)&a->d_cb_col, ((size_t)a->nnz + 256) * 4));
    SPAL_HIP_TRY(dev_alloc((void **)&a->d_cb_val, ((size_t)a->nnz + 256) * esz));
    SPAL_HIP_TRY(hipMemcpyAsync(a->d_cb_tile, tp.data(), tp.size() * 4, hipMemcpyHostToDevice, a->stream));
#define SPAL_CB_FILL(RPT) \
    case RPT: if (esz == 8) fill_t<double, RPT>(a, nrb, nbc); else fill_t<float, RPT>(a, nrb, nbc); break;
    switch (chosen) { SPAL_CB_FILL(16) SPAL_CB_FILL(8) SPAL_CB_FILL(4) SPAL_CB_FILL(2) }
#undef SPAL_CB_FILL
    SPAL_HIP_TRY(hipGetLastError());
    SPAL_HIP_TRY(hipStreamSynchronize(a->stream));             // `tp` goes out of scope
    p.cblock = 1;
    p.cblock_rpt = chosen;
    p.cblock_shift = (int)shift;
    p.cblock_nbc = (int)nbc;
    p.cblock_nrb = nrb;
    return SPAL_OK;
}

template <typename T, int RPT>
static hipError_t launch_cb(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    hipLaunchKernelGGL((csr_spmv_cblock<T, RPT, 8>), dim3(a->plan.cblock_nrb), dim3(kCbThreads), 0, st,
                       (const T *)a->d_cb_val, a->d_cb_col, a->d_cb_tile, a->d_cb_cnt, (const T *)x, (T *)y,
                       (uint32_t)a->nrows, (uint32_t)a->plan.cblock_nbc);
    return hipGetLastError();
}

hipError_t launch_cblock(const spal_csr *a, const void *x, void *y, hipStream_t st) {
#define SPAL_CB_CASE(RPT) \
    case RPT: return a->elem_size == 8 ? launch_cb<double, RPT>(a, x, y, st) : launch_cb<float, RPT>(a, x, y, st);
    switch (a->plan.cblock_rpt) {
        SPAL_CB_CASE(16) SPAL_CB_CASE(8) SPAL_CB_CASE(4) SPAL_CB_CASE(2)
        default: return hipErrorInvalidValue;
    }
#undef SPAL_CB_CASE
}

}  // namespace spal
