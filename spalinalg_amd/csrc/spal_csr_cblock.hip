// spal_csr_cblock.hip -- plan and launch of the column-blocked CSR kernel (csr_cblock.hpp).
#include <cmath>

#include "csr_cblock.hpp"
#include "spal_internal.hpp"

namespace spal {

void cblock_free(spal_csr *a) {
    __atomic_store_n(&a->plan.cblock, 0, __ATOMIC_RELEASE);
    (void)dev_free(a->d_cb_val); a->d_cb_val = nullptr;
    (void)dev_free(a->d_cb_col); a->d_cb_col = nullptr;
    (void)dev_free(a->d_cb_tile); a->d_cb_tile = nullptr;
    (void)dev_free(a->d_cb_cnt); a->d_cb_cnt = nullptr;
    (void)dev_free(a->d_cb_row); a->d_cb_row = nullptr;
    __atomic_store_n(&a->plan.cblock, 0, __ATOMIC_RELEASE);
}

// entry form: two product strips (producers fill one while consumers sum the other), the running sums, two row strips
static size_t cb_lds_bytes(uint32_t RB, uint32_t S, size_t esz) {
    return 2 * (size_t)S * esz + (size_t)((RB + 1u) & ~1u) * esz + 2 * ((size_t)S + 2) * 2;
}
// The entry form runs best with exactly TWO workgroups per CU (csr_cblock.hpp): the plan keeps a workgroup's LDS below
// half a CU's, the launch pads a smaller request so that a third does not fit.
constexpr size_t kCbLdsTwo = 79 * 1024, kCbLdsPad = 56 * 1024, kCbLdsMax = 159 * 1024;
// counts of one geometry: cnt8 (kept on the handle), the tiles' entry counts (host), the rows-with-entries-per-tile total
static int cb_count_tiles(spal_csr *a, uint32_t RB, uint32_t nbc, uint32_t shift, std::vector<uint32_t> &tile_n, uint64_t &runs,
                          bool &too_long) {
    const uint32_t nrb = (uint32_t)((a->nrows + RB - 1) / RB);
    const size_t cnt_bytes = (size_t)nrb * nbc * RB;
    uint32_t *d_flag = nullptr, *d_tile_n = nullptr;
    (void)dev_free(a->d_cb_cnt); a->d_cb_cnt = nullptr;
    SPAL_HIP_TRY(dev_alloc((void **)&a->d_cb_cnt, cnt_bytes));
    SPAL_HIP_TRY(dev_alloc((void **)&d_tile_n, (size_t)nrb * nbc * 4));
    SPAL_HIP_TRY(dev_alloc((void **)&d_flag, 16));              // {a count above 255, -, runs (8 bytes)}
    hipError_t e = hipMemsetAsync(a->d_cb_cnt, 0, cnt_bytes, a->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, 16, a->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(cb_count, dim3((uint32_t)((a->nrows + 255) / 256)), dim3(256), 0, a->stream, a->d_rowptr,
                           a->d_colind, (uint32_t)a->nrows, RB, nbc, shift, a->d_cb_cnt, d_flag);
        hipLaunchKernelGGL(cb_tile_totals, dim3(nrb * nbc), dim3(256), 0, a->stream, a->d_cb_cnt, RB, d_tile_n,
                           reinterpret_cast<unsigned long long *>(d_flag + 2));
        e = hipGetLastError();
    }
    uint32_t back[4] = {0, 0, 0, 0};
    tile_n.resize((size_t)nrb * nbc);
    if (e == hipSuccess) e = hipMemcpyAsync(back, d_flag, 16, hipMemcpyDeviceToHost, a->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tile_n.data(), d_tile_n, tile_n.size() * 4, hipMemcpyDeviceToHost, a->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(a->stream);
    (void)dev_free(d_flag);
    (void)dev_free(d_tile_n);
    SPAL_HIP_TRY(e);
    too_long = back[0] != 0;
    runs = (uint64_t)back[2] | ((uint64_t)back[3] << 32);
    return SPAL_OK;
}

// Builds the tiled copy when the matrix qualifies: x larger than an XCD's L2 keeps warm by itself, at most kCbMaxBlocks
// column blocks, at most 255 entries of a row per column block (the builder's counts are bytes), and a row-block height
// whose fullest tile fits the product strip.  Entry form: the height is chosen so that the launch is (nearly) a WHOLE
// number of rounds of the two workgroups per CU the kernel runs with -- the workgroups of a round run side by side through
// the column blocks (that is what keeps one x slice in L2), so a launch of 3.07 rounds takes as long as one of four
// (5M x 5M: 1536 workgroups of 3256 rows 412 us, 1571 of 3184 rows 485 us).  `force`: option "cblock" = 1 (tests: small
// matrices).  Returns SPAL_OK with a->plan.cblock = 0 when the matrix does not qualify.
static int cblock_build(spal_csr *a, bool force) {
    CsrPlan &p = a->plan;
    cblock_free(a);
    if (a->nnz == 0 || !a->parts.empty()) return SPAL_OK;
    const size_t esz = (size_t)a->elem_size;
    const uint32_t shift0 = esz == 8 ? 18 : 19;                // 2 MB of x per column block ...
    if (((a->ncols + (1ull << shift0) - 1) >> shift0) > kCbMaxBlocks && p.cblock_shift_user <= 0) return SPAL_OK;
    if (!force && a->ncols * esz <= (size_t)4 << 20) return SPAL_OK;   // x of 4 MB: the L2s hold it anyway
    const double mean = (double)a->nnz / (double)a->nrows;
    if (!force && mean > 64.0) return SPAL_OK;                 // long rows: the vector kernels' business
    std::vector<uint32_t> tile_n;
    uint64_t runs = 0;
    bool too_long = false;
    // Which form: how many entries a row holds per column block it touches (a RUN), counted at the 2 MB width.  Runs of
    // about one entry (10 per row over 20 blocks): the entry-parallel kernel; two and more (14 per row over 4 ... 8
    // blocks, wide bands): the rows form, whose threads own rows (csr_cblock.hpp).
    int form = p.cblock_form_user;
    if (form < 0) {
        const uint32_t sh = p.cblock_shift_user > 0 ? (uint32_t)p.cblock_shift_user : shift0;
        const uint64_t nb = (a->ncols + (1ull << sh) - 1) >> sh;
        if (nb > kCbMaxBlocks) return SPAL_OK;
        if ((size_t)((a->nrows + 1023) / 1024) * nb * 1024 > (size_t)a->nnz * 12 && !force) {
            // the counts (a byte per row and column block) would be heavier than the entries: rows this sparse per
            // column block hold about one entry per run -- the entry form, whose candidates carry the same cap below
            form = 0;
        } else {
            SPAL_TRY(cb_count_tiles(a, 1024, (uint32_t)nb, sh, tile_n, runs, too_long));
            if (too_long) { cblock_free(a); return SPAL_OK; }
            form = (runs && (double)a->nnz / (double)runs >= 1.6) ? 1 : 0;
        }
    }
    // 1 MB of f64 x (2^17 columns; f32: 2^17 as well, 0.5 MB) per column block measured best for both forms: the slice
    // shares its XCD's 4 MB L2 with the streamed entries and with the slices of workgroups a block ahead or behind
    // (5M x 5M entry form, 3256 rows: 2^16 440 us, 2^17 383, 2^18 414, 2^19 595; f32 366 / 308 / 330; rows form:
    // profiles/r03/cblock_sweeps.txt) -- wider only where the matrix has more than kCbMaxBlocks of them
    uint32_t shift = form == 1 ? shift0 - 1 : 17;
    while (((a->ncols + (1ull << shift) - 1) >> shift) > kCbMaxBlocks) ++shift;
    if (form == 0) {
        // ... narrower where rows are long: the thread that heads a row's run inside a tile adds the whole run, so runs
        // should hold about one entry
        while (shift > 15 && mean / (double)((a->ncols + (1ull << shift) - 1) >> shift) > 1.25 &&
               ((a->ncols + (1ull << (shift - 1)) - 1) >> (shift - 1)) <= kCbMaxBlocks) --shift;
    }
    if (p.cblock_shift_user > 0) shift = (uint32_t)p.cblock_shift_user;
    const uint64_t nbc64 = (a->ncols + (1ull << shift) - 1) >> shift;
    if (nbc64 > kCbMaxBlocks) return SPAL_OK;
    const uint32_t nbc = (uint32_t)nbc64;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, a->device);
    // what the fullest tile of a height is expected to hold (before the tiles are counted): the average + five sigma of
    // a Poisson count, in steps of 256 entries
    auto strip_for = [&](uint32_t RB) {
        const double avg = (double)RB * mean / (double)nbc;
        return std::min(kCbStrip, std::max(512u, ((uint32_t)(avg + 5.0 * std::sqrt(avg) + 32.0) + 255u) & ~255u));
    };
    std::vector<uint32_t> cands;
    if (p.cblock_rows_user > 0) cands.push_back((uint32_t)p.cblock_rows_user);
    else if (form == 1) {
        for (uint32_t RB : {4096u, 2048u, 1024u, 512u, 256u})     // 256 threads x 16 ... 1 rows each; the tallest whose tiles fit
            if ((double)RB * mean / (double)nbc <= 0.9 * (double)kCbStrip || RB == 256u) cands.push_back(RB);
    } else {
        // entry form, tallest first: the tallest height whose workgroup fits half a CU's LDS and whose launch fills 98.5 % of
        // its last round (or is 8 rounds and more long), else the best filled
        uint32_t best = 0;
        double best_eff = 0.0;
        const uint32_t slots = 2u * (uint32_t)cus;
        for (uint32_t RB = kCbMaxRows; RB >= 256; RB -= 8) {
            if (cb_lds_bytes(RB, strip_for(RB), esz) > kCbLdsTwo) continue;
            if ((size_t)((a->nrows + RB - 1) / RB) * nbc * RB > (size_t)a->nnz * 12 && !force) continue;   // counts heavier than the entries
            const double rounds = (double)((a->nrows + RB - 1) / RB) / (double)slots;
            const double eff = rounds / std::ceil(rounds);
            if (eff >= 0.985 || rounds >= 8.0) { cands.push_back(RB); break; }
            if (eff > best_eff) { best_eff = eff; best = RB; }
        }
        if (cands.empty() && best) cands.push_back(best);
        for (uint32_t RB : {2048u, 1024u, 512u}) cands.push_back(RB);   // (when the fullest tile of the first choice does not fit)
    }
    uint32_t chosen = 0, S = 0;
    for (uint32_t RB : cands) {
        if (RB < 1 || RB > kCbMaxRows) continue;
        if (form == 1 && RB != 4096 && RB != 2048 && RB != 1024 && RB != 512 && RB != 256) continue;   // whole rows per thread
        const uint32_t nrb = (uint32_t)((a->nrows + RB - 1) / RB);
        if (form == 1 && (size_t)nrb * nbc * RB > (size_t)a->nnz * 12 && !force) break;   // counts heavier than the entries
        SPAL_TRY(cb_count_tiles(a, RB, nbc, shift, tile_n, runs, too_long));
        if (too_long) break;                                   // a row with more than 255 entries in one column block
        uint32_t fullest = 0;
        for (uint32_t n : tile_n) fullest = std::max(fullest, n);
        if (fullest > kCbStrip) continue;
        S = std::max(512u, (fullest + 255u) & ~255u);         // (the strips are LDS: what the fullest tile needs, no more)
        if (form == 0 && cb_lds_bytes(RB, S, esz) > (p.cblock_rows_user > 0 || force ? kCbLdsMax : kCbLdsTwo)) continue;
        chosen = RB;
        break;
    }
    if (getenv("SPAL_CBLOCK_DEBUG"))
        fprintf(stderr, "[spal cblock] %llu x %llu, %.2f per row, %s form, %u column blocks of 2^%u: %u rows per block, strip %u, %.2f entries per run%s\n",
                (unsigned long long)a->nrows, (unsigned long long)a->ncols, mean, form ? "rows" : "entry", nbc, shift, chosen, S,
                runs ? (double)a->nnz / (double)runs : 0.0, chosen ? "" : " (does not qualify)");
    if (!chosen) { cblock_free(a); return SPAL_OK; }
    const uint32_t RB = chosen;
    const uint32_t nrb = (uint32_t)((a->nrows + RB - 1) / RB);
    std::vector<uint32_t> tp((size_t)nrb * nbc + 1);
    uint64_t run = 0;
    for (size_t i = 0; i < tile_n.size(); ++i) { tp[i] = (uint32_t)run; run += tile_n[i]; }
    tp[tile_n.size()] = (uint32_t)run;
    if (run != a->nnz) { cblock_free(a); return fail(SPAL_ERR_HIP, "cblock plan: %llu entries counted, %llu stored", (unsigned long long)run, (unsigned long long)a->nnz); }
    SPAL_HIP_TRY(dev_alloc((void **)&a->d_cb_tile, tp.size() * 4));
    SPAL_HIP_TRY(dev_alloc((void **)&a->d_cb_col, ((size_t)a->nnz + 256) * 4));
    SPAL_HIP_TRY(dev_alloc((void **)&a->d_cb_val, ((size_t)a->nnz + 256) * esz));
    if (form == 0) SPAL_HIP_TRY(dev_alloc((void **)&a->d_cb_row, ((size_t)a->nnz + 256) * 2));
    SPAL_HIP_TRY(hipMemcpyAsync(a->d_cb_tile, tp.data(), tp.size() * 4, hipMemcpyHostToDevice, a->stream));
    if (esz == 8)
        hipLaunchKernelGGL(cb_fill<double>, dim3(nrb), dim3(kCbThreads), 0, a->stream, a->d_rowptr, a->d_colind, (const double *)a->d_values,
                           a->d_cb_tile, a->d_cb_cnt, (uint32_t)a->nrows, nbc, RB, a->d_cb_col, (double *)a->d_cb_val, a->d_cb_row);
    else
        hipLaunchKernelGGL(cb_fill<float>, dim3(nrb), dim3(kCbThreads), 0, a->stream, a->d_rowptr, a->d_colind, (const float *)a->d_values,
                           a->d_cb_tile, a->d_cb_cnt, (uint32_t)a->nrows, nbc, RB, a->d_cb_col, (float *)a->d_cb_val, a->d_cb_row);
    SPAL_HIP_TRY(hipGetLastError());
    SPAL_HIP_TRY(hipStreamSynchronize(a->stream));             // `tp` goes out of scope
    if (form == 0) { (void)dev_free(a->d_cb_cnt); a->d_cb_cnt = nullptr; }   // the counts were the builder's; the entry form reads the entries' rows
    p.cblock_form = form;
    p.cblock_run = runs ? (float)((double)a->nnz / (double)runs) : 0.f;
    p.cblock_rows = (int)RB;
    p.cblock_strip = (int)S;
    p.cblock_shift = (int)shift;
    p.cblock_nbc = (int)nbc;
    p.cblock_nrb = nrb;
    // published LAST and with release: a product running on another thread (spal_csr_spmv_dev takes no lock) reads it
    // with acquire in launch_lanes and then sees either 0 -- the stream kernels, which touch none of this -- or the
    // complete geometry and arrays
    __atomic_store_n(&p.cblock, 1, __ATOMIC_RELEASE);
    return SPAL_OK;
}

// The tiled copy is OPTIONAL (the stream kernels compute the same product): when building it fails -- out of memory for
// the second copy, counts that do not add up -- the matrix "does not qualify": what was allocated is freed, the error is
// cleared and the handle keeps running the kernels it has.  `cblock_failed` in spal_csr_describe says so.
int cblock_plan(spal_csr *a, bool force) {
    a->cblock_failed = 0;
    const int st = cblock_build(a, force);
    if (st != SPAL_OK) {
        cblock_free(a);
        (void)hipGetLastError();
        a->cblock_failed = 1;
    }
    return SPAL_OK;
}

template <typename T>
static hipError_t launch_cb(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    const CsrPlan &p = a->plan;
    // (padded: a request below a third of a CU's LDS would let a third workgroup in -- two measured best, csr_cblock.hpp)
    const size_t lds = std::max(cb_lds_bytes((uint32_t)p.cblock_rows, (uint32_t)p.cblock_strip, sizeof(T)), kCbLdsPad);
    auto kern = csr_spmv_cblock<T, 2, 8>;
    static std::atomic<uint64_t> configured{0};
    const uint64_t bit = 1ull << (a->device & 63);
    if (!(configured.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCbLdsMax);
        if (e != hipSuccess) return e;
        configured.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(p.cblock_nrb), dim3(kCbThreads), lds, st, (const T *)a->d_cb_val, a->d_cb_col, a->d_cb_row,
                       a->d_cb_tile, a->d_rowptr, (const T *)x, (T *)y, (uint32_t)a->nrows, (uint32_t)p.cblock_nbc,
                       (uint32_t)p.cblock_rows, (uint32_t)p.cblock_strip);
    return hipGetLastError();
}

template <typename T, int RPT>
static hipError_t launch_cb_rows(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    hipLaunchKernelGGL((csr_spmv_cblock_rows<T, RPT, 8>), dim3(a->plan.cblock_nrb), dim3(kCbThreads), 0, st,
                       (const T *)a->d_cb_val, a->d_cb_col, a->d_cb_tile, a->d_cb_cnt, (const T *)x, (T *)y,
                       (uint32_t)a->nrows, (uint32_t)a->plan.cblock_nbc);
    return hipGetLastError();
}

hipError_t launch_cblock(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    if (a->plan.cblock_form == 0) return a->elem_size == 8 ? launch_cb<double>(a, x, y, st) : launch_cb<float>(a, x, y, st);
#define SPAL_CB_CASE(RPT) \
    case RPT * 256: return a->elem_size == 8 ? launch_cb_rows<double, RPT>(a, x, y, st) : launch_cb_rows<float, RPT>(a, x, y, st);
    switch (a->plan.cblock_rows) {
        SPAL_CB_CASE(16) SPAL_CB_CASE(8) SPAL_CB_CASE(4) SPAL_CB_CASE(2) SPAL_CB_CASE(1)
        default: return hipErrorInvalidValue;
    }
#undef SPAL_CB_CASE
}

}  // namespace spal
