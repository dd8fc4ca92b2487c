// spal_csr_slide.hip -- instantiations and launch of the sliding-window CSR kernel (csr_slide.hpp); a
// translation unit of its own so that the build compiles it beside spal_csr.hip.
#include "csr_panel.hpp"
#include "csr_slide.hpp"
#include "spal_internal.hpp"

namespace spal {

template <typename T, int RPT, int S, bool UNI>
static hipError_t launch_slide_inst(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    constexpr int PF = 2;
    const CsrPlan &p = a->plan;
    const size_t lds = ((size_t)kStreamWaves * stream_strip<false>() + (size_t)p.ring_pages * kPageCols) * sizeof(T);
    // as many workgroups as the device holds at once (160 KB of LDS per CU decide), each a contiguous chunk of
    // its XCD's run of steps
    const int per_cu = (int)std::min<size_t>(8, std::max<size_t>(1, (160 * 1024) / (lds + 512)));
    const uint32_t grid = p.persistent_blocks > 0 ? (uint32_t)p.persistent_blocks : 256u * (uint32_t)per_cu;
    const uint32_t per_xcd = (p.slide_steps + 7u) / 8u;
    const uint32_t slots = std::max(1u, grid / 8u);
    // steps per run: one run per workgroup, or what the plan says ("slide_run": shorter runs dealt round-robin)
    const uint32_t one_run = (per_xcd + slots - 1u) / slots;
    const uint32_t chunk = p.slide_run > 0 ? std::min<uint32_t>((uint32_t)p.slide_run, one_run) : one_run;
    const bool even = p.slide_run <= 0 && p.slide_even;        // one run per workgroup: the steps split evenly (csr_slide.hpp)
    const uint32_t used = even ? std::min(slots, per_xcd) : std::min(slots, (per_xcd + chunk - 1u) / chunk);
    auto kern = csr_spmv_slide<T, RPT, S, UNI, PF>;
    static std::atomic<uint64_t> configured{0};
    const uint64_t bit = 1ull << (a->device & 63);
    if (lds > 48 * 1024 && !(configured.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(used * 8u), dim3(kStreamBlock), lds, st, a->d_rowptr, a->d_col16,
                       (const T *)a->d_values, (const T *)x, (T *)y, a->d_sdesc, (uint32_t)a->nrows, (uint32_t)a->ncols,
                       p.slide_steps, per_xcd, chunk, (uint32_t)p.ring_pages, (uint32_t)p.slide_uniform,
                       (uint32_t)(p.nt_store ? 1 : 0) | (uint32_t)p.diag | ((UNI && p.all_rows_uniform && p.arith_bounds) ? 4u : 0u) | (even ? 8u : 0u));
    return hipGetLastError();
}

template <typename T, int RPT, bool UNI>
static hipError_t launch_slide_steps(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    // every tile issues S loads per array; S = 4 serves the plans whose tiles are smaller still
    switch (std::max(4, a->plan.slide_S)) {
        case 4: return launch_slide_inst<T, RPT, 4, UNI>(a, x, y, st);
        case 5: return launch_slide_inst<T, RPT, 5, UNI>(a, x, y, st);
        case 6: return launch_slide_inst<T, RPT, 6, UNI>(a, x, y, st);
        case 7: return launch_slide_inst<T, RPT, 7, UNI>(a, x, y, st);
        case 8: return launch_slide_inst<T, RPT, 8, UNI>(a, x, y, st);
        default: return hipErrorInvalidValue;
    }
}

template <typename T, int RPT>
static hipError_t launch_slide_uni(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    return (a->plan.slide_uniform && a->plan.uniform_rows) ? launch_slide_steps<T, RPT, true>(a, x, y, st)
                                                          : launch_slide_steps<T, RPT, false>(a, x, y, st);
}

template <typename T>
static hipError_t launch_slide_rpt(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    switch (a->plan.rows_per_tile) {
        case 64: return launch_slide_uni<T, 64>(a, x, y, st);
        case 32: return launch_slide_uni<T, 32>(a, x, y, st);
        case 24: return launch_slide_uni<T, 24>(a, x, y, st);
        case 16: return launch_slide_uni<T, 16>(a, x, y, st);
        case 12: return launch_slide_uni<T, 12>(a, x, y, st);
        case 8: return launch_slide_uni<T, 8>(a, x, y, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_slide(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    return a->elem_size == 8 ? launch_slide_rpt<double>(a, x, y, st) : launch_slide_rpt<float>(a, x, y, st);
}

template <typename T, int RPT>
static hipError_t launch_panel_inst(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    const CsrPlan &p = a->plan;
    // the panel window; the strips reuse it after the last panel
    const size_t lds = std::max((size_t)kStreamWaves * stream_strip<false>(), (size_t)p.panel_window_pages * kPageCols) * sizeof(T);
    auto kern = csr_spmv_panel<T, RPT>;
    static std::atomic<uint64_t> configured{0};
    const uint64_t bit = 1ull << (a->device & 63);
    if (lds > 48 * 1024 && !(configured.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(a->n_ptiles), dim3(kStreamBlock), lds, st, a->d_rowptr, a->d_col16,
                       (const T *)a->d_values, (const T *)x, (T *)y, a->d_ptiles, a->d_pwin, a->d_desc, a->n_ptiles,
                       (uint32_t)a->nrows, (uint32_t)a->ncols, (uint32_t)p.panel_window_pages,
                       (uint32_t)(p.nt_store == 1 ? 1 : 0));
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_panel_rpt(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    switch (a->plan.rows_per_tile) {
        case 64: return launch_panel_inst<T, 64>(a, x, y, st);
        case 32: return launch_panel_inst<T, 32>(a, x, y, st);
        case 24: return launch_panel_inst<T, 24>(a, x, y, st);
        case 16: return launch_panel_inst<T, 16>(a, x, y, st);
        case 12: return launch_panel_inst<T, 12>(a, x, y, st);
        case 8: return launch_panel_inst<T, 8>(a, x, y, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_panel(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    return a->elem_size == 8 ? launch_panel_rpt<double>(a, x, y, st) : launch_panel_rpt<float>(a, x, y, st);
}

}  // namespace spal
