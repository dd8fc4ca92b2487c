// spal_csc_rowtiles.hip -- CSC y = A*x by LDS-privatised atomic scatter over ROW TILES (config 4's path, round 3).
//
// The column-tiled scatter kernel (spal_csc.hip) gives a workgroup 4096 columns; the rows those touch overlap the
// neighbours' (a band as wide as the tile: every column adds to rows shared with BOTH neighbours), so the windows meet in
// y through a hand-off -- own rows stored and acknowledged, a flag, the neighbour's poll, its read-add-write -- which costs
// 5.6 of the launch's 36 us after the last entry and cannot start earlier (profiles/r03/csc_handoff.txt).  Here the
// handle keeps the entries a second time ordered (row tile, column, row): a workgroup owns RT consecutive ROWS of y
// outright and streams exactly the entries that fall into them, column after column as the reference's loop does
// (src/csc/ops/mul.rs:26-46: for each column k, for each entry (i, v): y[i] += v * x[k]); the columns those entries come
// from form a window of x that is staged in LDS (the mirror image of the column tile's row window).  Per entry the same
// work as before -- value * x from LDS, a float atomic on the y window in LDS -- but no row is shared: every row of y is
// stored once by plain stores, no memset, no global atomic, no flag, no order between workgroups or launches (the
// product can be captured into a graph and run on any stream).  The order of the adds inside a row is not fixed (LDS
// atomics): tolerance parity, as for the column tiles.
// Built where every row tile's column window fits LDS beside its rows (bands; config 4: 4096 rows + 8192 columns of f64
// = 96 KB, one workgroup per CU, 245 workgroups = one round); otherwise the column tiles run.  12 bytes per entry
// (value, packed row-in-tile | column-in-window << 16) beside the CSC arrays.
#include "csr_kernels.hpp"
#include "spal_internal.hpp"

namespace spal {

constexpr int kRtThreads = 1024;
constexpr uint32_t kRtU = 2;   // pairs per thread and batch

// per column: the row tiles its (row-sorted) entries fall into -- first / last column and entry count of every tile.
// (The 256 columns of a workgroup touch a handful of neighbouring tiles on a band: their minima, maxima and counts meet in
//  LDS first -- slots for the 16 tiles from the block's first on -- and leave with one global atomic per slot; 3M global
//  atomics on ~700 addresses took 4.3 ms at config 4.)
__global__ __launch_bounds__(256) void csc_rt_scan(const uint32_t *__restrict__ colptr, const uint32_t *__restrict__ rowind,
                                                   uint32_t ncols, uint32_t RT, uint32_t *__restrict__ cmin,
                                                   uint32_t *__restrict__ cmax, uint32_t *__restrict__ cnt,
                                                   uint32_t *__restrict__ unsorted) {
    constexpr uint32_t kSlots = 16;
    __shared__ uint32_t s_min[kSlots], s_max[kSlots], s_cnt[kSlots], s_base;
    if (threadIdx.x < kSlots) { s_min[threadIdx.x] = 0xffffffffu; s_max[threadIdx.x] = 0u; s_cnt[threadIdx.x] = 0u; }
    if (threadIdx.x == 0) s_base = 0xffffffffu;
    __syncthreads();
    const uint64_t k64 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t k = (uint32_t)k64;
    uint32_t p = 0, p1 = 0;
    if (k64 < ncols) { p = colptr[k]; p1 = colptr[k + 1]; }
    if (p < p1) atomicMin(&s_base, rowind[p] / RT);   // the lowest tile any column of the block starts in
    __syncthreads();
    const uint32_t base = s_base;
    uint32_t prev_tile = 0, prev_row = 0;
    bool any = false;
    while (p < p1) {
        const uint32_t r = rowind[p], t = r / RT;
        if (any && (r < prev_row || t < prev_tile)) atomicOr(unsorted, 1u);   // rows must ascend inside a column
        uint32_t n = 1;
        ++p;
        while (p < p1 && rowind[p] / RT == t && rowind[p] >= rowind[p - 1]) { ++p; ++n; }
        if (t - base < kSlots) {
            atomicMin(&s_min[t - base], k);
            atomicMax(&s_max[t - base], k);
            atomicAdd(&s_cnt[t - base], n);
        } else {
            atomicMin(&cmin[t], k);
            atomicMax(&cmax[t], k);
            atomicAdd(&cnt[t], n);
        }
        prev_tile = t; prev_row = rowind[p - 1]; any = true;
    }
    __syncthreads();
    if (threadIdx.x < kSlots && s_cnt[threadIdx.x]) {
        const uint32_t t = base + threadIdx.x;
        atomicMin(&cmin[t], s_min[threadIdx.x]);
        atomicMax(&cmax[t], s_max[threadIdx.x]);
        atomicAdd(&cnt[t], s_cnt[threadIdx.x]);
    }
}

// the row-tiled copy: one workgroup per tile walks the tile's column window 256 columns at a time; a thread finds its
// column's entries inside the tile's rows (two binary searches in the column's ascending rows), a block scan of the
// lengths gives their place (column order, rows ascending inside a column)
template <typename T>
__global__ __launch_bounds__(256) void csc_rt_fill(const uint32_t *__restrict__ colptr, const uint32_t *__restrict__ rowind,
                                                   const T *__restrict__ values, const uint4 *__restrict__ desc,
                                                   const uint32_t *__restrict__ ptr, T *__restrict__ out_val,
                                                   uint32_t *__restrict__ out_meta, uint32_t *__restrict__ bad) {
    __shared__ uint32_t s_w[4];
    const uint4 d = desc[blockIdx.x];   // {first row, rows, first column of the window, columns}
    const uint32_t r0 = d.x, r1 = d.x + d.y, lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t base = ptr[blockIdx.x];
    for (uint32_t c = 0; c < d.w; c += 256) {   // uniform
        const uint32_t k = d.z + c + threadIdx.x;
        uint32_t a = 0, b = 0;
        if (c + threadIdx.x < d.w) {
            uint32_t lo = colptr[k], hi = colptr[k + 1];
            const uint32_t end = hi;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rowind[mid] < r0) lo = mid + 1; else hi = mid; }
            a = lo;
            hi = end;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rowind[mid] < r1) lo = mid + 1; else hi = mid; }
            b = lo;
        }
        const uint32_t n = b - a;
        uint32_t inc = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, 64);
            if (lane >= (uint32_t)o) inc += t;
        }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        uint32_t pos = base + inc - n;
        for (uint32_t i = 0; i < w; ++i) pos += s_w[i];
        const uint32_t total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
        for (uint32_t p = a; p < b; ++p, ++pos) {
            out_val[pos] = values[p];
            out_meta[pos] = (rowind[p] - r0) | ((k - d.z) << 16);
        }
        base += total;
    }
    if (threadIdx.x == 0 && base != ptr[blockIdx.x + 1]) atomicOr(bad, 1u);
}

template <typename T>
__device__ __forceinline__ void rt_lds_add(T *p, T v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_f64 / ds_add_f32, no return value
}

// desc[t] = {first row, rows, first column of the x window, columns of it}; ptr[t] = the tile's first entry.
// LDS (dynamic): x window T[xcap] | y rows T[RT].
template <typename T>
__global__ __launch_bounds__(kRtThreads, 1) void csc_spmv_rowtiles(const T *__restrict__ vals, const uint32_t *__restrict__ meta,
                                                                   const uint32_t *__restrict__ ptr, const uint4 *__restrict__ desc,
                                                                   const T *__restrict__ x, T *__restrict__ y, uint32_t ntiles,
                                                                   uint32_t per_xcd, uint32_t last_pair, uint32_t xcap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char spal_smem[];
    using pair_t = typename Pair<T>::type;
    using u2_t = __attribute__((ext_vector_type(2))) uint32_t;
    T *xt = reinterpret_cast<T *>(spal_smem);
    T *yw = xt + xcap;
    // neighbouring row tiles read overlapping x windows: each XCD takes a contiguous run of tiles (one L2)
    const uint32_t t = xcd_contiguous_block(blockIdx.x, per_xcd);
    if (t >= ntiles) return;
    const uint4 d = desc[t];                       // block-uniform
    const uint32_t p0 = ptr[t], p1 = ptr[t + 1];   // (independent loads: one round trip for the three)
    constexpr uint32_t U = kRtU, kBatch = 2 * U * kRtThreads;
    uint32_t batch0 = p0 & ~1u;                    // entries in pairs from an even start
    const uint32_t tile_last_pair = min(last_pair, p1 ? (p1 - 1u) & ~1u : 0u);
    pair_t v[U];
    u2_t m[U];
    // the first batch is requested before the x window is staged and the rows are zeroed: both hide behind those loads
    // (unconditional clamped loads: a batch may reach past the tile's last entry and stays inside the allocation)
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
        const uint32_t e = min(batch0 + threadIdx.x * 2 + u * (kRtThreads * 2), last_pair);
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const pair_t *>(vals + e));
        m[u] = __builtin_nontemporal_load(reinterpret_cast<const u2_t *>(meta + e));
    }
    for (uint32_t i = threadIdx.x; i < d.w; i += kRtThreads) xt[i] = x[d.z + i];
    for (uint32_t i = threadIdx.x; i < d.y; i += kRtThreads) yw[i] = T(0);
    __syncthreads();
    if (p0 < p1) {
        // two batches in flight: the next batch's loads are issued before this one's LDS adds (loads return in order,
        // so the adds wait for the older batch only)
        while (true) {
            const uint32_t next0 = batch0 + kBatch;
            const bool more = next0 < p1;   // uniform
            pair_t vn[U];
            u2_t mn[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t e = min(next0 + threadIdx.x * 2 + u * (kRtThreads * 2), tile_last_pair);
                vn[u] = __builtin_nontemporal_load(reinterpret_cast<const pair_t *>(vals + e));
                mn[u] = __builtin_nontemporal_load(reinterpret_cast<const u2_t *>(meta + e));
            }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t e = batch0 + threadIdx.x * 2 + u * (kRtThreads * 2);
                if (e >= p0 && e < p1) rt_lds_add(&yw[m[u].x & 0xffffu], v[u].x * xt[m[u].x >> 16]);
                if (e + 1 >= p0 && e + 1 < p1) rt_lds_add(&yw[m[u].y & 0xffffu], v[u].y * xt[m[u].y >> 16]);
            }
            if (!more) break;
            batch0 = next0;
            for (uint32_t u = 0; u < U; ++u) { v[u] = vn[u]; m[u] = mn[u]; }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < d.y; i += kRtThreads) y[d.x + i] = yw[i];   // every row of the tile, once
}

void csc_rowtiles_free(spal_csc *a) {
    (void)dev_free(a->d_rt_val); a->d_rt_val = nullptr;
    (void)dev_free(a->d_rt_meta); a->d_rt_meta = nullptr;
    (void)dev_free(a->d_rt_ptr); a->d_rt_ptr = nullptr;
    (void)dev_free(a->d_rt_desc); a->d_rt_desc = nullptr;
    a->rowtiles = 0;
    a->rt_rows = a->rt_ntiles = a->rt_xcap = 0;
}

// Builds the row-tiled copy when every tile's window of x fits LDS beside its rows.  SPAL_OK with a->rowtiles = 0 when not.
template <typename T>
static int rowtiles_plan_t(spal_csc *a) {
    csc_rowtiles_free(a);
    if (a->nnz == 0 || a->rowtiles_user == 0 || !a->use_lds) return SPAL_OK;
    const size_t budget = ((size_t)159 * 1024) / sizeof(T);   // elements of x window + rows (one workgroup per CU)
    for (uint32_t RT : {4096u, 2048u, 1024u}) {
        if (a->rt_rows_user && RT != a->rt_rows_user) continue;
        const uint32_t nt = (uint32_t)((a->nrows + RT - 1) / RT);
        uint32_t *d_s = nullptr;   // cmin | cmax | cnt | flag
        SPAL_HIP_TRY(dev_alloc((void **)&d_s, ((size_t)3 * nt + 1) * 4));
        hipError_t e = hipMemsetAsync(d_s, 0xff, (size_t)nt * 4, a->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_s + nt, 0, ((size_t)2 * nt + 1) * 4, a->stream);
        std::vector<uint32_t> s((size_t)3 * nt + 1);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(csc_rt_scan, dim3((uint32_t)((a->ncols + 255) / 256)), dim3(256), 0, a->stream, a->d_colptr,
                               a->d_rowind, (uint32_t)a->ncols, RT, d_s, d_s + nt, d_s + 2 * (size_t)nt, d_s + 3 * (size_t)nt);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(s.data(), d_s, s.size() * 4, hipMemcpyDeviceToHost, a->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(a->stream);
        (void)dev_free(d_s);
        SPAL_HIP_TRY(e);
        if (s[(size_t)3 * nt]) return SPAL_OK;   // rows do not ascend inside some column: the column tiles' business
        uint32_t xcap = 0;
        bool fits = true;
        std::vector<uint4> desc(nt);
        std::vector<uint32_t> ptr((size_t)nt + 1);
        uint64_t run = 0;
        for (uint32_t t = 0; t < nt; ++t) {
            const uint32_t cnt = s[(size_t)2 * nt + t];
            const uint32_t rows = (uint32_t)std::min<uint64_t>(RT, a->nrows - (uint64_t)t * RT);
            const uint32_t c0 = cnt ? s[t] : 0u, span = cnt ? s[(size_t)nt + t] - c0 + 1u : 0u;
            if (span > 65535u || (size_t)span + RT > budget) { fits = false; break; }
            xcap = std::max(xcap, span);
            desc[t] = make_uint4(t * RT, rows, c0, span);
            ptr[t] = (uint32_t)run;
            run += cnt;
        }
        if (!fits) continue;
        ptr[nt] = (uint32_t)run;
        if (run != a->nnz) return fail(SPAL_ERR_HIP, "csc row tiles: %llu entries counted, %llu stored", (unsigned long long)run, (unsigned long long)a->nnz);
        xcap = (xcap + 1u) & ~1u;
        const size_t cap = (size_t)a->nnz + kStreamPad;
        uint32_t *d_bad = nullptr;
        SPAL_HIP_TRY(dev_alloc(&a->d_rt_val, cap * sizeof(T)));
        SPAL_HIP_TRY(dev_alloc((void **)&a->d_rt_meta, cap * 4));
        SPAL_HIP_TRY(dev_alloc((void **)&a->d_rt_ptr, ptr.size() * 4));
        SPAL_HIP_TRY(dev_alloc((void **)&a->d_rt_desc, desc.size() * sizeof(uint4)));
        SPAL_HIP_TRY(dev_alloc((void **)&d_bad, 4));
        e = hipMemsetAsync(d_bad, 0, 4, a->stream);
        if (e == hipSuccess) e = hipMemsetAsync((char *)a->d_rt_val + (size_t)a->nnz * sizeof(T), 0, kStreamPad * sizeof(T), a->stream);
        if (e == hipSuccess) e = hipMemsetAsync(a->d_rt_meta + a->nnz, 0, kStreamPad * 4, a->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(a->d_rt_ptr, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice, a->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(a->d_rt_desc, desc.data(), desc.size() * sizeof(uint4), hipMemcpyHostToDevice, a->stream);
        uint32_t bad = 0;
        if (e == hipSuccess) {
            hipLaunchKernelGGL(csc_rt_fill<T>, dim3(nt), dim3(256), 0, a->stream, a->d_colptr, a->d_rowind, (const T *)a->d_values,
                               a->d_rt_desc, a->d_rt_ptr, (T *)a->d_rt_val, a->d_rt_meta, d_bad);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, a->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(a->stream);   // desc / ptr go out of scope
        (void)dev_free(d_bad);
        if (e != hipSuccess || bad) {
            csc_rowtiles_free(a);
            if (e != hipSuccess) return fail(SPAL_ERR_HIP, "csc row tiles: %s", hipGetErrorString(e));
            return fail(SPAL_ERR_HIP, "csc row tiles: a tile's entries do not add up to its count");
        }
        a->rowtiles = 1;
        a->rt_rows = RT;
        a->rt_ntiles = nt;
        a->rt_xcap = xcap;
        return SPAL_OK;
    }
    return SPAL_OK;
}

// The row tiles are an OPTIONAL second copy of the entries (12 bytes each) that only the scatter path reads: built when
// kernel 1 is selected (spal_csc_set_option "kernel" = 1), not for the default transposed route; a failure to build it --
// out of memory, an inconsistent count -- means "does not qualify": the copy is freed, the error cleared and the column
// tiles run.  `rowtiles_failed` in spal_csc_describe says so.
int csc_rowtiles_plan(spal_csc *a) {
    a->rowtiles_failed = 0;
    if (a->kernel != 1) { csc_rowtiles_free(a); return SPAL_OK; }
    const int st = a->elem_size == 8 ? rowtiles_plan_t<double>(a) : rowtiles_plan_t<float>(a);
    if (st != SPAL_OK) {
        csc_rowtiles_free(a);
        (void)hipGetLastError();
        a->rowtiles_failed = 1;
    }
    return SPAL_OK;
}

template <typename T>
static hipError_t launch_rowtiles_t(const spal_csc *a, const void *x, void *y, hipStream_t st) {
    const size_t lds = ((size_t)a->rt_xcap + a->rt_rows) * sizeof(T);
    auto kern = csc_spmv_rowtiles<T>;
    static std::atomic<uint64_t> configured{0};
    const uint64_t bit = 1ull << (a->device & 63);
    if (lds > 48 * 1024 && !(configured.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured.fetch_or(bit, std::memory_order_relaxed);
    }
    const uint32_t per_xcd = (a->rt_ntiles + 7) / 8;
    hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(kRtThreads), lds, st, (const T *)a->d_rt_val, a->d_rt_meta, a->d_rt_ptr,
                       a->d_rt_desc, (const T *)x, (T *)y, a->rt_ntiles, per_xcd,
                       (uint32_t)(((a->nnz + kStreamPad) & ~(uint64_t)1) - 2), a->rt_xcap);
    return hipGetLastError();
}

hipError_t launch_csc_rowtiles(const spal_csc *a, const void *x, void *y, hipStream_t st) {
    return a->elem_size == 8 ? launch_rowtiles_t<double>(a, x, y, st) : launch_rowtiles_t<float>(a, x, y, st);
}

}  // namespace spal
