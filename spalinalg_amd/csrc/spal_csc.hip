// spal_csc.hip -- CSC handle and y = A*x by atomic scatter.
//
// Contract (reference src/csc/ops/mul.rs:26-46): for
// every stored entry (i, k): y[i] += values[p] * x[k]; rows never touched
// give 0.0.  On the GPU the adds into one y[i] arrive in no fixed order
// (hardware f64/f32 atomics), so parity with the sequential CPU order is to
// rounding (<= 1e-10 relative, tested), not bitwise.
//
// Two scatter paths in one launch, chosen per SUPER-TILE of 512 columns:
//   LDS-privatised : the rows a super-tile touches form a window
//                    [rmin, rmax]; when it fits LDS the adds go to an LDS copy
//                    of that window (ds_add_f64: conflicts cost cycles, not
//                    memory round trips) and the window is flushed once with
//                    contiguous global atomics (full-rate shape: 256 B per
//                    wave instruction).  Entries are streamed coalesced with a
//                    packed 32-bit (row - rmin | (col - k0) << 16) per entry.
//   global scatter : one global atomic per entry (windows that do not fit).
#include "csr_kernels.hpp"
#include "spal_internal.hpp"

namespace spal {

#ifndef SPAL_CSC_U
#define SPAL_CSC_U 2
#endif
#ifndef SPAL_CSC_BLOCK
#define SPAL_CSC_BLOCK 1024
#endif
#ifndef SPAL_CSC_FENCES
#define SPAL_CSC_FENCES 0   // 1: release / acquire fences around the hand-off's flag (see csc_spmv_scatter)
#endif

constexpr int kCscBlock = SPAL_CSC_BLOCK;      // threads of the scatter kernel
// Columns per super-tile: the widest of 4096 / 2048 / 1024 whose row windows fit LDS (csc_plan_build).  Wider
// super-tiles flush fewer window rows per column -- the contiguous global atomics of the flush are what bounds
// config 4: 1024 columns 39.7 MB of atomics 53.5 us, 2048 23.8 MB 44.6 us, 4096 15.9 MB 42.7 us
// (profiles/r02/csc_scatter_column_tiles.txt) -- and 245 workgroups are one round of the device.
constexpr int kCscColsMax = 4096;   // (the x tile of the widest form: 32 KiB of f64)
static_assert(kCscColsMax % kCscBlock == 0, "whole x elements per thread");
// LDS y window budget (+ 8 KiB x tile): two 1024-thread workgroups per CU either way; a band's
// clamped edge needs more rows than its interior (config 4: 6717 against 5120), and ONE super-tile
// left to the global-atomic path kept the whole launch busy (47 us -> 78 us)
// (x tile + window <= 80 KiB: 70 KiB of window beside 1024 columns of f64 x, 64 KiB beside 2048)
// LDS of a workgroup: x tile + y window.  1024 / 2048 columns: 80 KiB in all, two workgroups per CU; 4096 columns:
// 32 KiB of x + 94 KiB of window, one.
constexpr uint32_t csc_window_bytes(int cols) { return (cols >= 4096 ? 94u : cols >= 2048 ? 64u : 70u) * 1024u; }
constexpr uint32_t kCscModeGlobal = 0, kCscModeLds = 1;

// ---- plan-time kernels ---------------------------------------------------------
// (rowind is strictly increasing inside a column, src/csc.rs:152-156: the first
// and last entry of a column bound its rows)
__global__ __launch_bounds__(256) void csc_block_windows(const uint32_t *__restrict__ colptr,
                                                         const uint32_t *__restrict__ rowind,
                                                         uint32_t ncols, uint32_t cols, uint2 *__restrict__ out) {
    __shared__ uint32_t s_min, s_max;
    if (threadIdx.x == 0) { s_min = 0xffffffffu; s_max = 0u; }
    __syncthreads();
    const uint32_t k0 = blockIdx.x * cols, k1 = min(k0 + cols, ncols);
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (uint32_t k = k0 + threadIdx.x; k < k1; k += 256) {
        const uint32_t a0 = colptr[k], a1 = colptr[k + 1];
        if (a0 < a1) {
            lo = min(lo, rowind[a0]);
            hi = max(hi, rowind[a1 - 1] + 1u);
        }
    }
    atomicMin(&s_min, lo);
    atomicMax(&s_max, hi);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = make_uint2(s_min, s_max);
}

// *differs |= 1 when some column does not hold exactly `len` entries
__global__ __launch_bounds__(256) void csc_uniform_check(const uint32_t *__restrict__ colptr, uint32_t ncols, uint32_t len,
                                                         uint32_t *__restrict__ differs) {
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool bad = k < ncols && colptr[k + 1] - colptr[k] != len;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(differs, 1u);
}

// meta[p] = (row - rbase) | (col - k0) << 16 for LDS-mode super-tiles
__global__ __launch_bounds__(256) void csc_encode_meta(const uint32_t *__restrict__ colptr,
                                                       const uint32_t *__restrict__ rowind,
                                                       const uint4 *__restrict__ desc,
                                                       uint32_t *__restrict__ meta, uint32_t ncols, uint32_t cols) {
    const uint4 d = desc[blockIdx.x];
    if (d.z != kCscModeLds) return;
    const uint32_t k0 = blockIdx.x * cols, k1 = min(k0 + cols, ncols);
    for (uint32_t k = k0 + threadIdx.x; k < k1; k += 256)
        for (uint32_t p = colptr[k]; p < colptr[k + 1]; ++p)
            meta[p] = (rowind[p] - d.x) | ((k - k0) << 16);
}

// ---- the scatter kernel ------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void lds_add(T *p, T v) {
    // relaxed, workgroup scope: ds_add_f64 / ds_add_f32, no return value
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// desc[b] = {window base row, window length, mode, 0}
template <typename T, int kCscCols>
__global__ __launch_bounds__(kCscBlock, kCscBlock >= 1024 ? 1 : 2) void csc_spmv_scatter(
    const uint32_t *__restrict__ colptr, const uint32_t *__restrict__ rowind,
    const uint32_t *__restrict__ meta, const T *__restrict__ vals, const T *__restrict__ x,
    T *__restrict__ y, const uint4 *__restrict__ desc, uint32_t ncols, uint32_t nblocks,
    uint32_t per_xcd, uint32_t last_pair, T *__restrict__ windows, const uint32_t *__restrict__ prev_hi,
    uint32_t *__restrict__ flags, uint32_t epoch, uint32_t nrows, uint32_t ticket_base, uint32_t use_ticket,
    uint32_t spin_bound, uint32_t *__restrict__ gave_up, uint32_t ulen) {
    extern __shared__ __attribute__((aligned(16))) unsigned char spal_smem[];
    using pair_t = typename Pair<T>::type;
    using u2_t = __attribute__((ext_vector_type(2))) uint32_t;
    T *xt = reinterpret_cast<T *>(spal_smem);  // x of the super-tile's columns
    T *yw = xt + kCscCols;                     // y window accumulators

    // Neighbour hand-off: super-tile b waits for b - 1, which must therefore have STARTED before it.  The super-tile a
    // workgroup takes is its ticket -- one atomicAdd on flags[nblocks + 1] per workgroup, ticket_base = the counter's
    // value when this launch began (launches of one handle are chained) -- i.e. the order in which workgroups actually
    // start, not their blockIdx: the holder of b started after the holders of 0 ... b - 1, which are resident or
    // done, so the wait below always ends, whatever order the dispatcher takes the workgroups in.  (use_ticket = 0:
    // blockIdx order, option "ticket"; enough when all workgroups of the launch are resident together.)
    // Without the hand-off each XCD takes a contiguous run of super-tiles.
    uint32_t b;
    if (prev_hi) {
        b = blockIdx.x;
        if (use_ticket) {
            // (through the first word of the dynamic LDS: the kernel may use all 160 KiB of it, a static variable would not fit)
            uint32_t *s_ticket = reinterpret_cast<uint32_t *>(spal_smem);
            if (threadIdx.x == 0) *s_ticket = atomicAdd(&flags[nblocks + 1], 1u) - ticket_base;
            __syncthreads();
            b = *s_ticket;
            __syncthreads();   // before the x tile is written there
        }
    } else {
        b = xcd_contiguous_block(blockIdx.x, per_xcd);
    }
    if (b >= nblocks) return;
    const uint32_t k0 = b * kCscCols, k1 = min(k0 + (uint32_t)kCscCols, ncols);
    // desc and the two column pointers are independent loads: one round trip for the three
    const uint4 d = desc[b];  // block-uniform
    // (every column of the matrix holds ulen - 1 entries: the column pointers are arithmetic, and the first batch of
    //  entries can be requested without waiting for them -- one memory round trip less at the start of a workgroup
    //  that lives for a dozen)
    const uint32_t p0 = ulen ? k0 * (ulen - 1u) : colptr[k0], p1 = ulen ? k1 * (ulen - 1u) : colptr[k1];  // uniform

    if (d.z == kCscModeLds) {
        // entries in pairs from an even start; a batch = U pairs per thread.  The FIRST batch and this
        // thread's element of the x tile are requested before the window is zeroed and before the
        // barrier: zeroing and staging hide behind those loads.
        constexpr uint32_t U = SPAL_CSC_U;
        constexpr uint32_t kBatch = 2 * U * kCscBlock;
        static_assert(kCscCols % kCscBlock == 0 || kCscCols < kCscBlock, "whole x elements per thread");
        constexpr uint32_t XPT = kCscCols > kCscBlock ? kCscCols / kCscBlock : 1;   // x elements per thread
        uint32_t batch0 = p0 & ~1u;   // uniform: first entry of the current batch
        const uint32_t tile_last_pair = min(last_pair, p1 ? (p1 - 1u) & ~1u : 0u);   // pair holding the super-tile's last entry
        pair_t v[U];
        u2_t m[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            // unconditional loads; a batch may reach past the super-tile's last entry: stay inside
            // the allocation (last_pair = last even index of the padded arrays)
            const uint32_t e = min(batch0 + threadIdx.x * 2 + u * (kCscBlock * 2), last_pair);
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const pair_t *>(vals + e));
            m[u] = __builtin_nontemporal_load(reinterpret_cast<const u2_t *>(meta + e));
        }
        T xk[XPT];
#pragma unroll
        for (uint32_t q = 0; q < XPT; ++q) xk[q] = x[min(k0 + threadIdx.x + q * kCscBlock, k1 - 1u)];
        for (uint32_t i = threadIdx.x; i < d.y; i += kCscBlock) yw[i] = T(0);
#pragma unroll
        for (uint32_t q = 0; q < XPT; ++q)
            if (threadIdx.x + q * kCscBlock < k1 - k0) xt[threadIdx.x + q * kCscBlock] = xk[q];
        __syncthreads();
        // Two batches in flight: the next batch's loads are issued before this one's LDS adds (loads return in
        // order, so the adds wait for the older batch only) -- with one batch the wave sat out a full memory
        // round trip per batch, six times per super-tile of config 4.
        while (true) {
            const uint32_t next0 = batch0 + kBatch;
            const bool more = next0 < p1;   // uniform
            pair_t vn[U];
            u2_t mn[U];
            // (unconditional: loads under a uniform branch make the compiler wait for ALL loads before the adds;
            //  lanes past the super-tile's last pair re-read that pair: one line, no traffic)
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t e = min(next0 + threadIdx.x * 2 + u * (kCscBlock * 2), tile_last_pair);
                vn[u] = __builtin_nontemporal_load(reinterpret_cast<const pair_t *>(vals + e));
                mn[u] = __builtin_nontemporal_load(reinterpret_cast<const u2_t *>(meta + e));
            }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t e = batch0 + threadIdx.x * 2 + u * (kCscBlock * 2);
                if (e >= p0 && e < p1) lds_add(&yw[m[u].x & 0xffffu], v[u].x * xt[m[u].x >> 16]);
                if (e + 1 >= p0 && e + 1 < p1) lds_add(&yw[m[u].y & 0xffffu], v[u].y * xt[m[u].y >> 16]);
            }
            if (!more) break;
            batch0 = next0;
            for (uint32_t u = 0; u < U; ++u) { v[u] = vn[u]; m[u] = mn[u]; }
        }
        __syncthreads();
        if (windows) {
            // two-phase flush: the window goes to this super-tile's slot (plain coalesced
            // stores); csc_window_reduce adds the overlapping windows row by row
            T *slot = windows + d.w;
            for (uint32_t i = threadIdx.x; i < d.y; i += kCscBlock) slot[i] = yw[i];
            return;
        }
        if (prev_hi) {
            // Neighbour hand-off (plan: windows ascend, only adjacent super-tiles overlap).  Rows from the end of
            // the previous window on are this super-tile's own: STORED (written through, `sc1`), with zeros for
            // rows no window covers; then a flag.  The rows shared with the previous super-tile are updated after
            // ITS flag: y[r] = (previous tile's sum) + (this tile's) -- columns ascending, no atomics, no memset.
            const uint32_t lo = d.x, hi = d.x + d.y, ph = prev_hi[b];       // ph <= hi; ph = 0 for b = 0
            const uint32_t own0 = max(lo, ph);
            for (uint32_t r = min(ph, lo) + threadIdx.x; r < lo; r += kCscBlock) y[r] = T(0);   // gap before the window
            for (uint32_t r = own0 + threadIdx.x; r < hi; r += kCscBlock)
                __hip_atomic_store(&y[r], yw[r - lo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b + 1 == nblocks)
                for (uint32_t r = hi + threadIdx.x; r < nrows; r += kCscBlock) y[r] = T(0);     // rows after the last window
            // Publication.  Every access that takes part is an agent-scope atomic (sc1): the stores of y above are
            // written through to the device's point of coherence and acknowledged from there, and `s_waitcnt vmcnt(0)`
            // holds every storing wave until its acknowledgements are in -- the rows are PERFORMED at agent scope before
            // the barrier, the flag is stored after it.  The consumer reads the flag, then (control dependency + the
            // barrier) the rows, again with sc1 loads, which are served from the point of coherence, never from a line
            // its XCD's L2 happens to hold.  That is the ISA-level contract (gfx942 / gfx950 memory model: agent-scope
            // atomics bypass the non-coherent levels); what release / acquire would ADD is a write-back of the L2's
            // dirty non-atomic lines (`buffer_wbl2 sc1`) and an invalidate (`buffer_inv sc1`) -- there are no
            // non-atomic lines in this exchange.  Measured (-DSPAL_CSC_FENCES=1: flag stored with release, acquire fence
            // after the spin): 44.9 instead of 39.0 us per product at config 4, +15 % (profiles/r03/csc_handoff.txt),
            // hence off by default; tests/test_gpu_csc_coo.py runs 600 checked products back to back either way.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#if SPAL_CSC_FENCES
            if (threadIdx.x == 0) __hip_atomic_store(&flags[b], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
            if (threadIdx.x == 0) __hip_atomic_store(&flags[b], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            if (ph > lo) {   // uniform: rows [lo, ph) also belong to super-tile b - 1, which stores them
                if (threadIdx.x == 0) {
                    uint32_t spins = 0;
                    while (__hip_atomic_load(&flags[b - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                        if (++spins > spin_bound) {   // a backstop, so that the wave ends whatever happens: the host is told
                            __hip_atomic_store(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (host memory)
                            break;
                        }
                        __builtin_amdgcn_s_sleep(8);
                    }
#if SPAL_CSC_FENCES
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
                }
                __syncthreads();
                for (uint32_t r = lo + threadIdx.x; r < min(ph, hi); r += kCscBlock) {
                    const T before = __hip_atomic_load(&y[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    y[r] = before + yw[r - lo];
                }
            }
            return;
        }
        // flush: contiguous, one atomic per touched row (adding 0.0 changes nothing)
        for (uint32_t i = threadIdx.x; i < d.y; i += kCscBlock) {
            const T s = yw[i];
            if (s != T(0) || s != s) atomicAdd(&y[d.x + i], s);
        }
        return;
    }

    // Super-tiles whose row window does not fit LDS (at a band's clamped edges, or scattered rows):
    // one global atomic per entry, entry-parallel.  The tile's column pointers and x go to LDS;
    // thread t takes entries p0 + t, p0 + t + 1024, ...: coalesced loads of (row, value), all of a
    // batch in flight together, the entry's column by binary search in the LDS column pointers.
    // (A lanes-per-column walk costs three dependent memory round trips per 64 columns; on config 4
    // the two edge super-tiles alone kept the kernel busy for 75 us while all others took 47.)
    {
        uint32_t *cp = reinterpret_cast<uint32_t *>(yw);   // k1 - k0 + 1 column pointers
        const uint32_t nk = k1 - k0;
        for (uint32_t i = threadIdx.x; i <= nk; i += kCscBlock) cp[i] = colptr[k0 + i];
        for (uint32_t i = threadIdx.x; i < nk; i += kCscBlock) xt[i] = x[k0 + i];
        __syncthreads();
        constexpr uint32_t U = 4;   // (p0 = cp[0], p1 = cp[nk])
        for (uint32_t base = p0 + threadIdx.x; base < p1; base += kCscBlock * U) {
            uint32_t row[U];
            T val[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t e = min(base + u * kCscBlock, p1 - 1u);   // in bounds; the extra lanes re-read
                row[u] = load_stream(rowind + e);
                val[u] = load_stream(vals + e);
            }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t e = base + u * kCscBlock;
                if (e < p1) {
                    uint32_t lo = 0, hi = nk;   // largest c with cp[c] <= e (empty columns share a pointer with their successor)
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (cp[mid] <= e) lo = mid; else hi = mid;
                    }
                    atomicAdd(&y[row[u]], val[u] * xt[lo]);  // -munsafe-fp-atomics: global_atomic_add_f64 / _f32
                }
            }
        }
    }
}

// Second phase of the two-phase flush: one workgroup per chunk of kCscChunk rows adds
// the windows of the super-tiles that overlap the chunk, in ascending super-tile
// (= column) order, and writes y (ASSIGN: no super-tile went the global-atomic way,
// y was not zeroed) or adds to it (the kernel before has finished: plain update).
constexpr uint32_t kCscChunk = 1024;
template <typename T, bool ASSIGN>
__global__ __launch_bounds__(256) void csc_window_reduce(const T *__restrict__ windows,
                                                         const uint4 *__restrict__ desc,
                                                         const uint32_t *__restrict__ chunk_ptr,
                                                         const uint32_t *__restrict__ chunk_blk,
                                                         T *__restrict__ y, uint32_t nrows) {
    const uint32_t c0 = chunk_ptr[blockIdx.x], c1 = chunk_ptr[blockIdx.x + 1];  // uniform
    if (!ASSIGN && c0 == c1) return;
    const uint32_t r0 = blockIdx.x * kCscChunk;
    T acc[kCscChunk / 256];
#pragma unroll
    for (uint32_t k = 0; k < kCscChunk / 256; ++k) acc[k] = T(0);
    for (uint32_t c = c0; c < c1; ++c) {
        const uint4 d = desc[chunk_blk[c]];  // uniform: {window base, length, mode, slot offset}
#pragma unroll
        for (uint32_t k = 0; k < kCscChunk / 256; ++k) {
            const uint32_t i = r0 + threadIdx.x + 256u * k;
            if (i >= d.x && i - d.x < d.y) acc[k] = acc[k] + windows[d.w + (i - d.x)];
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < kCscChunk / 256; ++k) {
        const uint32_t i = r0 + threadIdx.x + 256u * k;
        if (i < nrows) {
            if (ASSIGN) y[i] = acc[k];
            else y[i] = y[i] + acc[k];
        }
    }
}

static int pick_lanes_csc(double mean) {
    int L = 2;
    while (L < 64 && (double)L < mean) L <<= 1;
    return L;
}

// workgroups of csc_spmv_scatter<T, COLS> one CU holds at once with `lds` bytes of dynamic LDS -- asked of the runtime
// (registers, LDS and wave slots of the compiled kernel), not derived from LDS alone; 0 when it cannot say
template <typename T, int COLS>
static int csc_scatter_per_cu(size_t lds) {
    auto kern = csc_spmv_scatter<T, COLS>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)kern, (int)kCscBlock, lds) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
static int csc_scatter_per_cu(const spal_csc *a, size_t lds) {
    const bool d = a->elem_size == 8;
    switch (a->cols_per_block) {
        case 4096: return d ? csc_scatter_per_cu<double, 4096>(lds) : csc_scatter_per_cu<float, 4096>(lds);
        case 2048: return d ? csc_scatter_per_cu<double, 2048>(lds) : csc_scatter_per_cu<float, 2048>(lds);
        default: return d ? csc_scatter_per_cu<double, 1024>(lds) : csc_scatter_per_cu<float, 1024>(lds);
    }
}

template <typename T, int COLS>
static hipError_t csc_launch_c(const spal_csc *a, const void *x, void *y, hipStream_t st, uint32_t epoch, uint32_t ticket_base) {
    const uint32_t per_xcd = (a->nblocks + 7) / 8;
    // x tile + the y window; global-mode super-tiles keep their column pointers where the window would be
    const size_t lds = std::max(((size_t)COLS + a->lds_entries) * sizeof(T),
                                (size_t)COLS * sizeof(T) + ((size_t)COLS + 2) * sizeof(uint32_t));
    auto kern = csc_spmv_scatter<T, COLS>;
    static std::atomic<uint64_t> configured{0};
    if (lds > 48 * 1024) {
        const uint64_t bit = 1ull << (a->device & 63);
        if (!(configured.load(std::memory_order_relaxed) & bit)) {
            hipError_t e = hipFuncSetAttribute((const void *)kern,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            configured.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(kCscBlock), lds, st, a->d_colptr, a->d_rowind,
                       a->d_meta, (const T *)a->d_values, (const T *)x, (T *)y, a->d_desc,
                       (uint32_t)a->ncols, a->nblocks, per_xcd,
                       (uint32_t)(((a->nnz + kStreamPad) & ~(uint64_t)1) - 2),
                       (a->flush == 1 && a->d_windows) ? (T *)a->d_windows : (T *)nullptr,
                       epoch ? a->d_prev_hi : (const uint32_t *)nullptr, a->d_flags, epoch, (uint32_t)a->nrows,
                       ticket_base, (uint32_t)(a->use_ticket < 0 ? a->ticket_auto : a->use_ticket), a->spin_bound, a->d_gave_up,
                       a->uniform_cols);
    return hipGetLastError();
}

template <typename T>
static hipError_t csc_launch_t(spal_csc *a, const void *x, void *y, hipStream_t st) {
    // row tiles where the plan built them: a workgroup owns rows of y outright -- no memset, no hand-off, no launch chain
    if (a->rowtiles && a->rowtiles_user != 0 && a->flush == 0 && a->nnz) return launch_csc_rowtiles(a, x, y, st);
    const bool two_phase = a->flush == 1 && a->d_windows;
    const bool assign = two_phase && a->all_lds;   // the reduce writes every row of y: no memset
    hipError_t e = hipSuccess;
    // Neighbour hand-off: the super-tiles of one launch talk through the handle's flags, so launches of one handle
    // are chained (each waits for the event of the one before, whatever its stream).  Not while the stream is being
    // captured into a graph (an event from outside the capture cannot be waited for): global atomics then.
    bool ordered = a->flush == 0 && a->ordered && a->nnz != 0;
    if (ordered) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) ordered = false;
    }
    std::unique_lock<std::mutex> chain(a->mu_launch, std::defer_lock);
    uint32_t epoch = 0, ticket_base = 0;
    if (ordered) {
        chain.lock();
        // A super-tile of an EARLIER launch hit the hand-off's spin bound (the backstop; the kernel reports it through
        // host memory, so looking costs nothing): that product's y is invalid.  The caller is told here, once, and the
        // handle flushes with global atomics from now on.  (The host-vector path sees it after its own
        // synchronisation and repeats the product itself, csc_spmv_host.)
        if (a->h_gave_up && __atomic_load_n(a->h_gave_up, __ATOMIC_RELAXED)) {
            __atomic_store_n(a->h_gave_up, 0u, __ATOMIC_RELAXED);
            a->ordered = 0;
            a->handoff_timeouts++;
            return hipErrorLaunchTimeOut;
        }
        // launches of one handle share its flags: each must run after the one before.  On ONE stream that is the
        // stream's order and costs nothing; only a launch on another stream than the last waits for an event, recorded
        // now on that last stream (it covers everything submitted there so far).  (An event wait + record around
        // every launch cost ~3 us of a 38 us product in a loop.)
        if (a->last_stream_valid && a->last_stream != st) {
            if (!a->ev_last) {
                e = hipEventCreateWithFlags(&a->ev_last, hipEventDisableTiming);
                if (e != hipSuccess) return e;
            }
            e = hipEventRecord(a->ev_last, a->last_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(st, a->ev_last, 0);
            if (e != hipSuccess) {   // (the caller may have destroyed that stream meanwhile: everything it held has then run or is flushed here)
                (void)hipGetLastError();
                e = hipDeviceSynchronize();
            }
            if (e != hipSuccess) return e;
        }
        a->last_stream = st;
        a->last_stream_valid = 1;
        if (++a->epoch == 0) {   // (wrapped after 2^32 launches: the flags and the ticket counter start over)
            e = hipMemsetAsync(a->d_flags, 0, ((size_t)a->nblocks + 2) * 4, st);
            if (e != hipSuccess) return e;
            a->epoch = 1;
            a->ticket_next = 0;
        }
        epoch = a->epoch;
        ticket_base = a->ticket_next;               // every workgroup of the grid takes one ticket
        a->ticket_next += ((a->nblocks + 7) / 8) * 8;
    }
    if (!ordered && (!assign || a->nnz == 0)) {
        // (a kernel, not hipMemsetAsync: as a node of a captured graph the memset zeroed every other element from
        //  the second replay on -- ROCm 7.2, tools/lab.py csc_capture)
        hipLaunchKernelGGL(fill_zero<T>, dim3((uint32_t)((a->nrows + 255) / 256)), dim3(256), 0, st, (T *)y, a->nrows);
        e = hipGetLastError();
    }
    if (e != hipSuccess || a->nnz == 0) return e;
    switch (a->cols_per_block) {
        case 1024: e = csc_launch_c<T, 1024>(a, x, y, st, epoch, ticket_base); break;
        case 2048: e = csc_launch_c<T, 2048>(a, x, y, st, epoch, ticket_base); break;
        case 4096: e = csc_launch_c<T, 4096>(a, x, y, st, epoch, ticket_base); break;
        default: return hipErrorInvalidValue;
    }
    if (ordered) return e;
    if (e != hipSuccess || !two_phase) return e;
    if (assign)
        hipLaunchKernelGGL((csc_window_reduce<T, true>), dim3(a->nchunks), dim3(256), 0, st,
                           (const T *)a->d_windows, a->d_desc, a->d_chunk_ptr, a->d_chunk_blk, (T *)y,
                           (uint32_t)a->nrows);
    else
        hipLaunchKernelGGL((csc_window_reduce<T, false>), dim3(a->nchunks), dim3(256), 0, st,
                           (const T *)a->d_windows, a->d_desc, a->d_chunk_ptr, a->d_chunk_blk, (T *)y,
                           (uint32_t)a->nrows);
    return hipGetLastError();
}

static int csc_ensure_csr(spal_csc *a);

static int csc_launch(spal_csc *a, const void *x, void *y, hipStream_t st) {
    if (a->kernel == 2) {  // the same matrix as CSR (built once), stream / vector CSR kernel
        SPAL_TRY(csc_ensure_csr(a));
        return csr_launch(a->as_csr, x, y, st);
    }
    hipError_t e = a->elem_size == 8 ? csc_launch_t<double>(a, x, y, st)
                                     : csc_launch_t<float>(a, x, y, st);
    if (e == hipErrorLaunchTimeOut)
        return fail(SPAL_ERR_HIP, "csc spmv: an earlier product of this handle gave up waiting in the neighbour hand-off "
                    "(spin bound reached): that product's y is invalid; nothing was launched now, and the handle flushes "
                    "with global atomics from here on");
    if (e != hipSuccess) return fail(SPAL_ERR_HIP, "csc spmv launch failed: %s", hipGetErrorString(e));
    return SPAL_OK;
}

// Per-super-tile windows and modes; packed metadata for the LDS mode.
static int csc_plan_build(spal_csc *a) {
    if (a->d_desc) { SPAL_HIP_TRY(dev_free(a->d_desc)); a->d_desc = nullptr; }
    if (a->d_windows) { SPAL_HIP_TRY(dev_free(a->d_windows)); a->d_windows = nullptr; }
    if (a->d_chunk_ptr) { SPAL_HIP_TRY(dev_free(a->d_chunk_ptr)); a->d_chunk_ptr = nullptr; }
    if (a->d_chunk_blk) { SPAL_HIP_TRY(dev_free(a->d_chunk_blk)); a->d_chunk_blk = nullptr; }
    if (a->d_prev_hi) { SPAL_HIP_TRY(dev_free(a->d_prev_hi)); a->d_prev_hi = nullptr; }
    if (a->d_flags) { SPAL_HIP_TRY(dev_free(a->d_flags)); a->d_flags = nullptr; }
    a->ordered = 0;
    a->epoch = 0;
    a->ticket_next = 0;
    a->ticket_auto = 0;
    a->spin_bound = 1u << 22;
    if (const char *e = getenv("SPAL_CSC_HANDOFF_SPINS")) a->spin_bound = (uint32_t)strtoul(e, nullptr, 10);   // (tests: 0 forces the backstop)
    a->windows_entries = 0;
    a->all_lds = 0;
    a->nchunks = 0;
    a->lds_entries = 0;
    a->lds_col_fraction = 0.0;
    a->cols_per_block = a->user_cols ? a->user_cols : 1024;
    a->nblocks = (uint32_t)((a->ncols + a->cols_per_block - 1) / a->cols_per_block);
    a->uniform_cols = 0;
    if (a->nnz && a->nnz % a->ncols == 0) {   // every column the same length?  (then the kernel computes the column pointers)
        const uint32_t len = (uint32_t)(a->nnz / a->ncols);
        uint32_t *d_f = nullptr, f = 1;
        SPAL_HIP_TRY(dev_alloc((void **)&d_f, 4));
        hipError_t e = hipMemsetAsync(d_f, 0, 4, a->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(csc_uniform_check, dim3((uint32_t)((a->ncols + 255) / 256)), dim3(256), 0, a->stream,
                               a->d_colptr, (uint32_t)a->ncols, len, d_f);
            e = hipMemcpyAsync(&f, d_f, 4, hipMemcpyDeviceToHost, a->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(a->stream);
        (void)dev_free(d_f);
        SPAL_HIP_TRY(e);
        if (!f) a->uniform_cols = len + 1;
    }
    std::vector<uint4> desc(a->nblocks, make_uint4(0, 0, kCscModeGlobal, 0));
    if (a->nnz && a->use_lds) {
        // row windows per 1024 columns (one device pass); wider super-tiles are unions of those
        const uint32_t nb1 = (uint32_t)((a->ncols + 1023) / 1024);
        uint2 *d_win = nullptr;
        SPAL_HIP_TRY(dev_alloc((void **)&d_win, (size_t)nb1 * sizeof(uint2)));
        hipLaunchKernelGGL(csc_block_windows, dim3(nb1), dim3(256), 0, a->stream, a->d_colptr,
                           a->d_rowind, (uint32_t)a->ncols, 1024u, d_win);
        std::vector<uint2> win1(nb1);
        hipError_t e = hipMemcpyAsync(win1.data(), d_win, (size_t)nb1 * sizeof(uint2),
                                      hipMemcpyDeviceToHost, a->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(a->stream);
        (void)dev_free(d_win);
        SPAL_HIP_TRY(e);
        auto windows_for = [&](int cols, std::vector<uint2> &win, uint64_t &fit_cols) {
            const uint32_t k = (uint32_t)cols / 1024u, nb = (uint32_t)((a->ncols + cols - 1) / cols);
            const uint32_t budget = std::min<uint32_t>(csc_window_bytes(cols) / (uint32_t)a->elem_size, 65536u);
            win.assign(nb, make_uint2(0xffffffffu, 0u));
            fit_cols = 0;
            for (uint32_t b = 0; b < nb; ++b) {
                for (uint32_t j = b * k; j < std::min(nb1, (b + 1) * k); ++j)
                    if (win1[j].y) { win[b].x = std::min(win[b].x, win1[j].x); win[b].y = std::max(win[b].y, win1[j].y); }
                if (win[b].y == 0 || win[b].y - win[b].x <= budget)
                    fit_cols += std::min<uint64_t>((uint64_t)cols, a->ncols - (uint64_t)b * cols);
            }
        };
        // the widest super-tile whose windows fit as well as the 1024-column ones do
        std::vector<uint2> win;
        uint64_t fit = 0, fit1024 = 0;
        windows_for(1024, win, fit1024);
        if (!a->user_cols) {
            for (int cols : {4096, 2048}) {
                std::vector<uint2> w;
                windows_for(cols, w, fit);
                if (fit >= fit1024) { a->cols_per_block = cols; win.swap(w); break; }
            }
        } else {
            windows_for(a->cols_per_block, win, fit);
        }
        a->nblocks = (uint32_t)((a->ncols + a->cols_per_block - 1) / a->cols_per_block);
        desc.assign(a->nblocks, make_uint4(0, 0, kCscModeGlobal, 0));
        const uint32_t kCscCols = (uint32_t)a->cols_per_block;
        const uint32_t budget = std::min<uint32_t>(csc_window_bytes(a->cols_per_block) / (uint32_t)a->elem_size, 65536u);
        uint64_t cols_lds = 0, slot = 0;
        bool all_lds = true;
        a->nchunks = (uint32_t)((a->nrows + kCscChunk - 1) / kCscChunk);
        std::vector<uint32_t> cover_count(a->nchunks + 1, 0);
        for (uint32_t b = 0; b < a->nblocks; ++b) {
            const uint2 w = win[b];
            if (w.y == 0) continue;  // no entries: the global path finds nothing to do
            const uint32_t len = w.y - w.x;
            if (len <= budget && slot + len < 0xffffffffull) {
                desc[b] = make_uint4(w.x, len, kCscModeLds, (uint32_t)slot);
                slot += (len + 1) & ~1ull;   // slots start on even elements
                a->lds_entries = std::max(a->lds_entries, len);
                cols_lds += std::min<uint64_t>(kCscCols, a->ncols - (uint64_t)b * kCscCols);
                for (uint32_t c = w.x / kCscChunk; c <= (w.y - 1) / kCscChunk; ++c) ++cover_count[c + 1];
            } else {
                all_lds = false;
            }
        }
        a->lds_col_fraction = (double)cols_lds / (double)a->ncols;
        a->all_lds = all_lds ? 1 : 0;
        // cover lists (chunk of rows -> LDS-mode super-tiles whose window overlaps it, ascending)
        for (uint32_t c = 0; c < a->nchunks; ++c) cover_count[c + 1] += cover_count[c];
        std::vector<uint32_t> cover(cover_count[a->nchunks]), fill(cover_count.begin(), cover_count.end() - 1);
        for (uint32_t b = 0; b < a->nblocks; ++b) {
            if (desc[b].z != kCscModeLds) continue;
            for (uint32_t c = desc[b].x / kCscChunk; c <= (desc[b].x + desc[b].y - 1) / kCscChunk; ++c)
                cover[fill[c]++] = b;
        }
        // Neighbour hand-off instead of atomics: every super-tile in LDS mode, windows ascending, and a window may
        // overlap its neighbours' only (hi[b-1] <= lo[b+1]); rows no window covers are zero-filled by the next
        // super-tile (the last one takes the tail), which must stay a small job.  A super-tile without entries
        // becomes an empty window at the end of the previous one.
        if (all_lds) {
            std::vector<uint32_t> prev_hi(a->nblocks, 0);
            bool ok = true;
            uint32_t lo1 = 0, hi1 = 0, hi2 = 0;     // window of b - 1, end of the window of b - 2
            const uint64_t fill_cap = 4ull * kCscCols;
            for (uint32_t b = 0; b < a->nblocks && ok; ++b) {
                if (desc[b].z != kCscModeLds) desc[b] = make_uint4(hi1, 0, kCscModeLds, 0);
                const uint32_t lo = desc[b].x, hi = lo + desc[b].y;
                prev_hi[b] = hi1;
                ok = lo >= lo1 && hi >= hi1 && lo >= hi2 && (lo <= hi1 || (uint64_t)(lo - hi1) <= fill_cap);
                hi2 = hi1; lo1 = lo; hi1 = hi;
            }
            if (ok && a->nrows - hi1 > fill_cap) ok = false;
            if (ok) {
                SPAL_HIP_TRY(dev_alloc((void **)&a->d_prev_hi, (size_t)a->nblocks * 4));
                SPAL_HIP_TRY(dev_alloc((void **)&a->d_flags, ((size_t)a->nblocks + 2) * 4));   // flags, -, ticket counter
                SPAL_HIP_TRY(hipMemcpyAsync(a->d_prev_hi, prev_hi.data(), (size_t)a->nblocks * 4, hipMemcpyHostToDevice, a->stream));
                SPAL_HIP_TRY(hipMemsetAsync(a->d_flags, 0, ((size_t)a->nblocks + 2) * 4, a->stream));
                if (!a->h_gave_up) {   // one word of mapped host memory: the kernel's "gave up" report
                    SPAL_HIP_TRY(hipHostMalloc((void **)&a->h_gave_up, 64, hipHostMallocMapped));
                    *a->h_gave_up = 0;
                    SPAL_HIP_TRY(hipHostGetDevicePointer((void **)&a->d_gave_up, a->h_gave_up, 0));
                }
                SPAL_HIP_TRY(hipStreamSynchronize(a->stream));   // prev_hi goes out of scope
                a->ordered = 1;
                // Which super-tile a workgroup takes: when the device holds ALL workgroups of the launch at once
                // (config 4: 245 workgroups, 256 CUs x 1), every one of them becomes resident whatever the dispatch
                // order and a waiting workgroup never keeps its predecessor off the device: blockIdx will do, and the
                // ticket's round trip at the start of every workgroup (+ 4 us of 39 at config 4) is saved.  Larger
                // launches take their super-tile from the start-order ticket (see csc_spmv_scatter).
                {
                    int dev_id = 0, cus = 0, per_cu = 0;
                    (void)hipGetDevice(&dev_id);
                    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id);
                    const size_t lds = std::max(((size_t)kCscCols + a->lds_entries) * (size_t)a->elem_size,
                                                (size_t)kCscCols * (size_t)a->elem_size + ((size_t)kCscCols + 2) * 4);
                    per_cu = csc_scatter_per_cu(a, lds);      // the runtime's occupancy of the compiled kernel; 0 = unknown: ticket
                    a->ticket_auto = (per_cu <= 0 || (uint64_t)((a->nblocks + 7) / 8) * 8 > (uint64_t)cus * (uint64_t)per_cu) ? 1 : 0;
                }
            }
        }
        a->windows_entries = slot;
        if (slot) {
            SPAL_HIP_TRY(dev_alloc(&a->d_windows, (size_t)slot * a->elem_size));
            SPAL_HIP_TRY(dev_alloc((void **)&a->d_chunk_ptr, (size_t)(a->nchunks + 1) * 4));
            SPAL_HIP_TRY(dev_alloc((void **)&a->d_chunk_blk, std::max<size_t>(cover.size(), 1) * 4));
            SPAL_HIP_TRY(hipMemcpyAsync(a->d_chunk_ptr, cover_count.data(), (size_t)(a->nchunks + 1) * 4,
                                        hipMemcpyHostToDevice, a->stream));
            if (!cover.empty())
                SPAL_HIP_TRY(hipMemcpyAsync(a->d_chunk_blk, cover.data(), cover.size() * 4,
                                            hipMemcpyHostToDevice, a->stream));
            SPAL_HIP_TRY(hipStreamSynchronize(a->stream));  // the host vectors go out of scope
        }
    }
    SPAL_HIP_TRY(dev_alloc((void **)&a->d_desc, (size_t)a->nblocks * sizeof(uint4)));
    SPAL_HIP_TRY(hipMemcpyAsync(a->d_desc, desc.data(), (size_t)a->nblocks * sizeof(uint4),
                                hipMemcpyHostToDevice, a->stream));
    if (a->lds_entries) {
        if (!a->d_meta) {
            SPAL_HIP_TRY(dev_alloc((void **)&a->d_meta, (size_t)(a->nnz + kStreamPad) * sizeof(uint32_t)));
            SPAL_HIP_TRY(hipMemsetAsync(a->d_meta, 0, (size_t)(a->nnz + kStreamPad) * sizeof(uint32_t), a->stream));
        }
        hipLaunchKernelGGL(csc_encode_meta, dim3(a->nblocks), dim3(256), 0, a->stream, a->d_colptr,
                           a->d_rowind, a->d_desc, a->d_meta, (uint32_t)a->ncols, (uint32_t)a->cols_per_block);
        SPAL_HIP_TRY(hipGetLastError());
    }
    SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
    return csc_rowtiles_plan(a);
}

// CSC -> CSR on the device (stable sort of the entries by row), kept on the handle.
static int csc_ensure_csr(spal_csc *a) {
    if (a->as_csr) return SPAL_OK;
    uint32_t *rp = nullptr, *ci = nullptr;
    void *va = nullptr;
    uint64_t cap = 0;
    SPAL_TRY(transpose_device(a->device, a->elem_size, a->ncols, a->nrows, a->nnz, a->d_colptr,
                              a->d_rowind, a->d_values, a->stream, &rp, &ci, &va, &cap));
    int st = csr_adopt_device(a->device, a->elem_size, a->nrows, a->ncols, a->nnz, cap, rp, ci, va,
                              &a->as_csr);
    if (st != SPAL_OK) { (void)dev_free(rp); (void)dev_free(ci); (void)dev_free(va); }
    return st;
}

int csc_adopt_device(int device, int elem_size, uint64_t nrows, uint64_t ncols, uint64_t nnz,
                     uint32_t *d_colptr, uint32_t *d_rowind, void *d_values, spal_csc **out) {
    spal_csc *a = new spal_csc;
    a->device = device;
    a->elem_size = elem_size;
    a->nrows = nrows; a->ncols = ncols; a->nnz = nnz;
    a->d_colptr = d_colptr; a->d_rowind = d_rowind; a->d_values = d_values;
    a->lanes_per_col = 16;
    {
        int L = 2;
        const double mean = ncols ? (double)nnz / (double)ncols : 0.0;
        while (L < 64 && (double)L < mean) L <<= 1;
        a->lanes_per_col = L;
    }
    auto bail = [&](int st) {
        a->d_colptr = nullptr; a->d_rowind = nullptr; a->d_values = nullptr;  // stay with the caller
        (void)dev_free(a->d_meta); (void)dev_free(a->d_desc);
        (void)dev_free(a->d_windows); (void)dev_free(a->d_chunk_ptr); (void)dev_free(a->d_chunk_blk);
        (void)dev_free(a->d_prev_hi); (void)dev_free(a->d_flags);
        csc_rowtiles_free(a);
        if (a->h_gave_up) (void)hipHostFree(a->h_gave_up);
        stream_release(a->stream);
        delete a;
        return st;
    };
    hipError_t e = stream_acquire(&a->stream);
    if (e != hipSuccess) return bail(fail(SPAL_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)));
    int st = csc_plan_build(a);
    if (st == SPAL_OK && a->kernel == 2) st = csc_ensure_csr(a);  // setup work, not the first product's
    if (st != SPAL_OK) {
        if (a->as_csr) { (void)spal_csr_destroy(a->as_csr); a->as_csr = nullptr; }
        return bail(st);
    }
    *out = a;
    return SPAL_OK;
}

static void csc_free(spal_csc *a) {
    if (!a) return;
    if (a->as_csr) (void)spal_csr_destroy(a->as_csr);
    (void)dev_free(a->d_colptr);
    (void)dev_free(a->d_rowind);
    (void)dev_free(a->d_values);
    (void)dev_free(a->d_meta);
    (void)dev_free(a->d_desc);
    (void)dev_free(a->d_windows);
    (void)dev_free(a->d_chunk_ptr);
    (void)dev_free(a->d_chunk_blk);
    (void)dev_free(a->d_prev_hi);
    (void)dev_free(a->d_flags);
    csc_rowtiles_free(a);
    if (a->ev_last) (void)hipEventDestroy(a->ev_last);
    if (a->h_gave_up) (void)hipHostFree(a->h_gave_up);
    (void)dev_free(a->d_x);
    (void)dev_free(a->d_y);
    stream_release(a->stream);
    delete a;
}

template <typename T>
static int csc_create(int device, uint64_t nrows, uint64_t ncols, const uint64_t *colptr,
                      uint64_t colptr_len, const uint64_t *rowind, uint64_t rowind_len,
                      const T *values, uint64_t values_len, spal_csc_t *out) {
    if (!out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_create: out is NULL");
    *out = nullptr;
    if (!colptr || (!rowind && rowind_len) || (!values && values_len))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_create: null array");
    int reason = 0;
    SPAL_TRY(spal_csc_validate(nrows, ncols, colptr, colptr_len, rowind, rowind_len, values_len, &reason));
    const uint64_t nnz = colptr[ncols];
    if (ncols >= 0xffffffffull || nrows > 0xffffffffull || nnz > kMaxEntries)
        return fail(SPAL_ERR_UNSUPPORTED,
                    "shape %llu x %llu with %llu entries does not fit 32-bit device indices",
                    (unsigned long long)nrows, (unsigned long long)ncols, (unsigned long long)nnz);
    DeviceGuard guard(device);
    if (guard.status != SPAL_OK) return guard.status;
    std::vector<uint32_t> cp32(ncols + 1), ri32(nnz);
    parallel_for(ncols + 1, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b; i < e; ++i) cp32[i] = (uint32_t)colptr[i];
    });
    parallel_for(nnz, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b; i < e; ++i) ri32[i] = (uint32_t)rowind[i];
    });
    spal_csc *a = new spal_csc;
    a->device = device;
    a->elem_size = (int)sizeof(T);
    a->nrows = nrows; a->ncols = ncols; a->nnz = nnz;
    a->lanes_per_col = pick_lanes_csc(ncols ? (double)nnz / (double)ncols : 0.0);
    const uint64_t cap = nnz + kStreamPad;  // whole-step reads of the LDS-mode stream
    hipError_t e = dev_alloc((void **)&a->d_colptr, (ncols + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = dev_alloc((void **)&a->d_rowind, cap * sizeof(uint32_t));
    if (e == hipSuccess) e = dev_alloc((void **)&a->d_values, cap * sizeof(T));
    if (e == hipSuccess) e = hipMemset((char *)a->d_values + nnz * sizeof(T), 0, kStreamPad * sizeof(T));
    if (e == hipSuccess) e = hipMemcpy(a->d_colptr, cp32.data(), (ncols + 1) * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) e = hipMemcpy(a->d_rowind, ri32.data(), nnz * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) e = hipMemcpy(a->d_values, values, nnz * sizeof(T), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = stream_acquire(&a->stream);
    if (e != hipSuccess) {
        csc_free(a);
        return fail(e == hipErrorOutOfMemory ? SPAL_ERR_OUT_OF_MEMORY : SPAL_ERR_HIP,
                    "spal_csc_create: upload failed: %s", hipGetErrorString(e));
    }
    int st = csc_plan_build(a);
    if (st == SPAL_OK && a->kernel == 2) st = csc_ensure_csr(a);  // setup work, not the first product's
    if (st != SPAL_OK) { csc_free(a); return st; }
    *out = a;
    return SPAL_OK;
}

template <typename T>
static int csc_spmv_host(spal_csc_t a, const T *x, uint64_t x_len, T *y, uint64_t y_len) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_spmv: handle is NULL");
    if (a->elem_size != (int)sizeof(T))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_spmv: handle holds %s values",
                    a->elem_size == 8 ? "f64" : "f32");
    if (x_len != a->ncols)
        return fail(SPAL_ERR_INVALID_ARGUMENT,
                    "dimension mismatch: x.len() = %llu but ncols = %llu (assert_eq!, csc/ops/mul.rs:9)",
                    (unsigned long long)x_len, (unsigned long long)a->ncols);
    if (y_len != a->nrows)
        return fail(SPAL_ERR_INVALID_ARGUMENT, "y.len() = %llu but nrows = %llu",
                    (unsigned long long)y_len, (unsigned long long)a->nrows);
    if (!x || !y) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_spmv: null vector");
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    std::lock_guard<std::mutex> lock(a->mu);
    if (!a->d_x) SPAL_HIP_TRY(dev_alloc((void **)&a->d_x, a->ncols * sizeof(T)));
    if (!a->d_y) SPAL_HIP_TRY(dev_alloc((void **)&a->d_y, a->nrows * sizeof(T)));
    SPAL_HIP_TRY(hipMemcpyAsync(a->d_x, x, a->ncols * sizeof(T), hipMemcpyHostToDevice, a->stream));
    SPAL_TRY(csc_launch(a, a->d_x, a->d_y, a->stream));
    SPAL_HIP_TRY(hipMemcpyAsync(y, a->d_y, a->nrows * sizeof(T), hipMemcpyDeviceToHost, a->stream));
    SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
    // neighbour hand-off: a super-tile hit its spin bound (the backstop; see csc_spmv_scatter) -- this handle keeps to
    // the atomics flush from now on and the product is repeated
    if (a->h_gave_up && __atomic_load_n(a->h_gave_up, __ATOMIC_RELAXED)) {
        {
            std::lock_guard<std::mutex> chain(a->mu_launch);
            __atomic_store_n(a->h_gave_up, 0u, __ATOMIC_RELAXED);
            a->ordered = 0;
            a->handoff_timeouts++;
        }
        SPAL_TRY(csc_launch(a, a->d_x, a->d_y, a->stream));
        SPAL_HIP_TRY(hipMemcpyAsync(y, a->d_y, a->nrows * sizeof(T), hipMemcpyDeviceToHost, a->stream));
        SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
    }
    return SPAL_OK;
}

template <typename T>
static int csc_spmv_dev(spal_csc_t a, const T *x_dev, T *y_dev, void *stream) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_spmv_dev: handle is NULL");
    if (a->elem_size != (int)sizeof(T))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_spmv_dev: handle holds %s values",
                    a->elem_size == 8 ? "f64" : "f32");
    if (!x_dev || !y_dev) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_spmv_dev: null vector");
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    return csc_launch(a, x_dev, y_dev, (hipStream_t)stream);
}

template <typename T>
static int csc_download(spal_csc_t a, uint64_t *colptr, uint64_t *rowind, T *values) {
    if (!a || !colptr) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_download: null argument");
    if (a->elem_size != (int)sizeof(T))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_download: handle holds %s values",
                    a->elem_size == 8 ? "f64" : "f32");
    if (a->nnz && (!rowind || !values)) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_download: null array");
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    std::vector<uint32_t> cp(a->ncols + 1), ri(a->nnz);
    SPAL_HIP_TRY(hipMemcpy(cp.data(), a->d_colptr, cp.size() * 4, hipMemcpyDeviceToHost));
    if (a->nnz) {
        SPAL_HIP_TRY(hipMemcpy(ri.data(), a->d_rowind, ri.size() * 4, hipMemcpyDeviceToHost));
        SPAL_HIP_TRY(hipMemcpy(values, a->d_values, a->nnz * sizeof(T), hipMemcpyDeviceToHost));
    }
    for (uint64_t i = 0; i <= a->ncols; ++i) colptr[i] = cp[i];
    for (uint64_t i = 0; i < a->nnz; ++i) rowind[i] = ri[i];
    return SPAL_OK;
}
}  // namespace spal

using namespace spal;

extern "C" {

int spal_csc_create_f64(int device, uint64_t nrows, uint64_t ncols, const uint64_t *colptr,
                        uint64_t colptr_len, const uint64_t *rowind, uint64_t rowind_len,
                        const double *values, uint64_t values_len, spal_csc_t *out) {
    return csc_create<double>(device, nrows, ncols, colptr, colptr_len, rowind, rowind_len, values,
                              values_len, out);
}
int spal_csc_create_f32(int device, uint64_t nrows, uint64_t ncols, const uint64_t *colptr,
                        uint64_t colptr_len, const uint64_t *rowind, uint64_t rowind_len,
                        const float *values, uint64_t values_len, spal_csc_t *out) {
    return csc_create<float>(device, nrows, ncols, colptr, colptr_len, rowind, rowind_len, values,
                             values_len, out);
}
int spal_csc_destroy(spal_csc_t a) {
    if (!a) return SPAL_OK;
    DeviceGuard guard(a->device);
    csc_free(a);
    return SPAL_OK;
}
int spal_csc_shape(spal_csc_t a, uint64_t *nrows, uint64_t *ncols, uint64_t *nnz, int *elem_size) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_shape: handle is NULL");
    if (nrows) *nrows = a->nrows;
    if (ncols) *ncols = a->ncols;
    if (nnz) *nnz = a->nnz;
    if (elem_size) *elem_size = a->elem_size;
    return SPAL_OK;
}
int spal_csc_spmv_f64(spal_csc_t a, const double *x, uint64_t x_len, double *y, uint64_t y_len) {
    return csc_spmv_host<double>(a, x, x_len, y, y_len);
}
int spal_csc_spmv_f32(spal_csc_t a, const float *x, uint64_t x_len, float *y, uint64_t y_len) {
    return csc_spmv_host<float>(a, x, x_len, y, y_len);
}
int spal_csc_spmv_dev_f64(spal_csc_t a, const double *x_dev, double *y_dev, void *stream) {
    return csc_spmv_dev<double>(a, x_dev, y_dev, stream);
}
int spal_csc_spmv_dev_f32(spal_csc_t a, const float *x_dev, float *y_dev, void *stream) {
    return csc_spmv_dev<float>(a, x_dev, y_dev, stream);
}
int spal_csc_to_csr(spal_csc_t a, spal_csr_t *out) {
    if (!a || !out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_to_csr: null argument");
    *out = nullptr;
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    std::lock_guard<std::mutex> lock(a->mu);
    uint32_t *rp = nullptr, *ci = nullptr;
    void *va = nullptr;
    uint64_t cap = 0;
    SPAL_TRY(transpose_device(a->device, a->elem_size, a->ncols, a->nrows, a->nnz, a->d_colptr,
                              a->d_rowind, a->d_values, a->stream, &rp, &ci, &va, &cap));
    int st = csr_adopt_device(a->device, a->elem_size, a->nrows, a->ncols, a->nnz, cap, rp, ci, va, out);
    if (st != SPAL_OK) { (void)dev_free(rp); (void)dev_free(ci); (void)dev_free(va); }
    return st;
}

int spal_csr_to_csc(spal_csr_t a, spal_csc_t *out) {
    if (!a || !out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_to_csc: null argument");
    *out = nullptr;
    if (!a->parts.empty())
        return fail(SPAL_ERR_UNSUPPORTED, "spal_csr_to_csc: %llu entries do not fit one set of 32-bit device offsets "
                    "(a CSC handle is not split into blocks)", (unsigned long long)a->nnz);
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    std::lock_guard<std::mutex> lock(a->mu);
    uint32_t *cp = nullptr, *ri = nullptr;
    void *va = nullptr;
    uint64_t cap = 0;
    SPAL_TRY(transpose_device(a->device, a->elem_size, a->nrows, a->ncols, a->nnz, a->d_rowptr,
                              a->d_colind, a->d_values, a->stream, &cp, &ri, &va, &cap));
    int st = csc_adopt_device(a->device, a->elem_size, a->nrows, a->ncols, a->nnz, cp, ri, va, out);
    if (st != SPAL_OK) { (void)dev_free(cp); (void)dev_free(ri); (void)dev_free(va); }
    return st;
}

int spal_csc_download_f64(spal_csc_t a, uint64_t *colptr, uint64_t *rowind, double *values) {
    return csc_download<double>(a, colptr, rowind, values);
}
int spal_csc_download_f32(spal_csc_t a, uint64_t *colptr, uint64_t *rowind, float *values) {
    return csc_download<float>(a, colptr, rowind, values);
}

int spal_csc_autotune_f64(spal_csc_t a, const double *x_dev, double *y_dev, void *stream, int iters) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_autotune: handle is NULL");
    if (a->kernel != 2) return SPAL_OK;  // the scatter kernel has a single form
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    SPAL_TRY(csc_ensure_csr(a));
    return spal_csr_autotune_f64(a->as_csr, x_dev, y_dev, stream, iters);
}
int spal_csc_autotune_f32(spal_csc_t a, const float *x_dev, float *y_dev, void *stream, int iters) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_autotune: handle is NULL");
    if (a->kernel != 2) return SPAL_OK;
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    SPAL_TRY(csc_ensure_csr(a));
    return spal_csr_autotune_f32(a->as_csr, x_dev, y_dev, stream, iters);
}

int spal_csc_set_option(spal_csc_t a, const char *key, int64_t value) {
    if (!a || !key) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_set_option: null argument");
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    std::lock_guard<std::mutex> lock(a->mu);
    if (!strcmp(key, "lanes_per_col")) {
        if (value == 0) {
            a->lanes_per_col = pick_lanes_csc(a->ncols ? (double)a->nnz / (double)a->ncols : 0.0);
            return SPAL_OK;
        }
        if (value < 2 || value > 64 || (value & (value - 1)))
            return fail(SPAL_ERR_INVALID_ARGUMENT, "lanes_per_col must be one of 2,4,8,16,32,64");
        a->lanes_per_col = (int)value;
        return SPAL_OK;
    }
    if (!strcmp(key, "kernel")) {
        // 1 = atomic scatter (the path BASELINE config 4 names), 2 = transposed
        // (device CSC->CSR once, then the CSR kernels), 0 = auto = 2
        if (value < 0 || value > 2) return fail(SPAL_ERR_INVALID_ARGUMENT, "kernel must be 0, 1 or 2");
        a->kernel = value == 1 ? 1 : 2;
        if (a->kernel == 2) return csc_ensure_csr(a);
        if (!a->rowtiles) return csc_rowtiles_plan(a);   // the scatter path's row tiles: built when the path is selected
        return SPAL_OK;
    }
    if (!strcmp(key, "flush")) {
        // 0 (default) = window rows flushed with global atomics; 1 = LDS windows stored per super-tile,
        // then an ordered reduce (no global atomics; measured 89.6 vs 80.1 us at config 4: the LDS
        // atomics, not the flush, bound the kernel)
        // 2 = global atomics even where the plan allows the neighbour hand-off (0: hand-off when allowed)
        if (value < 0 || value > 2) return fail(SPAL_ERR_INVALID_ARGUMENT, "flush must be 0, 1 or 2");
        a->flush = (int)value;
        return SPAL_OK;
    }
    if (!strcmp(key, "cols_per_block")) {
        // columns of a super-tile of the scatter kernel: 0 = the widest of 4096 / 2048 / 1024 whose row windows fit LDS
        if (value != 0 && value != 1024 && value != 2048 && value != 4096)
            return fail(SPAL_ERR_INVALID_ARGUMENT, "cols_per_block must be 0 (auto), 1024, 2048 or 4096");
        a->user_cols = (int)value;
        return csc_plan_build(a);
    }
    if (!strcmp(key, "ticket")) {
        // neighbour hand-off: 1 = a workgroup's super-tile is its start-order ticket, 0 = its blockIdx, -1 (default) =
        // the ticket unless the device holds all workgroups of the launch at once
        if (value < -1 || value > 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "ticket must be -1 (auto), 0 or 1");
        a->use_ticket = (int)value;
        return SPAL_OK;
    }
    if (!strcmp(key, "row_tiles")) {
        // the scatter path over row tiles (spal_csc_rowtiles.hip): -1 / 1 = where every tile's window of x fits LDS, 0 = never
        if (value < -1 || value > 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "row_tiles must be -1 (auto), 0 or 1");
        a->rowtiles_user = (int)value;
        return csc_rowtiles_plan(a);
    }
    if (!strcmp(key, "row_tile_rows")) {
        if (value != 0 && value != 1024 && value != 2048 && value != 4096)
            return fail(SPAL_ERR_INVALID_ARGUMENT, "row_tile_rows must be 0 (auto), 1024, 2048 or 4096");
        a->rt_rows_user = (uint32_t)value;
        return csc_rowtiles_plan(a);
    }
    if (!strcmp(key, "lds")) {
        if (value != 0 && value != 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "lds must be 0 or 1");
        a->use_lds = (int)value;
        return csc_plan_build(a);
    }
    return fail(SPAL_ERR_INVALID_ARGUMENT, "unknown option '%s'", key);
}
int spal_csc_status(spal_csc_t a, int *invalid_products) {
    if (!a || !invalid_products) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_status: null argument");
    std::lock_guard<std::mutex> chain(a->mu_launch);
    *invalid_products = a->handoff_timeouts + ((a->h_gave_up && __atomic_load_n(a->h_gave_up, __ATOMIC_RELAXED)) ? 1 : 0);
    return SPAL_OK;
}
int spal_csc_describe(spal_csc_t a, char *buf, size_t buf_len) {
    if (!a || !buf || !buf_len) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_describe: null argument");
    snprintf(buf, buf_len,
             "{\"format\": \"csc\", \"dtype\": \"%s\", \"nrows\": %llu, \"ncols\": %llu, \"nnz\": %llu, "
             "\"kernel\": \"%s\", \"cols_per_block\": %d, \"blocks\": %u, \"lanes_per_col\": %d, "
             "\"lds_window_bytes\": %llu, \"lds_col_fraction\": %.4f, \"flush\": \"%s\", "
             "\"window_store_bytes\": %llu, \"ticket\": %d, \"handoff_timeouts\": %d, \"uniform_columns\": %d, "
             "\"row_tiles\": %d, \"row_tile_rows\": %u, \"row_tile_count\": %u, \"row_tile_x_window\": %u, \"row_tiles_failed\": %d}",
             a->elem_size == 8 ? "f64" : "f32", (unsigned long long)a->nrows,
             (unsigned long long)a->ncols, (unsigned long long)a->nnz,
             a->kernel == 2 ? "transposed_csr" : a->lds_entries ? "lds_privatised_scatter" : "atomic_scatter",
             a->cols_per_block, a->nblocks,
             a->lanes_per_col, (unsigned long long)a->lds_entries * (unsigned long long)a->elem_size,
             a->lds_col_fraction, (a->flush == 1 && a->d_windows) ? "windows_then_reduce"
                                  : (a->flush == 0 && a->ordered) ? "neighbour_handoff" : "global_atomics",
             (unsigned long long)a->windows_entries * (unsigned long long)a->elem_size, a->use_ticket < 0 ? a->ticket_auto : a->use_ticket,
             a->handoff_timeouts + ((a->h_gave_up && __atomic_load_n(a->h_gave_up, __ATOMIC_RELAXED)) ? 1 : 0),
             a->uniform_cols ? 1 : 0,
             (a->rowtiles && a->rowtiles_user != 0 && a->flush == 0) ? 1 : 0, a->rt_rows, a->rt_ntiles, a->rt_xcap, a->rowtiles_failed);
    return SPAL_OK;
}

}  // extern "C"
