// spal_csr_blockwin.hip -- CSR y = A * x for SKEWED row lengths whose columns stay near the rows' own: entries streamed, rows
// summed out of a product strip, x out of ONE LDS window per row block (gfx950).  Round 4, late; reference:
// src/csr/ops/mul.rs:25-45 (the order of a row's additions).
//
// The stream kernels give a lane a row: a 64-row tile costs as many steps as its LONGEST row holds entries, and a row beyond
// what a lane may sum sends the tile elsewhere.  Power-law row lengths (mean 10, 0.7 % of the rows above 128 entries holding a
// fifth of the entries) made that 0.18 of the roofline; the row split (A = A_short + A_long) 0.255: its short part still
// walks ragged tiles, its long rows gather x through the vector memory path (~4 clocks per gathered line and CU).  Here the
// ENTRIES are the unit of work, whatever row they belong to:
//   * a workgroup of 1024 threads takes a block of RB consecutive rows (512 ... 4096, the plan's choice) whose columns span at
//     most what LDS holds beside the rest (14 336 columns of f64): the window of x is staged ONCE per block -- every gather of
//     the block, short row or long, is an LDS read;
//   * the block's entries go by in passes of 4096 (four per thread, coalesced, the next pass's loads in flight): every entry's
//     product -- rounded once, as in the reference -- lands in a strip in LDS in entry order;
//   * the rows that a pass touches are found from the block's rowptr (LDS): a row of at most kBwShort entries is summed by ONE
//     thread, left to right, continuing from the carry when the pass boundary cut it -- the reference's order of additions,
//     bit for bit; a longer row by a wave (strided partial sums, a shuffle tree: 1e-10), carried across passes the same way.
// One row is open at the end of a pass at most; its running sum waits in one of two carry slots (by pass parity).
// Chosen at setup by time against the row split (spal_csr.hip: csr_plan_build); option "blockwin" -1 / 0 / 1.
#include <atomic>

#include "spal_internal.hpp"

namespace spal {

constexpr int kBwThreads = 1024;
constexpr uint32_t kBwPass = 4096;     // entries per pass
constexpr uint32_t kBwItems = kBwPass / kBwThreads;
constexpr uint32_t kBwShort = 32;      // rows up to this many entries are summed by one thread, in the reference's order
constexpr uint32_t kBwLongCap = kBwPass / (kBwShort + 1) + 4;   // long rows a pass can touch
constexpr uint32_t kBwUnit = 512;      // rows: the windows are measured per unit, a block is 1, 2, 4 or 8 units
constexpr size_t kBwLdsMax = 160 * 1024;

// setup: {first column, one past the last column} of every unit of kBwUnit rows ({~0, 0}: no entries)
__global__ __launch_bounds__(256) void bw_unit_windows(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
                                                       uint32_t nrows, uint2 *__restrict__ out) {
    __shared__ uint32_t s_lo[4], s_hi[4];
    const uint32_t r0 = blockIdx.x * kBwUnit, r1 = min(r0 + kBwUnit, nrows);
    const uint32_t e0 = rowptr[r0], e1 = rowptr[r1];
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (uint32_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const uint32_t c = colind[e];
        lo = min(lo, c);
        hi = max(hi, c + 1u);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0)
        out[blockIdx.x] = make_uint2(min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])), max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3])));
}

// first i in [lo, n] with rp[i] >= v (rp ascending, rp[n] >= v), by the whole wave: two rounds of 64 probes for n - lo <= 4096
__device__ __forceinline__ uint32_t bw_first_at_least(const uint32_t *rp, uint32_t lo, uint32_t n, uint32_t v, uint32_t lane) {
    uint32_t a = lo, len = n - lo + 1u;   // candidates a ... a + len - 1; the last one qualifies
    while (len > 64u) {                   // (wave-uniform)
        const uint32_t step = (len + 63u) / 64u;
        const uint32_t idx = min(a + (lane + 1u) * step - 1u, a + len - 1u);   // the last candidate of the lane's chunk
        const uint64_t m = __ballot(rp[idx] >= v);
        const uint32_t f = (uint32_t)__builtin_ctzll(m);                       // m != 0: the last lane probes the last candidate
        const uint32_t na = a + f * step;
        len = min(a + (f + 1u) * step, a + len) - na;
        a = na;
    }
    const uint64_t m = __ballot(lane < len && rp[min(a + lane, a + len - 1u)] >= v);
    return a + (uint32_t)__builtin_ctzll(m);
}

template <typename T>
__global__ __launch_bounds__(kBwThreads) void csr_spmv_blockwin(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
                                                                const T *__restrict__ vals, const T *__restrict__ x, T *__restrict__ y,
                                                                const uint2 *__restrict__ bwin, uint32_t nrows, uint32_t ncols,
                                                                uint32_t RB, uint32_t nblocks, uint32_t per_xcd, uint32_t win_cols) {
    extern __shared__ __attribute__((aligned(16))) unsigned char spal_bw_smem[];
    T *xw = reinterpret_cast<T *>(spal_bw_smem);                      // win_cols (a multiple of 256)
    T *sp = xw + win_cols;                                             // kBwPass products, in entry order
    T *s_carry = sp + kBwPass;                                         // [2]: the running sum of the row a pass boundary cut
    uint32_t *s_rp = reinterpret_cast<uint32_t *>(s_carry + 2);        // RB + 1 (+ 1 pad)
    uint32_t *s_long = s_rp + RB + 2;                                  // kBwLongCap
    uint32_t *s_nlong = s_long + kBwLongCap;

    // consecutive blocks share most of their windows: a block's neighbours run on the same XCD (one L2)
    const uint32_t blk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (blk >= nblocks) return;   // block-uniform
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t r0 = blk * RB, nr = min(RB, nrows - r0);
    const uint32_t e0 = rowptr[r0], e1 = rowptr[r0 + nr];
    // the first pass's entries are requested before anything else
    uint32_t cc[kBwItems], nc[kBwItems];
    T cv[kBwItems], nv[kBwItems];
#pragma unroll
    for (uint32_t k = 0; k < kBwItems; ++k) {
        const uint32_t idx = min(e0 + k * kBwThreads + t, e1 ? e1 - 1u : 0u);
        cc[k] = e1 > e0 ? colind[idx] : 0u;
        cv[k] = e1 > e0 ? vals[idx] : T(0);
    }
    for (uint32_t i = t; i <= nr; i += kBwThreads) s_rp[i] = rowptr[r0 + i];
    const uint2 win = bwin[blk];            // {first column (a multiple of 256), columns}
    const uint32_t c0 = win.x;
    {
        const uint32_t wn = min(win.y, ncols - min(c0, ncols));
        if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0 && sizeof(T) == 8) {   // (uniform) 16-byte loads of two columns
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2 *xs = reinterpret_cast<const d2 *>(x + c0);
            d2 *xd = reinterpret_cast<d2 *>(xw);
            const uint32_t pairs = wn / 2u;
            for (uint32_t i = t; i < pairs; i += kBwThreads) xd[i] = xs[i];
            if (t == 0 && (wn & 1u)) xw[wn - 1u] = x[c0 + wn - 1u];
        } else {
            for (uint32_t i = t; i < wn; i += kBwThreads) xw[i] = x[c0 + i];
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < nr; i += kBwThreads)   // empty rows: nothing below writes them
        if (s_rp[i + 1] == s_rp[i]) y[r0 + i] = T(0);

    uint32_t rlo = 0, parity = 0;
    for (uint32_t ps = e0; ps < e1; ps += kBwPass, parity ^= 1u) {   // (block-uniform)
        const uint32_t pe = min(ps + kBwPass, e1);
        const bool more = pe < e1;
        if (more) {
#pragma unroll
            for (uint32_t k = 0; k < kBwItems; ++k) {
                const uint32_t idx = min(pe + k * kBwThreads + t, e1 - 1u);
                nc[k] = colind[idx];
                nv[k] = vals[idx];
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < kBwItems; ++k) {
            const uint32_t idx = ps + k * kBwThreads + t;
            if (idx < pe) sp[k * kBwThreads + t] = cv[k] * xw[cc[k] - c0];   // one rounding, as `val * x[col]` in the reference
        }
        if (t == 0) *s_nlong = 0u;
        __syncthreads();
        // rows rlo ... rhi - 1 may hold entries of [ps, pe)
        const uint32_t rhi = bw_first_at_least(s_rp, rlo, nr, pe, lane);
        for (uint32_t i = rlo + t; i < rhi; i += kBwThreads) {
            const uint32_t rs = s_rp[i], re = s_rp[i + 1];
            const uint32_t a = max(rs, ps), b = min(re, pe);
            if (a >= b) continue;                               // empty, or ended where the pass begins
            if (re - rs > kBwShort) {
                s_long[atomicAdd(s_nlong, 1u)] = i;             // (at most kBwLongCap of them touch a pass)
                continue;
            }
            T acc = rs < ps ? s_carry[parity ^ 1u] : T(0);      // a cut row goes on where the last pass stopped
            uint32_t j = a - ps;
            const uint32_t jb = b - ps;
            for (; j + 4u <= jb; j += 4u) {
                const T v0 = sp[j], v1 = sp[j + 1], v2 = sp[j + 2], v3 = sp[j + 3];
                acc = acc + v0;
                acc = acc + v1;
                acc = acc + v2;
                acc = acc + v3;
            }
            for (; j < jb; ++j) acc = acc + sp[j];
            if (re <= pe) y[r0 + i] = acc;
            else s_carry[parity] = acc;
        }
        __syncthreads();
        const uint32_t nlong = *s_nlong;
        for (uint32_t li = wave; li < nlong; li += kBwThreads / 64) {   // wave-uniform
            const uint32_t i = s_long[li];
            const uint32_t rs = s_rp[i], re = s_rp[i + 1];
            const uint32_t a = max(rs, ps) - ps, b = min(re, pe) - ps;
            T part = T(0);
            for (uint32_t j = a + lane; j < b; j += 64u) part = part + sp[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part = part + __shfl_xor(part, o, 64);
            const T tot = (rs < ps ? s_carry[parity ^ 1u] : T(0)) + part;
            if (lane == 0) {
                if (re <= pe) y[r0 + i] = tot;
                else s_carry[parity] = tot;
            }
        }
        __syncthreads();
        rlo = s_rp[rhi] == pe ? rhi : rhi - 1u;   // the row that holds entry pe (rhi >= 1: rp[0] = e0 < pe)
        if (more) {
#pragma unroll
            for (uint32_t k = 0; k < kBwItems; ++k) { cc[k] = nc[k]; cv[k] = nv[k]; }
        }
    }
}

static size_t bw_lds_bytes(uint32_t RB, uint32_t win_cols, size_t esz) {
    return (size_t)win_cols * esz + (size_t)kBwPass * esz + 2 * esz + (size_t)(RB + 2 + kBwLongCap + 2) * 4;
}

void blockwin_free(spal_csr *a) {
    a->bw_on = 0;
    (void)dev_free(a->d_bwin); a->d_bwin = nullptr;
    a->bw_blocks = 0; a->bw_rows = 0; a->bw_cols = 0;
}

// Measures the units' windows, picks the tallest block whose widest window fits LDS; leaves a->bw_rows = 0 when none does
// (or the matrix is too small to be worth a 1024-thread workgroup per block).
int blockwin_plan(spal_csr *a) {
    blockwin_free(a);
    if (a->nnz == 0 || a->nrows < kBwUnit || !a->parts.empty()) return SPAL_OK;
    const uint32_t nunits = (uint32_t)((a->nrows + kBwUnit - 1) / kBwUnit);
    DevBuf d_win;
    SPAL_HIP_TRY(d_win.alloc((size_t)nunits * sizeof(uint2)));
    hipLaunchKernelGGL(bw_unit_windows, dim3(nunits), dim3(256), 0, a->stream, a->d_rowptr, a->d_colind, (uint32_t)a->nrows,
                       d_win.as<uint2>());
    SPAL_HIP_TRY(hipGetLastError());
    std::vector<uint2> win(nunits);
    SPAL_HIP_TRY(hipMemcpyAsync(win.data(), d_win.p, win.size() * sizeof(uint2), hipMemcpyDeviceToHost, a->stream));
    SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
    const size_t esz = (size_t)a->elem_size;
    for (uint32_t units = 8; units >= 1; units >>= 1) {
        const uint32_t RB = units * kBwUnit;
        const uint32_t nb = (uint32_t)((a->nrows + RB - 1) / RB);
        std::vector<uint2> bw(nb);
        uint32_t widest = 0;
        for (uint32_t b = 0; b < nb; ++b) {
            uint32_t lo = 0xffffffffu, hi = 0;
            for (uint32_t u = b * units; u < std::min(nunits, (b + 1) * units); ++u) { lo = std::min(lo, win[u].x); hi = std::max(hi, win[u].y); }
            if (hi <= lo) { bw[b] = make_uint2(0u, 0u); continue; }
            const uint32_t c0 = lo & ~255u;
            bw[b] = make_uint2(c0, hi - c0);
            widest = std::max(widest, hi - c0);
        }
        const uint32_t win_cols = std::max(256u, (widest + 255u) & ~255u);
        if (bw_lds_bytes(RB, win_cols, esz) > kBwLdsMax) continue;
        SPAL_HIP_TRY(dev_alloc((void **)&a->d_bwin, (size_t)nb * sizeof(uint2)));
        SPAL_HIP_TRY(hipMemcpy(a->d_bwin, bw.data(), (size_t)nb * sizeof(uint2), hipMemcpyHostToDevice));
        a->bw_blocks = nb;
        a->bw_rows = RB;
        a->bw_cols = win_cols;
        return SPAL_OK;
    }
    return SPAL_OK;
}

template <typename T>
static hipError_t bw_launch_t(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    static std::atomic<uint64_t> configured{0};   // devices on which this instantiation's LDS cap has been raised
    const uint64_t bit = 1ull << (a->device & 63);
    if (!(configured.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(csr_spmv_blockwin<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwLdsMax);
        if (e != hipSuccess) return e;
        configured.fetch_or(bit, std::memory_order_relaxed);
    }
    const uint32_t per_xcd = (a->bw_blocks + 7u) / 8u;
    const size_t lds = bw_lds_bytes(a->bw_rows, a->bw_cols, sizeof(T));
    hipLaunchKernelGGL(csr_spmv_blockwin<T>, dim3(per_xcd * 8u), dim3(kBwThreads), lds, st, a->d_rowptr, a->d_colind,
                       (const T *)a->d_values, (const T *)x, (T *)y, a->d_bwin, (uint32_t)a->nrows, (uint32_t)a->ncols, a->bw_rows,
                       a->bw_blocks, per_xcd, a->bw_cols);
    return hipGetLastError();
}

hipError_t blockwin_launch(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    return a->elem_size == 8 ? bw_launch_t<double>(a, x, y, st) : bw_launch_t<float>(a, x, y, st);
}

}  // namespace spal
