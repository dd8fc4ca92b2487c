// spal_csr_blockwin.hip -- CSR y = A * x for SKEWED row lengths whose columns stay near the rows' own: entries streamed, rows
// summed out of a product strip, x out of ONE LDS window per row block (gfx950).  Round 4, late; reference:
// src/csr/ops/mul.rs:25-45 (the order of a row's additions).
//
// The stream kernels give a lane a row: a 64-row tile costs as many steps as its LONGEST row holds entries, and a row beyond
// what a lane may sum sends the tile elsewhere.  Power-law row lengths (mean 10, 0.7 % of the rows above 128 entries holding a
// fifth of the entries) made that 0.18 of the roofline; the row split (A = A_short + A_long) 0.255: its short part still
// walks ragged tiles, its long rows gather x through the vector memory path (~4 clocks per gathered line and CU).  Here the
// ENTRIES are the unit of work, whatever row they belong to (0.47 - 0.48 on the same matrix):
//   * a workgroup of 1024 threads takes a block of RB consecutive rows at a time (512 ... 4096, the plan's choice) whose columns
//     span at most what LDS holds beside the rest (~12 000 columns of f64): the window of x is staged ONCE per block -- every
//     gather of the block, short row or long, is an LDS read.  The grid is RESIDENT, a workgroup per CU: it walks ~4 blocks of
//     its XCD's eighth of the matrix (neighbours share most of their windows and one L2; the plan deals the blocks longest
//     first, each to the walk with the least work so far) with the NEXT block's window and row pointers on their way into
//     registers while the current block's passes go by; all it needs to know about a block is one 32-byte record;
//   * the block's entries go by in passes of kBwPass = 3072 (three per thread, coalesced, requested two passes ahead and
//     unconditionally, so that the waits are counted): every entry's product -- rounded once, as in the reference -- lands in
//     one of two strips in LDS in entry order;
//   * ONE phase and one barrier per pass: while pass p's rows are summed out of strip p & 1, pass p + 1's products go into
//     the other strip and its rows are looked at (three rotating lists of the rows that are not one thread's business);
//   * rows: the first eight waves take a row per thread -- a row of at most kBwShort = 32 entries is summed by ONE thread,
//     left to right, continuing from the carry when the pass boundary cut it: the reference's order of additions, bit for bit;
//     the other eight waves take the listed rows, 16 lanes per row (a whole wave above 512 entries inside the pass): strided
//     partial sums and a DPP tree (1e-10), carried across passes the same way.
// One row is open at the end of a pass at most; its running sum waits in one of two carry slots (by pass parity).
// What bounds it (lab build -DSPAL_BW_STAMPS, per block of 20.9K entries: block change 2.7 us + first pass 1.9 us, phases 13 us):
// the chain inside a phase -- a barrier, the row search, a row's dependent sums -- on a CU that holds one workgroup (counters: the
// LDS pipe is busy a third of the time; 1.16 x the algorithmic bytes fetched, 4.4 TB/s) and the spread of the walks' ends
// (54 ... 75 us although their entry counts are equal).
// Chosen at setup by time against the row split (spal_csr.hip: csr_plan_build); option "blockwin" -1 / 0 / 1.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <vector>

#include "spal_internal.hpp"

namespace spal {

constexpr int kBwThreads = 1024;
#ifndef SPAL_BW_PASS
#define SPAL_BW_PASS 3072
#endif
#ifndef SPAL_BW_ROW_WAVES
#define SPAL_BW_ROW_WAVES 8
#endif
constexpr uint32_t kBwPass = SPAL_BW_PASS;     // entries per pass
constexpr uint32_t kBwItems = kBwPass / kBwThreads;
constexpr uint32_t kBwRowThreads = SPAL_BW_ROW_WAVES * 64;                  // waves 0 ... : a short row per thread
constexpr uint32_t kBwGroups = (kBwThreads - kBwRowThreads) / 16;           // the other waves: a long row per 16 lanes
constexpr uint32_t kBwShort = 32;      // rows up to this many entries are summed by one thread, in the reference's order
constexpr uint32_t kBwLongCap = (kBwPass / (kBwShort + 1) + 4 + 1) & ~1u;   // long rows a pass can touch
constexpr uint32_t kBwHeavy = 512;     // entries of ONE row inside a pass above which a whole wave sums it (16 lanes below)
constexpr uint32_t kBwHeavyCap = (kBwPass / (kBwHeavy + 1) + 4 + 1) & ~1u;   // (even: what follows it in LDS is 8-byte aligned)
constexpr uint32_t kBwListCap = kBwLongCap + kBwHeavyCap;   // one pass's lists: 16-lane rows, then heavy rows
constexpr uint32_t kBwWinLoads = 8;    // 16-byte loads of the window a thread has in flight at once
constexpr uint32_t kBwUnit = 512;      // rows: the windows are measured per unit, a block is 1, 2, 4 or 8 units
constexpr size_t kBwLdsMax = 160 * 1024;

// setup: {first column, one past the last column} of every unit of kBwUnit rows ({~0, 0}: no entries)
__global__ __launch_bounds__(256) void bw_unit_windows(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
                                                       uint32_t nrows, uint2 *__restrict__ out, uint32_t *__restrict__ first_entry) {
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
    if (threadIdx.x == 0) {
        out[blockIdx.x] = make_uint2(min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])), max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3])));
        first_entry[blockIdx.x] = e0;
        if (blockIdx.x + 1 == gridDim.x) first_entry[gridDim.x] = e1;
    }
}

// sum over the 16 lanes of a DPP row, in lane 0 of the row (row_shl: lane i receives lane i + n of its row, 0.0 past the end)
template <int CTRL>
__device__ __forceinline__ double bw_dpp(double v) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    int lo = (int)(uint32_t)u, hi = (int)(uint32_t)(u >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
template <int CTRL>
__device__ __forceinline__ float bw_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <typename T>
__device__ __forceinline__ T bw_sum16(T v) {
    v = v + bw_dpp<0x108>(v);
    v = v + bw_dpp<0x104>(v);
    v = v + bw_dpp<0x102>(v);
    v = v + bw_dpp<0x101>(v);
    return v;
}
template <typename T>
__device__ __forceinline__ T bw_sum64(T v) {   // in lane 0 of the wave
    v = bw_sum16(v);
    v = v + __shfl_down(v, 16, 64);
    v = v + __shfl_down(v, 32, 64);
    return v;
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

// -DSPAL_BW_STAMPS (lab builds): thread 0 of every workgroup adds up wall_clock64() (100 MHz) per phase into g_bw_stamps[block][8]
#ifdef SPAL_BW_STAMPS
__device__ unsigned long long *g_bw_stamps = nullptr;
// (summed in registers and written once at the end: a read-modify-write per stamp would wait for every load in flight)
#define BW_STAMP(i) do { const unsigned long long now_ = wall_clock64(); acc_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define BW_STAMP(i) do { } while (0)
#endif

// WIN = false: no block's window fits LDS (columns anywhere) -- the same passes, strips and row sums, x gathered from global
// memory / L2 one phase ahead of its product (three register sets take turns instead of two).
// PRE (WIN, x 16-byte aligned): a workgroup walks its blocks with the NEXT block's window and row pointers on their way into
// registers (kBwWinRegs 16-byte vectors and kBwRpRegs words per thread) while the current block's passes go by: between two blocks
// they are written to LDS instead of being waited for behind a barrier (4.4 of a block's 18.4 us).
constexpr uint32_t kBwWinRegs = 7;   // 7 x 16 bytes x 1024 threads = 112 KB of window
constexpr uint32_t kBwRpRegs = 5;    // 5 x 1024 words >= 4097 row pointers
template <typename T, bool WIN, bool PRE>
__global__ __launch_bounds__(kBwThreads) void csr_spmv_blockwin(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
                                                                const T *__restrict__ vals, const T *__restrict__ x, T *__restrict__ y,
                                                                const uint32_t *__restrict__ order,
                                                                uint32_t nrows, uint32_t ncols, uint32_t RB, uint32_t nblocks,
                                                                uint32_t win_cols, uint32_t per_xcd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char spal_bw_smem[];
    T *xw = reinterpret_cast<T *>(spal_bw_smem);                      // win_cols (a multiple of 256)
    T *sp = xw + win_cols;                                             // [2][kBwPass] products in entry order: pass p in strip p & 1
    T *s_carry = sp + 2 * kBwPass;                                     // [2]: the running sum of the row a pass boundary cut
    uint32_t *s_rp = reinterpret_cast<uint32_t *>(s_carry + 2);        // RB + 1 (+ 1 pad)
    uint32_t *s_list = s_rp + RB + 2;                                  // [3][kBwListCap]: pass p's listed rows in list p % 3
    uint32_t *s_n = s_list + 3 * kBwListCap;                           // [3][2]: the lists' lengths {16-lane rows, heavy rows}

    // blocks are dealt in the plan's order: XCD k (workgroups k, k + 8, ...) takes the k-th eighth of the blocks -- neighbours
    // share most of their windows and meet in one L2 --, inside it the blocks with the most entries first (a workgroup per CU
    // and four rounds of them: the launch ends with its shortest blocks)
    // (everything a workgroup needs to know about its block comes in ONE record, by blockIdx: read one after the other --
    //  position in the order, then rowptr and the window -- it was two more round trips before the first useful load)
    // A RESIDENT grid (8 x J workgroups, J <= CUs per XCD): workgroup w walks the records (w >> 3) + i * J of XCD w & 7's list,
    // which the plan lays out so that every walk holds a block of every size class (longest first, dealt back and forth).
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t xcd = blockIdx.x & 7u, J = gridDim.x >> 3;
    uint32_t slot = blockIdx.x >> 3;
    if (slot >= per_xcd) return;
    uint4 rec0 = reinterpret_cast<const uint4 *>(order)[(slot * 8u + xcd) * 2u];       // {block, first entry, one past the last, first column}
    uint32_t win_n = order[(slot * 8u + xcd) * 8u + 4u];                               // columns of the window
    if (rec0.x >= nblocks) return;   // block-uniform (the XCD's list is padded at its end)
#ifdef SPAL_BW_STAMPS
    unsigned long long last_ = wall_clock64(), acc_[4] = {0, 0, 0, 0};
    const unsigned long long first_ = last_;
    uint32_t nblk_ = 0;
#endif
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 wr[kBwWinRegs];
    uint32_t rpr[kBwRpRegs];
    // (unconditional, clamped: the counted waits of the passes' loads must not depend on a branch)
    auto prefetch_block = [&](const uint4 &rec, uint32_t wn_cols) {
        constexpr uint32_t V = 16 / sizeof(T);
        const uint32_t wn = min(wn_cols, ncols - min(rec.w, ncols));
        const uint32_t nvec = max((wn + V - 1u) / V, 1u);
        const u4 *xs = reinterpret_cast<const u4 *>(x + rec.w);
        // (the last vector of a window that ends at the matrix's last column may reach past x by one element: clamped below)
        const uint32_t safe = (ncols - min(rec.w, ncols)) / V;   // whole vectors of x from the window's first column
#pragma unroll
        for (uint32_t k = 0; k < kBwWinRegs; ++k) wr[k] = xs[min(min(k * kBwThreads + t, nvec - 1u), safe ? safe - 1u : 0u)];
        const uint32_t rb0 = rec.x * RB, nrb = min(RB, nrows - rb0);
#pragma unroll
        for (uint32_t k = 0; k < kBwRpRegs; ++k) rpr[k] = rowptr[rb0 + min(k * kBwThreads + t, nrb)];
    };
    if (PRE) prefetch_block(rec0, win_n);
  for (;;) {   // the workgroup's blocks
    const uint32_t blk = rec0.x;
    const uint32_t nslot = slot + J;
    uint4 rec1 = rec0;
    uint32_t win_n1 = win_n;
    bool has_next = false;
    if (nslot < per_xcd) {   // (block-uniform)
        rec1 = reinterpret_cast<const uint4 *>(order)[(nslot * 8u + xcd) * 2u];
        win_n1 = order[(nslot * 8u + xcd) * 8u + 4u];
        has_next = rec1.x < nblocks;
        if (!has_next) { rec1 = rec0; win_n1 = win_n; }
    }
    const uint32_t r0 = blk * RB, nr = min(RB, nrows - r0);
    const uint32_t e0 = rec0.y, e1 = rec0.z;
    const uint32_t npass = (e1 - e0 + kBwPass - 1u) / kBwPass;
    // Entries of pass p, requested unconditionally (a pass beyond the block's last reads the last entry again): under
    // conditions the compiler no longer knows how many loads are in flight and waits for ALL of them before the first product.
    struct Set { uint32_t c[kBwItems]; T v[kBwItems]; T xg[kBwItems]; };   // a pass's entries (and, WIN = false, their x)
    Set sa, sb, sc;
    auto load_pass = [&](uint32_t p, Set &q) {
        const uint32_t last = e1 > e0 ? e1 - 1u : 0u;
#pragma unroll
        for (uint32_t k = 0; k < kBwItems; ++k) {
            const uint32_t idx = min(e0 + min(p, npass) * kBwPass + k * kBwThreads + t, last);
            q.c[k] = colind[idx];
            q.v[k] = vals[idx];
        }
    };
    auto gather = [&](Set &q) {   // (WIN = false) x of the set's entries, through the vector memory path
#pragma unroll
        for (uint32_t k = 0; k < kBwItems; ++k) q.xg[k] = x[q.c[k]];
    };
    load_pass(0, sa);   // (a block without entries reads the matrix's first entry: unconditional, like every load here)
    load_pass(1, sb);
    if (!WIN) load_pass(2, sc);
    if (PRE) {
#pragma unroll
        for (uint32_t k = 0; k < kBwRpRegs; ++k)
            if (k * kBwThreads + t <= nr) s_rp[k * kBwThreads + t] = rpr[k];
    } else {
        for (uint32_t i = t; i <= nr; i += kBwThreads) s_rp[i] = rowptr[r0 + i];
    }
    if (t < 6u) s_n[t] = 0u;
    const uint32_t c0 = rec0.w;             // first column of the window (a multiple of 256)
    if (PRE) {   // the window is in registers (requested a block ago, or at the kernel's start)
        constexpr uint32_t V = 16 / sizeof(T);
        const uint32_t wn = min(win_n, ncols - min(c0, ncols));
        const uint32_t nvec = wn / V;
        u4 *xd = reinterpret_cast<u4 *>(xw);
#pragma unroll
        for (uint32_t k = 0; k < kBwWinRegs; ++k)
            if (k * kBwThreads + t < nvec) xd[k * kBwThreads + t] = wr[k];
        for (uint32_t i = nvec * V + t; i < wn; i += kBwThreads) xw[i] = x[c0 + i];   // (a window that ends inside a vector)
    } else if (WIN) {
        const uint32_t wn = min(win_n, ncols - min(c0, ncols));
        if ((reinterpret_cast<uintptr_t>(x + c0) & 15u) == 0) {   // (uniform) 16-byte loads, kBwWinLoads of them in flight per thread
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            constexpr uint32_t V = 16 / sizeof(T);
            const u4 *xs = reinterpret_cast<const u4 *>(x + c0);
            u4 *xd = reinterpret_cast<u4 *>(xw);
            const uint32_t nvec = wn / V;
            for (uint32_t i0 = 0; i0 < nvec; i0 += kBwWinLoads * kBwThreads) {   // (uniform)
                u4 r[kBwWinLoads];
#pragma unroll
                for (uint32_t k = 0; k < kBwWinLoads; ++k) r[k] = xs[min(i0 + k * kBwThreads + t, nvec - 1u)];
#pragma unroll
                for (uint32_t k = 0; k < kBwWinLoads; ++k)
                    if (i0 + k * kBwThreads + t < nvec) xd[i0 + k * kBwThreads + t] = r[k];
            }
            for (uint32_t i = nvec * V + t; i < wn; i += kBwThreads) xw[i] = x[c0 + i];
        } else {
            for (uint32_t i0 = 0; i0 < wn; i0 += kBwWinLoads * kBwThreads) {
                T r[kBwWinLoads];
#pragma unroll
                for (uint32_t k = 0; k < kBwWinLoads; ++k) r[k] = x[c0 + min(i0 + k * kBwThreads + t, wn - 1u)];
#pragma unroll
                for (uint32_t k = 0; k < kBwWinLoads; ++k)
                    if (i0 + k * kBwThreads + t < wn) xw[i0 + k * kBwThreads + t] = r[k];
            }
        }
    }
    __syncthreads();
    BW_STAMP(0);
    if (PRE) prefetch_block(rec1, win_n1);   // (the next block's -- this block's again when there is none: unconditional)
    for (uint32_t i = t; i < nr; i += kBwThreads)   // empty rows: nothing below writes them
        if (s_rp[i + 1] == s_rp[i]) y[r0 + i] = T(0);
    if (e1 != e0) {   // block-uniform (a block without entries: its rows are zeroed, that is all)

    // the products of pass p, into strip p & 1 (all gathers first: window and strip are both LDS, a write between two reads
    // keeps them in order; written unconditionally: places beyond the pass's end are never read)
    auto products = [&](uint32_t p, const Set &q) {
        T *strip = sp + (p & 1u) * kBwPass;
        T xv[kBwItems];
#pragma unroll
        for (uint32_t k = 0; k < kBwItems; ++k) xv[k] = WIN ? xw[q.c[k] - c0] : q.xg[k];
#pragma unroll
        for (uint32_t k = 0; k < kBwItems; ++k) strip[k * kBwThreads + t] = q.v[k] * xv[k];   // one rounding, as `val * x[col]` in the reference
    };
    // the rows lo ... hi - 1 may hold entries of pass p = [ps, pe): those of more than kBwShort entries go on pass p's lists
    auto classify = [&](uint32_t p, uint32_t lo, uint32_t hi) {
        const uint32_t ps = e0 + p * kBwPass, pe = min(ps + kBwPass, e1);
        uint32_t *lst = s_list + (p % 3u) * kBwListCap, *cnt = s_n + (p % 3u) * 2u;
        for (uint32_t i = lo + t; i < hi; i += kBwThreads) {
            const uint32_t rs = s_rp[i], re = s_rp[i + 1];
            const uint32_t a = max(rs, ps), b = min(re, pe);
            if (a < b && re - rs > kBwShort) {                  // (at most kBwLongCap / kBwHeavyCap of them touch a pass)
                if (b - a > kBwHeavy) lst[kBwLongCap + atomicAdd(&cnt[1], 1u)] = i;
                else lst[atomicAdd(&cnt[0], 1u)] = i;
            }
        }
    };

    // prologue: pass 0's products and lists
    uint32_t cur_lo = 0, cur_hi = bw_first_at_least(s_rp, 0u, nr, min(e0 + kBwPass, e1), lane);
    if (!WIN) { gather(sa); gather(sb); }
    products(0, sa);
    classify(0, cur_lo, cur_hi);
    __syncthreads();
    BW_STAMP(1);
    // One PHASE per pass, one barrier per phase: while pass p's rows are summed out of strip p & 1, pass p + 1's products go
    // into the other strip (their entries were requested a phase ago), its rows are looked at and listed, and pass p + 2's
    // entries are requested.  (First form: products | barrier | rows looked at | barrier | rows summed | barrier per pass.)
    auto phase = [&](uint32_t p, Set &s1, Set &s2, Set &s3) {
        // WIN: s1 = pass p + 1's entries, in registers since the last phase; s2 = free, takes pass p + 2's (s3 unused).
        // else: s1 = pass p + 1's entries and their x; s2 = pass p + 2's entries, their x is gathered now; s3 takes pass p + 3's.
        const uint32_t ps = e0 + p * kBwPass, pe = min(ps + kBwPass, e1), parity = p & 1u;
        const T *strip = sp + parity * kBwPass;
        uint32_t nxt_lo = 0, nxt_hi = 0;
        products(p + 1u, s1);               // (beyond the last pass: of the last entry again, into the strip nobody reads)
        if (WIN) {
            load_pass(p + 2u, s2);
        } else {
            gather(s2);
            load_pass(p + 3u, s3);
        }
        if (p + 1u < npass) {               // (block-uniform)
            nxt_lo = s_rp[cur_hi] == pe ? cur_hi : cur_hi - 1u;   // the row that holds entry pe (cur_hi >= 1: rp[0] = e0 < pe)
            nxt_hi = bw_first_at_least(s_rp, nxt_lo, nr, min(pe + kBwPass, e1), lane);
            classify(p + 1u, nxt_lo, nxt_hi);
        }
        if (t == 0) { s_n[((p + 2u) % 3u) * 2u] = 0u; s_n[((p + 2u) % 3u) * 2u + 1u] = 0u; }   // (last read a phase ago)
        const uint32_t *lst = s_list + (p % 3u) * kBwListCap, *cnt = s_n + (p % 3u) * 2u;
        // (pass p's lists are complete since the last barrier; their lengths are read HERE: the next phase's thread 0 zeroes
        //  them for pass p + 3 while a slower thread may still be on its way out of this phase)
        const uint32_t nlong = cnt[0], nheavy = cnt[1];
        // the first kBwRowThreads threads take a short row each (a pass of 4096 entries touches ~400 rows of ten entries; it
        // lasts as long as its longest row), the other waves the listed rows, 16 lanes per row
        if (t < kBwRowThreads) {
            for (uint32_t i = cur_lo + t; i < cur_hi; i += kBwRowThreads) {
                const uint32_t rs = s_rp[i], re = s_rp[i + 1];
                const uint32_t a = max(rs, ps), b = min(re, pe);
                if (a >= b || re - rs > kBwShort) continue;         // empty, ended where the pass begins, or listed
                T acc = rs < ps ? s_carry[parity ^ 1u] : T(0);      // a cut row goes on where the last pass stopped
                const uint32_t jb = b - ps;
                uint32_t j = a - ps;
                for (; j + 4u <= jb; j += 4u) {   // (eight requested at once and added under predicates: 7.4 -> 9.4 us per block)
                    const T v0 = strip[j], v1 = strip[j + 1], v2 = strip[j + 2], v3 = strip[j + 3];
                    acc = acc + v0;
                    acc = acc + v1;
                    acc = acc + v2;
                    acc = acc + v3;
                }
                for (; j < jb; ++j) acc = acc + strip[j];
                if (re <= pe) y[r0 + i] = acc;
                else s_carry[parity] = acc;
            }
        } else {
            // rows with more than kBwHeavy entries in this pass first: a whole wave each (two running sums per lane)
            for (uint32_t hi = wave - kBwRowThreads / 64u; hi < nheavy; hi += (kBwThreads - kBwRowThreads) / 64u) {   // wave-uniform
                const uint32_t i = lst[kBwLongCap + hi];
                const uint32_t rs = s_rp[i], re = s_rp[i + 1];
                const uint32_t a = max(rs, ps) - ps, b = min(re, pe) - ps;
                T p0 = T(0), p1 = T(0);
                uint32_t j = a + lane;
                for (; j + 64u < b; j += 128u) { p0 = p0 + strip[j]; p1 = p1 + strip[j + 64u]; }
                if (j < b) p0 = p0 + strip[j];
                const T part = bw_sum64(p0 + p1);
                if (lane == 0) {
                    const T tot = (rs < ps ? s_carry[parity ^ 1u] : T(0)) + part;
                    if (re <= pe) y[r0 + i] = tot;
                    else s_carry[parity] = tot;
                }
            }
            const uint32_t g = (t - kBwRowThreads) >> 4, gl = t & 15u;
            for (uint32_t li = g; li < nlong; li += kBwGroups) {
                const uint32_t i = lst[li];
                const uint32_t rs = s_rp[i], re = s_rp[i + 1];
                const uint32_t a = max(rs, ps) - ps, b = min(re, pe) - ps;
                T p0 = T(0), p1 = T(0);
                uint32_t j = a + gl;
                for (; j + 16u < b; j += 32u) { p0 = p0 + strip[j]; p1 = p1 + strip[j + 16u]; }
                if (j < b) p0 = p0 + strip[j];
                const T part = bw_sum16(p0 + p1);
                if (gl == 0) {
                    const T tot = (rs < ps ? s_carry[parity ^ 1u] : T(0)) + part;
                    if (re <= pe) y[r0 + i] = tot;
                    else s_carry[parity] = tot;
                }
            }
        }
        __syncthreads();
        BW_STAMP(2);
        BW_STAMP(3);
        cur_lo = nxt_lo;
        cur_hi = nxt_hi;
    };
    if (WIN) {
        for (uint32_t p = 0; p < npass; p += 2u) {   // (block-uniform) two register sets take turns
            phase(p, sb, sa, sc);
            if (p + 1u < npass) phase(p + 1u, sa, sb, sc);
        }
    } else {
        for (uint32_t p = 0; p < npass; p += 3u) {   // (block-uniform) three register sets take turns
            phase(p, sb, sc, sa);
            if (p + 1u < npass) phase(p + 1u, sc, sa, sb);
            if (p + 2u < npass) phase(p + 2u, sa, sb, sc);
        }
    }
    } else {
        __syncthreads();   // (s_rp and the counters are written again by the next block)
    }
#ifdef SPAL_BW_STAMPS
    ++nblk_;
#endif
    if (!has_next) break;
    rec0 = rec1;
    win_n = win_n1;
    slot = nslot;
  }
#ifdef SPAL_BW_STAMPS
    if (t == 0 && g_bw_stamps) {
        for (int i = 0; i < 4; ++i) g_bw_stamps[(size_t)blockIdx.x * 8 + i] = acc_[i];
        g_bw_stamps[(size_t)blockIdx.x * 8 + 6] = nblk_;
        g_bw_stamps[(size_t)blockIdx.x * 8 + 7] = first_;
    }
#endif
}

static size_t bw_lds_bytes(uint32_t RB, uint32_t win_cols, size_t esz) {
    return (size_t)win_cols * esz + 2 * (size_t)kBwPass * esz + 2 * esz + (size_t)(RB + 2 + 3 * kBwListCap + 6) * 4;
}

void blockwin_free(spal_csr *a) {
    a->bw_on = 0;
    (void)dev_free(a->d_bworder); a->d_bworder = nullptr;
    a->bw_blocks = 0; a->bw_rows = 0; a->bw_cols = 0;
}

// Measures the units' windows, picks the tallest block whose widest window fits LDS; leaves a->bw_rows = 0 when none does
// (or the matrix is too small to be worth a 1024-thread workgroup per block).
int blockwin_plan(spal_csr *a) {
    blockwin_free(a);
    if (a->nnz == 0 || a->nrows < kBwUnit || !a->parts.empty()) return SPAL_OK;
    const uint32_t nunits = (uint32_t)((a->nrows + kBwUnit - 1) / kBwUnit);
    DevBuf d_win, d_first;
    SPAL_HIP_TRY(d_win.alloc((size_t)nunits * sizeof(uint2)));
    SPAL_HIP_TRY(d_first.alloc((size_t)(nunits + 1) * 4));
    hipLaunchKernelGGL(bw_unit_windows, dim3(nunits), dim3(256), 0, a->stream, a->d_rowptr, a->d_colind, (uint32_t)a->nrows,
                       d_win.as<uint2>(), d_first.as<uint32_t>());
    SPAL_HIP_TRY(hipGetLastError());
    std::vector<uint2> win(nunits);
    std::vector<uint32_t> first(nunits + 1);
    SPAL_HIP_TRY(hipMemcpyAsync(win.data(), d_win.p, win.size() * sizeof(uint2), hipMemcpyDeviceToHost, a->stream));
    SPAL_HIP_TRY(hipMemcpyAsync(first.data(), d_first.p, first.size() * 4, hipMemcpyDeviceToHost, a->stream));
    SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
    const size_t esz = (size_t)a->elem_size;
    int cu_count = 0;
    SPAL_HIP_TRY(hipDeviceGetAttribute(&cu_count, hipDeviceAttributeMultiprocessorCount, a->device));
    const uint32_t cus = (uint32_t)std::max(cu_count, 8);
    // units = 8, 4, 2, 1 with the window in LDS; when none fits (columns anywhere): blocks of 2048 rows, x gathered from memory.
    // The TALLEST block that fits stages its window for the most entries -- but the walks want three blocks per CU and more
    // (250 000 rows of 400 entries in blocks of 4096 rows are 61 blocks for 256 CUs: 0.16 of the roofline): the tallest height with
    // that many blocks, else the lowest that fits.
    auto fits = [&](uint32_t units, uint32_t *nb_out) {
        const uint32_t RB = units * kBwUnit, nb = (uint32_t)((a->nrows + RB - 1) / RB);
        uint32_t widest = 0;
        for (uint32_t b = 0; b < nb; ++b) {
            uint32_t lo = 0xffffffffu, hi = 0;
            for (uint32_t u = b * units; u < std::min(nunits, (b + 1) * units); ++u) { lo = std::min(lo, win[u].x); hi = std::max(hi, win[u].y); }
            if (hi > lo) widest = std::max(widest, hi - (lo & ~255u));
        }
        const uint32_t win_cols = std::max(256u, (widest + 255u) & ~255u);
        *nb_out = nb;
        return bw_lds_bytes(RB, win_cols, esz) <= kBwLdsMax && (size_t)win_cols * esz <= (size_t)kBwWinRegs * 16 * kBwThreads &&
               RB + 1 <= kBwRpRegs * kBwThreads;
    };
    int first_attempt = 4;   // (4: the window-less form)
    for (int at = 0; at < 4; ++at) {
        uint32_t nb = 0;
        if (!fits(8u >> at, &nb)) continue;
        first_attempt = at;                  // the lowest height that fits so far ...
        if (nb >= 3u * cus) break;           // ... and the tallest with blocks enough
    }
    for (int attempt = first_attempt; attempt < 5; ++attempt) {
        const bool windowed = attempt < 4;
        const uint32_t units = windowed ? (8u >> attempt) : 4u;
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
        const uint32_t win_cols = windowed ? std::max(256u, (widest + 255u) & ~255u) : 0u;
        if (bw_lds_bytes(RB, win_cols, esz) > kBwLdsMax) continue;
        if ((size_t)win_cols * esz > (size_t)kBwWinRegs * 16 * kBwThreads || RB + 1 > kBwRpRegs * kBwThreads) continue;   // (what a walk prefetches)
        const uint32_t per_xcd = (nb + 7u) / 8u;
        const uint32_t J = std::max(1u, std::min(per_xcd, cus / 8u));   // resident workgroups per XCD
        // Every XCD's eighth of the blocks is dealt to its J walks longest first, each block to the walk with the least work so
        // far (a block costs its entries + what changing blocks costs, ~8000 entries' worth): the walks end within a block's
        // fraction of each other.  Walk j's r-th block is record (r * J + j) * 8 + XCD; a walk ends at its first padded record.
        auto entries = [&](uint32_t b) { return first[std::min(nunits, (b + 1) * units)] - first[b * units]; };
        std::vector<std::vector<uint32_t>> walks((size_t)8u * J);
        std::vector<uint32_t> part;
        uint32_t rows = 1;
        for (uint32_t k = 0; k < 8u; ++k) {
            part.clear();
            for (uint32_t b = k * per_xcd; b < std::min(nb, (k + 1) * per_xcd); ++b) part.push_back(b);
            std::stable_sort(part.begin(), part.end(), [&](uint32_t p, uint32_t q) { return entries(p) > entries(q); });
            std::vector<uint64_t> load(J, 0);
            for (uint32_t b : part) {
                const uint32_t j = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
                load[j] += (uint64_t)entries(b) + 8000u;
                walks[(size_t)k * J + j].push_back(b);
                rows = std::max<uint32_t>(rows, (uint32_t)walks[(size_t)k * J + j].size());
            }
        }
        std::vector<uint32_t> order((size_t)rows * J * 8u, 0xffffffffu), rec((size_t)rows * J * 8u * 8u, 0xffffffffu);
        for (uint32_t k = 0; k < 8u; ++k)
            for (uint32_t j = 0; j < J; ++j)
                for (size_t r = 0; r < walks[(size_t)k * J + j].size(); ++r) order[(r * J + j) * 8u + k] = walks[(size_t)k * J + j][r];
        for (size_t i = 0; i < order.size(); ++i) {   // 32-byte records in dealing order
            const uint32_t b = order[i];
            if (b == 0xffffffffu) continue;
            uint32_t *r = &rec[i * 8u];
            r[0] = b; r[1] = first[b * units]; r[2] = first[std::min(nunits, (b + 1) * units)]; r[3] = bw[b].x; r[4] = bw[b].y;
            r[5] = r[6] = r[7] = 0u;
        }
        SPAL_HIP_TRY(dev_alloc((void **)&a->d_bworder, rec.size() * 4));
        SPAL_HIP_TRY(hipMemcpy(a->d_bworder, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
        a->bw_blocks = nb;
        a->bw_rows = RB;
        a->bw_cols = win_cols;
        a->bw_grid = 8u * J;
        a->bw_list_rows = rows * J;
        return SPAL_OK;
    }
    return SPAL_OK;
}

template <typename T, bool WIN, bool PRE>
static hipError_t bw_launch_t(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    static std::atomic<uint64_t> configured{0};   // devices on which this instantiation's LDS cap has been raised
    const uint64_t bit = 1ull << (a->device & 63);
    if (!(configured.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(csr_spmv_blockwin<T, WIN, PRE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwLdsMax);
        if (e != hipSuccess) return e;
        configured.fetch_or(bit, std::memory_order_relaxed);
    }
    const size_t lds = bw_lds_bytes(a->bw_rows, a->bw_cols, sizeof(T));
#ifdef SPAL_BW_STAMPS
    static unsigned long long *d_st = nullptr;
    static uint32_t st_for = 0, calls = 0;
    if (st_for < a->bw_blocks) {
        if (d_st) (void)hipFree(d_st);
        if (hipMalloc((void **)&d_st, (size_t)a->bw_blocks * 64 + 4096 * 64) != hipSuccess) return hipErrorOutOfMemory;
        st_for = a->bw_blocks;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bw_stamps), &d_st, sizeof(d_st));
    }
    (void)hipMemsetAsync(d_st, 0, (size_t)a->bw_grid * 64, st);
#endif
    hipLaunchKernelGGL((csr_spmv_blockwin<T, WIN, PRE>), dim3(a->bw_grid), dim3(kBwThreads), lds, st, a->d_rowptr, a->d_colind,
                       (const T *)a->d_values, (const T *)x, (T *)y, a->d_bworder, (uint32_t)a->nrows, (uint32_t)a->ncols,
                       a->bw_rows, a->bw_blocks, a->bw_cols, a->bw_list_rows);
#ifdef SPAL_BW_STAMPS
    if (++calls % 16 == 0) {
        std::vector<unsigned long long> h((size_t)a->bw_grid * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
        double sum[4] = {0, 0, 0, 0}, tot_max = 0, tot_min = 1e30, nb = 0;
        unsigned long long first = ~0ull, last = 0;
        for (uint32_t b = 0; b < a->bw_grid; ++b) {
            if (!h[(size_t)b * 8 + 6]) continue;
            double tot = 0;
            for (int i = 0; i < 4; ++i) { sum[i] += (double)h[(size_t)b * 8 + i]; tot += (double)h[(size_t)b * 8 + i]; }
            tot_max = std::max(tot_max, tot); tot_min = std::min(tot_min, tot);
            nb += (double)h[(size_t)b * 8 + 6];
            first = std::min(first, h[(size_t)b * 8 + 7]);
            last = std::max(last, h[(size_t)b * 8 + 7] + (unsigned long long)tot);
        }
        fprintf(stderr, "[spal blockwin stamps] %.0f blocks by %u workgroups, kernel %.1f us; mean us per block: window + first loads %.2f, first pass's "
                "products %.2f, phases %.2f, (unused) %.2f; a workgroup's walk: %.1f ... %.1f us\n", nb, a->bw_grid, (double)(last - first) / 100.0,
                sum[0] / nb / 100.0, sum[1] / nb / 100.0, sum[2] / nb / 100.0, sum[3] / nb / 100.0, tot_min / 100.0, tot_max / 100.0);
    }
#endif
    return hipGetLastError();
}

hipError_t blockwin_launch(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    if (a->bw_cols) {
        if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0)   // (the windows begin at multiples of 256 columns)
            return a->elem_size == 8 ? bw_launch_t<double, true, true>(a, x, y, st) : bw_launch_t<float, true, true>(a, x, y, st);
        return a->elem_size == 8 ? bw_launch_t<double, true, false>(a, x, y, st) : bw_launch_t<float, true, false>(a, x, y, st);
    }
    return a->elem_size == 8 ? bw_launch_t<double, false, false>(a, x, y, st) : bw_launch_t<float, false, false>(a, x, y, st);
}

}  // namespace spal
