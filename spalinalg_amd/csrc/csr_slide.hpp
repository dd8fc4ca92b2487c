// csr_slide.hpp -- the sliding-window form of the CSR stream kernel (gfx950).
//
// Same tiles, same arithmetic and the same bit-identical results as csr_spmv_stream (csr_kernels.hpp); what changes
// is how the x window gets into LDS and how the loads are counted.
//
//  1. csr_spmv_stream stages the whole window of a 1024-row super-tile (40 KB for config 3) behind a barrier before
//     its first product: ablation (profiles/r02) prices that at 10 % of the launch, although nine tenths of the
//     staged columns were in LDS for the previous super-tile already -- on another CU.  Here a fixed grid of
//     workgroups (two per CU) each walks a contiguous run of STEPS of 4 x RPT rows (one tile per wave), and the
//     window is a RING of NP pages in LDS (slot = page mod NP, which is also how col16 addresses it): going from one
//     step to the next only the pages that ENTER the window are loaded -- one 2 KB page per 256 rows of a band
//     instead of 36 KB -- and they are loaded asynchronously: requested at the start of step i, written to their
//     (free) ring slots at its end, published by the one barrier that separates the steps.
//  2. csr_spmv_stream issues a tile's loads under `if (j < steps)`; the number of loads in flight is then unknown
//     to the compiler, which waits for ALL of them (s_waitcnt vmcnt(0)) before the first product -- including the
//     next tile's, issued a moment earlier: no wave ever computed while its own loads were in flight.  Here every
//     tile issues exactly S value loads + S column loads (S = the plan's largest tile, a template parameter; a
//     smaller tile re-reads its last step, which L2 serves) and, when rows differ in length, 2 row-pointer loads:
//     the waits become counted (vmcnt(N)) and a wave keeps PF tiles in flight while it multiplies and sums.
//
// Eligibility (csr_plan_build): every super-tile streams from LDS with a contiguous run of pages (bands, block
// diagonals and similar structure), 64 / 32 / 16 / 8 rows per tile, no skewed strips; x 16-byte aligned (checked at
// launch: the kernels of csr_kernels.hpp read the same ring-encoded col16).  Everything else keeps those kernels.
#pragma once
#include "csr_kernels.hpp"

namespace spal {

// step descriptor: .x = first page of the step's window, .y = npages (8 bits) | skip << 8 (4 bits: tiles left to
// csr_spmv_overflow) | flags << 16
constexpr uint32_t kSlideAsync = 1u << 16;   // the pages entering with this step fit beside the previous step's, are at most
                                             // kSlideAsyncVecs * 256 vectors and hold no partial vector of x: prefetched
constexpr uint32_t kSlideAsyncVecs = 1;      // 16-byte vectors of entering pages a thread holds (2 pages of f64 per step)
#ifndef SPAL_SLIDE_FIRST_BATCH
#define SPAL_SLIDE_FIRST_BATCH 12
#endif
constexpr uint32_t kSlideFirstBatch = SPAL_SLIDE_FIRST_BATCH;   // vectors of the first window a thread requests at once

template <typename T>
struct SlideVec { using type = __attribute__((ext_vector_type(4))) uint32_t; };

// vector v (16 bytes) of page p of x; `last_vec` = index of the last whole vector of x: the pages' tail past it reads
// that vector again (columns there are never referenced)
template <typename T>
__device__ __forceinline__ typename SlideVec<T>::type slide_load_vec(const T *__restrict__ x, uint32_t page, uint32_t v,
                                                                      uint32_t last_vec) {
    using vec_t = typename SlideVec<T>::type;
    constexpr uint32_t VP = kPageCols / (16 / sizeof(T));
    return reinterpret_cast<const vec_t *>(x)[min(page * VP + v, last_vec)];
}

// ... and for the synchronous paths (first window of a run, steps the plan did not mark kSlideAsync): x may end
// inside the page, and in the middle of a vector when ncols is not a multiple of it
template <typename T>
__device__ __forceinline__ typename SlideVec<T>::type slide_load_vec_tail(const T *__restrict__ x, uint32_t page, uint32_t v,
                                                                           uint32_t ncols) {
    using vec_t = typename SlideVec<T>::type;
    constexpr uint32_t V = 16 / sizeof(T);
    const uint32_t e = page * kPageCols + v * V;
    if (e + V <= ncols) return *reinterpret_cast<const vec_t *>(x + e);
    vec_t r = {0u, 0u, 0u, 0u};
    T *rt = reinterpret_cast<T *>(&r);
#pragma unroll
    for (uint32_t q = 0; q < V; ++q)
        if (e + q < ncols) rt[q] = x[e + q];
    return r;
}

// the pages of window B = [fb, fb + nb) that are not in window A = [fa, fa + na): a low and a high run
struct SlideNew { uint32_t lo0, lo1, hi0, hi1; };
__device__ __forceinline__ SlideNew slide_new_pages(uint32_t fa, uint32_t na, uint32_t fb, uint32_t nb) {
    const uint32_t ea = fa + na, eb = fb + nb;
    SlideNew r;
    if (fa >= eb || ea <= fb) {   // disjoint: all of B, as one run
        r.lo0 = fb; r.lo1 = eb; r.hi0 = eb; r.hi1 = eb;
        return r;
    }
    r.lo0 = fb; r.lo1 = max(fb, min(fa, eb));     // [fb, fa) when B starts below A
    r.hi0 = min(eb, max(ea, fb)); r.hi1 = eb;     // [ea, eb) when B ends above A
    return r;
}

// One tile's loads, exactly S + S (+ 2) of them whatever the tile holds.
template <typename T, int RPT, int S, bool UNI>
__device__ __forceinline__ void slide_tile_load(StreamTile<T> &t, const uint32_t *__restrict__ rowptr,
                                                const uint16_t *__restrict__ col16, const T *__restrict__ vals,
                                                uint32_t row0, uint32_t nrows, uint32_t b, uint32_t e, uint32_t lane,
                                                uint32_t ulen) {
    using pair_t = typename Pair<T>::type;
    static_assert(RPT <= 64, "a row per lane");
    const uint32_t rlast = min(row0 + (uint32_t)RPT, nrows);
    t.start = b & ~1u;
    t.steps = (e - t.start + 127u) >> 7;          // <= S by the plan
    const uint32_t e0 = t.start + lane * 2;
    const uint32_t jmax = max(t.steps, 1u) - 1u;  // steps past the tile's last re-read it (an L2 hit, no new bytes)
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const uint32_t at = e0 + min((uint32_t)j, jmax) * 128u;
        t.v[j] = __builtin_nontemporal_load(reinterpret_cast<const pair_t *>(vals + at));
        t.c[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(col16 + at));
    }
    if constexpr (UNI) {     // every row of the matrix's streamed tiles holds ulen - 1 entries: rowptr is not read
        const uint32_t len = ulen - 1u, nrt = rlast - min(row0, rlast);
        t.rp0 = b + min(lane, nrt) * len;
        t.rp1 = b + min(lane + 1u, nrt) * len;
    } else {
        t.rp0 = rowptr[min(row0 + lane, rlast)];
        t.rp1 = rowptr[min(row0 + lane + 1u, rlast)];
    }
}

template <typename T, int RPT, int S, bool UNI, int PF>
__global__ __launch_bounds__(kStreamBlock, 2) void csr_spmv_slide(
    const uint32_t *__restrict__ rowptr, const uint16_t *__restrict__ col16, const T *__restrict__ vals,
    const T *__restrict__ x, T *__restrict__ y, const uint2 *__restrict__ sdesc, uint32_t nrows, uint32_t ncols,
    uint32_t nsteps, uint32_t per_xcd, uint32_t chunk, uint32_t NP, uint32_t ulen, uint32_t flags) {
    static_assert(PF == 1 || PF == 2, "one or two tiles of loads ahead");
    static_assert(S >= 1 && S <= kStreamSteps, "a tile holds at most kStreamSteps steps");
    using vec_t = typename SlideVec<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char spal_smem[];
    constexpr uint32_t V = 16 / sizeof(T), VP = kPageCols / V;
    constexpr int NB = PF + 1;
    const bool nt_store = flags & 1u;
    T *prod_all = reinterpret_cast<T *>(spal_smem);
    T *xw = prod_all + kStreamWaves * stream_strip<false>();
    vec_t *xw4 = reinterpret_cast<vec_t *>(xw);

    // this workgroup's steps: its XCD owns a contiguous range of steps, cut into RUNS of `chunk` steps that are dealt
    // round-robin to the XCD's workgroups (blockIdx >> 3 = slot).  Inside a run the window slides; a new run stages
    // its first window whole.  (One run per workgroup = chunk >= steps per XCD / slots is the fully persistent form;
    // shorter runs keep the workgroups of an XCD on neighbouring memory -- one front instead of 64.)
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const bool even = flags & 8u;
    const uint32_t xbase = even ? xcd * (nsteps / 8u) + min(xcd, nsteps % 8u) : xcd * per_xcd;
    if (xbase >= nsteps) return;
    const uint32_t nx = even ? nsteps / 8u + (xcd < nsteps % 8u ? 1u : 0u) : min(per_xcd, nsteps - xbase);   // steps of this XCD
    if (nx == 0u) return;
    // (flags bit 3, one run per workgroup: the steps are split EVENLY over the XCDs and an XCD's over all its workgroups --
    //  nx / slots steps each, the first nx % slots one more -- instead of runs of ceil(nx / slots): a 1M-row shard has 489
    //  steps per XCD, which as runs of 8 kept 62 of the 64 workgroups of an XCD busy)
    const uint32_t nruns = even ? min(slots, nx) : (nx + chunk - 1u) / chunk;
    if (slot >= nruns) return;
    const uint32_t myruns = even ? 1u : (nruns - slot + slots - 1u) / slots;
    // steps of this workgroup: full runs, except that the XCD's last run may be short and is then this one's last
    const uint32_t lastrun = slot + (myruns - 1u) * slots;
    const uint32_t even_lo = nx / slots, even_rem = nx % slots;
    const uint32_t even_start = slot * even_lo + min(slot, even_rem);
    const uint32_t nk = even ? even_lo + (slot < even_rem ? 1u : 0u) : (myruns - 1u) * chunk + min(chunk, nx - lastrun * chunk);
    // k-th step of this workgroup -> its global step index.  k past the end: the matrix's first step -- its tile is
    // what every workgroup's last PF load slots then read (the count of loads must not depend on the path), so it
    // stays in L2; re-reading the workgroup's own last tile cost 37 MB of HBM reads per launch (streaming loads
    // are not kept).
    auto gi = [&](uint32_t k) {
        return k >= nk ? 0u : even ? xbase + even_start + k : xbase + (slot + (k / chunk) * slots) * chunk + k % chunk;
    };

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);   // scalar: tile bounds come by s_load
    T *prod = prod_all + wave * stream_strip<false>();
    const uint32_t wmax = NP * kPageCols - 1u;
    const uint32_t last_vec = ncols / V - 1u;     // (ncols >= kPageCols by the plan)

    // rows / entry range of this wave's tile at the workgroup's k-th step; steps past its last re-read that one's tile
    // (same count of loads on every path -- that is the point -- and nobody uses them)
    auto tile_row = [&](uint32_t k) { return (gi(k) * kStreamWaves + wave) * (uint32_t)RPT; };
    // (flags bit 2, UNI plans only: EVERY row of the matrix holds ulen - 1 entries, so rowptr[r] = r * (ulen - 1) and a
    //  workgroup's first tile loads do not wait for a round trip to rowptr -- at a shard's 7.6 steps per workgroup that is
    //  a tenth of the launch, profiles/r04/shard_sized_launches.txt)
    const bool arith = UNI && (flags & 4u);
    auto tile_b = [&](uint32_t k) { const uint32_t r = min(tile_row(k), nrows); return arith ? r * (ulen - 1u) : rowptr[r]; };
    auto tile_e = [&](uint32_t k) { const uint32_t r = min(tile_row(k) + (uint32_t)RPT, nrows); return arith ? r * (ulen - 1u) : rowptr[r]; };

    StreamTile<T> t[NB];
    uint32_t tb[NB], te[NB];     // entry ranges of the tiles whose loads go out next (asked for a step early, by s_load)
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        tb[q] = tile_b((uint32_t)q);
        te[q] = tile_e((uint32_t)q);
    }
    uint2 d = sdesc[gi(0u)];
    uint2 dn = sdesc[gi(1u)];
#pragma unroll
    for (int q = 0; q < PF; ++q)
        slide_tile_load<T, RPT, S, UNI>(t[q], rowptr, col16, vals, tile_row((uint32_t)q), nrows, tb[q], te[q], lane, ulen);
    // the first window, whole: kSlideFirstBatch vectors per thread requested back to back (12: the 24 pages of an f64 window
    // in ONE round trip; four at a time made it three, and a shard-sized launch is only 7.6 steps per workgroup long)
    {
        const uint32_t f = d.x, nv = (d.y & 0xffu) * VP;
        for (uint32_t j0 = threadIdx.x; j0 < nv; j0 += kSlideFirstBatch * kStreamBlock) {
            vec_t r[kSlideFirstBatch];
#pragma unroll
            for (uint32_t k = 0; k < kSlideFirstBatch; ++k) {
                const uint32_t j = min(j0 + k * kStreamBlock, nv - 1u);
                r[k] = slide_load_vec_tail<T>(x, f + j / VP, j % VP, ncols);
            }
#pragma unroll
            for (uint32_t k = 0; k < kSlideFirstBatch; ++k) {
                const uint32_t j = j0 + k * kStreamBlock;
                if (j < nv) xw4[((f + j / VP) % NP) * VP + j % VP] = r[k];
            }
        }
    }
    __syncthreads();

    // ---- the workgroup's k-th step; K = k % NB is the slot of its tile
    auto step = [&](uint32_t i, auto kc) {
        constexpr int K = decltype(kc)::value;
        constexpr int KN = (K + PF) % NB;            // slot of the tile PF steps ahead: its loads go out now
        const uint32_t skip = (d.y >> 8) & 0xfu;
        const bool more = i + 1u < nk;               // block-uniform
        // pages that enter the window with step i + 1: requested first (loads return in order: what is written to LDS
        // at the end of this step must not queue behind the tile requested below), stored after this step's products.
        // Unconditional (a thread without a vector to fetch reads x[0 ...]): a counted wait again.
        SlideNew nw = slide_new_pages(d.x, d.y & 0xffu, dn.x, dn.y & 0xffu);
        const uint32_t nlo = (nw.lo1 - nw.lo0) * VP;
        const uint32_t ntot = more ? nlo + (nw.hi1 - nw.hi0) * VP : 0u;
        const bool async = more && (dn.y & kSlideAsync) && (even || (i + 1u) % chunk != 0u);   // (a new run's window is staged whole)
        auto new_page = [&](uint32_t j) { return j < nlo ? nw.lo0 + j / VP : nw.hi0 + (j - nlo) / VP; };
        vec_t nv[kSlideAsyncVecs];
#pragma unroll
        for (uint32_t k = 0; k < kSlideAsyncVecs; ++k) {
            const uint32_t j = threadIdx.x + k * kStreamBlock;
            const bool mine = async && j < ntot;
            nv[k] = slide_load_vec<T>(x, mine ? new_page(j) : 0u, mine ? j % VP : 0u, last_vec);
        }
        // the tile PF steps ahead
        slide_tile_load<T, RPT, S, UNI>(t[KN], rowptr, col16, vals, tile_row(i + PF), nrows, tb[KN], te[KN], lane, ulen);
        // ... and the entry range of the one after it (slot K is free: step i's loads went out PF steps ago)
        tb[K] = tile_b(i + PF + 1u);
        te[K] = tile_e(i + PF + 1u);
        // this step's tile
        const uint32_t r0 = (gi(i) * kStreamWaves + wave) * (uint32_t)RPT;
#ifdef SPAL_DIAG
        if (SPAL_DIAG_ON(flags, 11)) {   // ablation: no products, no sums -- the loads, the window and the barriers only
            T sacc = T(0);
#pragma unroll
            for (int j = 0; j < S; ++j) sacc += t[K].v[j].x + t[K].v[j].y + T(t[K].c[j]);
            if (r0 + lane < nrows) y[r0 + lane] = sacc + T(t[K].rp1 - t[K].rp0);
        } else
#endif
        if (r0 < nrows && !((skip >> wave) & 1u)) {
            stream_compute<T, RPT, false>(t[K], xw, wmax, prod, y, r0, nrows, lane, nt_store, flags);
        } else if (r0 < nrows && ((d.y >> (20u + wave)) & 1u)) {
            // More than the strip's 1024 entries, but each half of the rows fits (the plan checked): two passes
            // through the strip, rows [r0, rm) then [rm, re) -- same loads, products and left-to-right sums as any
            // tile, so these rows stay bit-identical too.  The slot's registers hold the tile's first 1024 entries,
            // which nobody needs: they take the halves.  (Rare: the loads are waited for in place.)
            const uint32_t re = min(r0 + (uint32_t)RPT, nrows), rm = min(r0 + (uint32_t)RPT / 2u, re);
            const uint32_t eb = rowptr[r0], em = rowptr[rm], ee = rowptr[re];
            slide_tile_load<T, RPT, S, false>(t[K], rowptr, col16, vals, r0, rm, eb, em, lane, 0u);
            stream_compute<T, RPT, false>(t[K], xw, wmax, prod, y, r0, rm, lane, nt_store, flags);
            if (rm < re) {
                __builtin_amdgcn_wave_barrier();   // the strip is read by the first half's sums until here
                slide_tile_load<T, RPT, S, false>(t[K], rowptr, col16, vals, rm, re, em, ee, lane, 0u);
                stream_compute<T, RPT, false>(t[K], xw, wmax, prod, y, rm, re, lane, nt_store, flags);
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < kSlideAsyncVecs; ++k) {
            const uint32_t j = threadIdx.x + k * kStreamBlock;
            if (async && j < ntot) xw4[(new_page(j) % NP) * VP + j % VP] = nv[k];
        }
        if (!SPAL_DIAG_ON(flags, 12)) __syncthreads();   // (ablation bit 12: no barrier between steps -- wrong results)
        if (more && !async && ntot) {
            // the entering pages would overwrite slots this step still read (or are too many to hold in registers):
            // load them now that every wave is done with the step
            for (uint32_t j0 = threadIdx.x; j0 < ntot; j0 += 4u * kStreamBlock) {
                vec_t r[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const uint32_t j = min(j0 + k * kStreamBlock, ntot - 1u);
                    r[k] = slide_load_vec_tail<T>(x, new_page(j), j % VP, ncols);
                }
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const uint32_t j = j0 + k * kStreamBlock;
                    if (j < ntot) xw4[(new_page(j) % NP) * VP + j % VP] = r[k];
                }
            }
            __syncthreads();
        }
        d = dn;
        dn = sdesc[gi(i + 2u)];
    };

    for (uint32_t i = 0; i < nk; i += NB) {
        step(i, std::integral_constant<int, 0>{});
        if (i + 1u < nk) step(i + 1u, std::integral_constant<int, 1>{});
        if constexpr (NB > 2)
            if (i + 2u < nk) step(i + 2u, std::integral_constant<int, 2>{});
    }
}

}  // namespace spal
