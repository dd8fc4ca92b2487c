// csr_kernels.hpp -- gfx950 kernels for y = A*x with A in CSR (32-bit indices).
//
// Semantics follow the reference-derived contract (SURVEY.md section 8a-1,
// reference src/csr/ops/mul.rs:25-45): y[i] = sum over the stored entries of
// row i of values[p] * x[colind[p]]; rows without entries give 0.0.  The GPU
// sums a row in a tree over L lanes (and may fuse mul+add), so results agree
// with the sequential CPU order to rounding (<= 1e-10 relative, tested), not
// bit for bit.
//
// The path is HBM-bandwidth bound (0.15 flop/byte): no MFMA.  What matters is
//  - values / colind streamed once, coalesced, non-temporal;
//  - the block's window of x staged in LDS so the gather never leaves the CU;
//  - row blocks dealt to XCDs in contiguous runs so neighbouring windows share
//    one L2;
//  - enough independent loads in flight per wave (U row groups per iteration).
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spal {

constexpr int kWave = 64;

// how a row block gets its x (desc[b].z): gathered from global memory, from an
// LDS-staged window, (stream kernel) LDS window + 16-bit relative columns, or
// (stream kernel, window too wide for LDS) x gathered from global with 32-bit columns
constexpr uint32_t kModeVectorGlobal = 0, kModeVectorLds = 1, kModeStream = 2, kModeStreamGlobal = 3;

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the
// blocks that share an XCD).  Map them so that each XCD owns one contiguous
// run of row blocks: used for L2 locality only, never for correctness.
__device__ __forceinline__ uint32_t xcd_contiguous_block(uint32_t bid, uint32_t per_xcd) {
    return (bid & 7u) * per_xcd + (bid >> 3);
}

// ... or, with the top bit of `per_xcd` set, interleaved in CHUNKS of C = per_xcd & 0x7fffffff consecutive row blocks
// (XCD x takes chunks x, x + 8, ...): neighbouring blocks still share an L2 inside a chunk, and all 8 XCDs work inside
// one moving window of 8 * C blocks instead of at 8 places an eighth of the arrays apart -- worth 5 % in every
// placement class in the footprint micro-benchmark (260 -> 245 us, 248 -> 235 us; profiles/r03/placement_pairs.txt).
// The grid covers ceil(nblocks / 8C) * 8C blocks; blocks past the last return.
constexpr uint32_t kXcdChunked = 0x80000000u;
__device__ __forceinline__ uint32_t xcd_block(uint32_t bid, uint32_t per_xcd) {
    if (per_xcd & kXcdChunked) {
        const uint32_t C = per_xcd & ~kXcdChunked, xcd = bid & 7u, slot = bid >> 3;
        return ((slot / C) * 8u + xcd) * C + slot % C;
    }
    return xcd_contiguous_block(bid, per_xcd);
}

template <typename T>
__device__ __forceinline__ T load_stream(const T *p) {
    return __builtin_nontemporal_load(p);
}

// Pins a loaded value so the compiler cannot sink its load into the (rarely
// false) branch that consumes it; costs no instruction.
__device__ __forceinline__ void keep_unconditional(double &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keep_unconditional(float &v) { asm volatile("" : "+v"(v)); }

// ---- reduction over the L lanes that share a row ----------------------------
// DPP row shifts move data inside 16-lane rows without touching LDS; the two
// steps that cross a 16-lane row (L = 32, 64) go through __shfl_down.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    int lo = (int)(uint32_t)u, hi = (int)(uint32_t)(u >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    int i = __builtin_bit_cast(int, v);
    i = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(float, i);
}
// row_shl:n -- lane i receives lane i+n of its 16-lane row, 0.0 past the end
constexpr int DPP_ROW_SHL(int n) { return 0x100 + n; }

template <typename T, int L, bool USE_DPP>
__device__ __forceinline__ T group_reduce(T v) {
    if constexpr (USE_DPP) {
        if constexpr (L >= 64) v += __shfl_down(v, 32, 64);
        if constexpr (L >= 32) v += __shfl_down(v, 16, 32);
        if constexpr (L >= 16) v += dpp_mov<DPP_ROW_SHL(8)>(v);
        if constexpr (L >= 8) v += dpp_mov<DPP_ROW_SHL(4)>(v);
        if constexpr (L >= 4) v += dpp_mov<DPP_ROW_SHL(2)>(v);
        if constexpr (L >= 2) v += dpp_mov<DPP_ROW_SHL(1)>(v);
    } else {
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) v += __shfl_down(v, o, L);
    }
    return v;
}

// ---- per-block x window ------------------------------------------------------
// desc[b] = {first column of the window, window length}; length 0 means the
// window does not fit the LDS budget and the block gathers x from global.
// All of a thread's loads are issued before the first LDS write so that the
// staging costs one memory round trip, not one per 16 bytes.
template <typename T, int BLOCK>
__device__ __forceinline__ void stage_window(T *xw, const T *__restrict__ x, uint32_t cbase,
                                             uint32_t wlen) {
    constexpr uint32_t V = 16 / sizeof(T);  // elements per 16-byte load
    const T *src = x + cbase;
    if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        using vec_t = __attribute__((ext_vector_type(4))) uint32_t;
        const uint32_t nvec = wlen / V;
        const vec_t *s4 = reinterpret_cast<const vec_t *>(src);
        vec_t *d4 = reinterpret_cast<vec_t *>(xw);
        constexpr uint32_t K = 4;
        for (uint32_t i0 = threadIdx.x; i0 < nvec; i0 += K * BLOCK) {
            vec_t t[K];
#pragma unroll
            for (uint32_t k = 0; k < K; ++k) {
                const uint32_t i = i0 + k * BLOCK;
                t[k] = s4[i < nvec ? i : i0];
            }
#pragma unroll
            for (uint32_t k = 0; k < K; ++k) {
                const uint32_t i = i0 + k * BLOCK;
                if (i < nvec) d4[i] = t[k];
            }
        }
        for (uint32_t i = nvec * V + threadIdx.x; i < wlen; i += BLOCK) xw[i] = src[i];
    } else {
        for (uint32_t i = threadIdx.x; i < wlen; i += BLOCK) xw[i] = src[i];
    }
}

// The stream kernel's x "window" is a set of PAGES of kPageCols consecutive columns
// (2 KB of f64): the pages the super-tile's rows touch, in ascending order, staged
// back to back in LDS.  A band is a run of consecutive pages; a stencil matrix
// touches a few short runs far apart (the span may be millions of columns while
// only a few thousand distinct ones are read).  col16 = slot * kPageCols + column
// inside the page, so the hot loop does not know the difference.
//   contiguous: page ids are first, first + 1, ...      (no table needed)
//   else      : page ids from pages[first ... first + npages), at most 64
constexpr uint32_t kPageShift = 8;
constexpr uint32_t kPageCols = 1u << kPageShift;
// ring > 0 (contiguous runs only): the window is a ring of `ring` pages, page p at slot p % ring -- the layout the
// sliding kernel (csr_slide.hpp) keeps across steps; col16 is encoded for it (csr_encode_col16).
#ifndef SPAL_STAGE_BATCH
#define SPAL_STAGE_BATCH 12
#endif
template <typename T, int BLOCK>
__device__ __forceinline__ void stage_pages(T *xw, const T *__restrict__ x,
                                            const uint32_t *__restrict__ pages, uint32_t first,
                                            uint32_t npages, bool contiguous, uint32_t ncols, uint32_t ring = 0u) {
    static_assert(BLOCK == 256, "one page of 256 columns per pass of the scalar path");
    constexpr uint32_t V = 16 / sizeof(T);      // elements per 16-byte load
    constexpr uint32_t VP = kPageCols / V;      // 16-byte vectors per page: 128 (f64) / 64 (f32)
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t pid_lane = first + lane;           // contiguous run
    if (!contiguous) pid_lane = pages[first + min(lane, npages - 1u)];   // npages >= 1 here
    if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0 && ncols >= V) {
        using vec_t = __attribute__((ext_vector_type(4))) uint32_t;
        const uint32_t total = npages * VP;     // a multiple of 64: whole waves are in or out
        const uint32_t last_full = ((ncols - V) / V) * V;   // first column of the last whole vector of x
        vec_t *d4 = reinterpret_cast<vec_t *>(xw);
        // (K vectors per thread requested back to back: 12 bring the 24 pages of an f64 window in ONE round trip; with four
        //  a super-tile's 20 pages took three -- round 4, profiles/r04/shard_sized_launches.txt)
        constexpr uint32_t K = SPAL_STAGE_BATCH;
        for (uint32_t i0 = threadIdx.x; i0 < total; i0 += K * BLOCK) {
            vec_t t[K];
            uint32_t e[K];
            // all loads unconditional (clamped, in bounds) so that they leave together
#pragma unroll
            for (uint32_t k = 0; k < K; ++k) {
                const uint32_t ic = min(i0 + k * BLOCK, total - 1u);
                // the 64 vectors of a wave lie inside one page: its id comes from a lane, not from memory
                const uint32_t slot = __builtin_amdgcn_readfirstlane(ic / VP);
                const uint32_t pid = __builtin_amdgcn_readlane(pid_lane, slot);
                e[k] = pid * kPageCols + (ic % VP) * V;     // first column of this vector
                t[k] = *reinterpret_cast<const vec_t *>(x + min(e[k], last_full));
            }
#pragma unroll
            for (uint32_t k = 0; k < K; ++k) {
                const uint32_t i = i0 + k * BLOCK;
                if (i < total) {
                    // where the vector goes: slot i / VP, or the ring slot of its page
                    const uint32_t at = (contiguous && ring) ? ((first + i / VP) % ring) * VP + i % VP : i;
                    d4[at] = t[k];
                    if (e[k] > last_full) {   // at / past the end of x (last page only): columns one by one
#pragma unroll
                        for (uint32_t q = 0; q < V; ++q) xw[at * V + q] = e[k] + q < ncols ? x[e[k] + q] : T(0);
                    }
                }
            }
        }
    } else {   // x not 16-byte aligned: element by element, one page per pass of the workgroup
        for (uint32_t s0 = 0; s0 < npages; ++s0) {
            const uint32_t pid = __builtin_amdgcn_readlane(pid_lane, s0);
            const uint32_t e = pid * kPageCols + threadIdx.x;
            const uint32_t sl = (contiguous && ring) ? pid % ring : s0;
            xw[sl * kPageCols + threadIdx.x] = e < ncols ? x[e] : T(0);
        }
    }
}

// ---- the "vector" kernel: L lanes per row ------------------------------------
// One workgroup owns R consecutive rows.  Each wave walks its rows G = 64/L at
// a time, U such groups ("a step" = G*U rows) per iteration.  Rows longer
// than L loop.
//
// HBM latency under load is microseconds, so a wave must keep several KB of
// loads in flight.  The row loop is software-pipelined by hand:
//     step i+2: row pointers          (L2/L1 hits mostly)
//     step i+1: colind + values       (the HBM stream)
//     step i  : x gather from LDS, multiply, reduce, store
// All loads are unconditional with clamped addresses (an idle lane re-reads a
// neighbour's element: no extra traffic) so they issue back to back and the
// compiler can wait with a counted vmcnt; idle lanes are masked by a select.
template <int U>
struct RowStep {
    uint32_t p[U];  // this lane's first entry of its row
    uint32_t e[U];  // one past the row's last entry (0: no row)
};
template <typename T, int U>
struct EntryStep {
    uint32_t c[U];
    T v[U];
};

template <int L, int U>
__device__ __forceinline__ RowStep<U> load_rows(const uint32_t *__restrict__ rowptr,
                                                uint32_t base, uint32_t row1, uint32_t g,
                                                uint32_t s) {
    constexpr uint32_t G = kWave / L;
    RowStep<U> o;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t r = base + u * G + g;
        const uint32_t rc = min(r, row1 - 1);
        const uint32_t a0 = rowptr[rc], a1 = rowptr[rc + 1];
        o.p[u] = a0 + s;
        o.e[u] = (r < row1) ? a1 : 0u;
    }
    return o;
}

template <typename T, int U, typename CI = uint32_t>
__device__ __forceinline__ EntryStep<T, U> load_entries(const CI *__restrict__ colind,
                                                        const T *__restrict__ vals,
                                                        const RowStep<U> &rs, uint32_t last_nz) {
    EntryStep<T, U> o;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t q = min(rs.p[u], last_nz);
        o.c[u] = load_stream(colind + q);
        o.v[u] = load_stream(vals + q);
    }
    return o;
}

// CI = uint16_t: `colind` holds columns relative to the block's LDS window (column - cbase), 2 bytes per entry
// instead of 4 (INLDS only).
template <typename T, int L, int U, bool INLDS, bool USE_DPP, int BLOCK, int LB = 1, typename CI = uint32_t>
__device__ __forceinline__ void vector_rows(const uint32_t *__restrict__ rowptr,
                                            const CI *__restrict__ colind,
                                            const T *__restrict__ vals, const T *__restrict__ x,
                                            const T *xw, T *__restrict__ y, uint32_t row0,
                                            uint32_t row1, uint32_t cbase, uint32_t last_nz) {
    constexpr uint32_t G = kWave / L;
    constexpr uint32_t NW = BLOCK / kWave;
    constexpr uint32_t STRIDE = NW * G * U;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t g = lane / L, s = lane % L;
    constexpr bool REL = sizeof(CI) == 2;            // columns already relative to the window
    static_assert(!REL || INLDS, "16-bit columns address the LDS window");
    const uint32_t coff = REL ? 0u : cbase;          // what to subtract from a column / an idle lane's column

    uint32_t base = row0 + wave * (G * U);
    if (base >= row1) return;  // wave-uniform
    RowStep<U> rs = load_rows<L, U>(rowptr, base, row1, g, s);
    EntryStep<T, U> es = load_entries<T, U>(colind, vals, rs, last_nz);
    RowStep<U> rs_next = load_rows<L, U>(rowptr, base + STRIDE, row1, g, s);

    for (;;) {
        // step i+1: the stream; step i+2: its row pointers
        EntryStep<T, U> es_next = load_entries<T, U>(colind, vals, rs_next, last_nz);
        RowStep<U> rs_next2 = load_rows<L, U>(rowptr, base + 2 * STRIDE, row1, g, s);

        // step i: all U gathers go out before the first product needs one
        T acc[U], xv[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            live[u] = rs.p[u] < rs.e[u];
            // an idle lane's clamped column may lie outside this block's window
            const uint32_t cc = live[u] ? es.c[u] : coff;
            xv[u] = INLDS ? xw[cc - coff] : x[cc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            keep_unconditional(xv[u]);
            // select, not multiply by zero: an idle lane must not inject x's NaN/Inf
            acc[u] = live[u] ? es.v[u] * xv[u] : T(0);
        }
        // rows with more than L entries
        if constexpr (LB == 1) {
            // one more (colind, value) pair per lane and iteration: the lean form, used
            // wherever rows seldom exceed L (keeps the hot loop's registers out of scratch)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                for (uint32_t q = rs.p[u] + L; q < rs.e[u]; q += L) {
                    const uint32_t cc = load_stream(colind + q);
                    const T vv = load_stream(vals + q);
                    const T xv = INLDS ? xw[cc - coff] : x[cc];
                    acc[u] = __builtin_fma(vv, xv, acc[u]);
                }
            }
        } else {
            // long rows (planner: mean above 64 entries): LB more pairs per lane go out
            // together (clamped addresses, selects on the products), so a row costs a
            // memory round trip per LB*L entries, not per L
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t q = rs.p[u] + L;
                const uint32_t e = rs.e[u];
                while (__any(q < e)) {   // wave-uniform trip count (lanes of other rows idle along)
                    uint32_t cc[LB];
                    T vv[LB], xv[LB];
#pragma unroll
                    for (int k = 0; k < LB; ++k) {
                        const uint32_t qq = min(q + k * L, last_nz);
                        cc[k] = load_stream(colind + qq);
                        vv[k] = load_stream(vals + qq);
                    }
#pragma unroll
                    for (int k = 0; k < LB; ++k) {
                        const bool live4 = q + k * L < e;
                        const uint32_t c4 = live4 ? cc[k] : coff;
                        xv[k] = INLDS ? xw[c4 - coff] : x[c4];
                    }
#pragma unroll
                    for (int k = 0; k < LB; ++k) {
                        keep_unconditional(xv[k]);
                        if (q + k * L < e) acc[u] = __builtin_fma(vv[k], xv[k], acc[u]);
                    }
                    q += LB * L;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = group_reduce<T, L, USE_DPP>(acc[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = base + u * G + g;
            if (s == 0 && r < row1) y[r] = acc[u];
        }

        base += STRIDE;
        if (base >= row1) break;  // wave-uniform
        rs = rs_next;
        es = es_next;
        rs_next = rs_next2;
    }
}

template <typename T, int L, int U, bool LDSX, bool USE_DPP, int BLOCK, int LB = 1>
__global__ __launch_bounds__(BLOCK, 8) void csr_spmv_vector(
    const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
    const T *__restrict__ vals, const T *__restrict__ x, T *__restrict__ y,
    const uint4 *__restrict__ desc, uint32_t nrows, uint32_t nnz, uint32_t R, uint32_t nblocks,
    uint32_t per_xcd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char spal_smem[];
    T *xw = reinterpret_cast<T *>(spal_smem);

    const uint32_t b = xcd_contiguous_block(blockIdx.x, per_xcd);
    if (b >= nblocks) return;
    const uint32_t row0 = b * R;
    const uint32_t row1 = min(row0 + R, nrows);
    const uint32_t last_nz = nnz - 1;  // nnz >= 1 (the host never launches an empty matrix)

    if constexpr (LDSX) {
        const uint4 d = desc[b];  // block-uniform
        if (d.z == kModeVectorLds) {
            stage_window<T, BLOCK>(xw, x, d.x, d.y);
            __syncthreads();
            vector_rows<T, L, U, true, USE_DPP, BLOCK, LB>(rowptr, colind, vals, x, xw, y, row0, row1,
                                                           d.x, last_nz);
            return;
        }
    }
    vector_rows<T, L, U, false, USE_DPP, BLOCK, LB>(rowptr, colind, vals, x, nullptr, y, row0, row1, 0u,
                                                    last_nz);
}

// The long-row form of the vector kernel with 16-bit columns: blocks whose x window is in LDS read `col16`
// (column - window base, written by csr_encode_col16_window at plan time) -- 10 instead of 12 bytes per entry,
// and these rows run at the HBM rate (400 entries per row: 5.2 TB/s actual with 32-bit columns); the other blocks
// read the 32-bit columns and gather x from global memory.  A wave per row (L = 64, or 32), one row group in flight.
template <typename T, int L, int BLOCK, int LB>
__global__ __launch_bounds__(BLOCK, 8) void csr_spmv_vector_col16(
    const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
    const uint16_t *__restrict__ col16, const T *__restrict__ vals, const T *__restrict__ x,
    T *__restrict__ y, const uint4 *__restrict__ desc, uint32_t nrows, uint32_t nnz, uint32_t R,
    uint32_t nblocks, uint32_t per_xcd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char spal_smem[];
    T *xw = reinterpret_cast<T *>(spal_smem);
    const uint32_t b = xcd_contiguous_block(blockIdx.x, per_xcd);
    if (b >= nblocks) return;
    const uint32_t row0 = b * R;
    const uint32_t row1 = min(row0 + R, nrows);
    const uint32_t last_nz = nnz - 1;
    const uint4 d = desc[b];  // block-uniform
    if (d.z == kModeVectorLds) {
        stage_window<T, BLOCK>(xw, x, d.x, d.y);
        __syncthreads();
        vector_rows<T, L, 1, true, true, BLOCK, LB, uint16_t>(rowptr, col16, vals, x, xw, y, row0, row1, d.x,
                                                              last_nz);
        return;
    }
    vector_rows<T, L, 1, false, true, BLOCK, LB>(rowptr, colind, vals, x, nullptr, y, row0, row1, 0u, last_nz);
}

// ---- the "stream" kernel: one lane per row, products parked in LDS ----------
// (CSR-Adaptive's CSR-Stream, re-tiled for wave64 / 160 KB LDS.)
//
// A workgroup (4 waves) owns a SUPER-TILE of 256 * TPW rows (TPW = 4: 1024) and
// stages its x window in LDS once.  Each wave owns TPW TILES of 64 consecutive
// rows.  For a tile the wave
//   1. loads the tile's values and 16-bit columns (page slot * 256 + column inside the page) with wide,
//      perfectly coalesced loads: lane l, step j holds entries
//      start + (64 j + l) * 2 + {0, 1}   (tile k+1 is in flight while tile k
//      is being processed),
//   2. gathers x from the LDS window, multiplies, and writes the products to
//      its private LDS strip in entry order,
//   3. lane l then sums row (first + l) left to right out of LDS -- exactly the
//      reference's order of additions (src/csr/ops/mul.rs:31-38; first product
//      assigned, later ones added, mul and add rounded separately), so rows
//      handled here are BIT-IDENTICAL to the sequential CPU result,
//   4. stores 64 consecutive y (one coalesced 512-byte store).
// About 0.17 wave-instructions per stored entry against 1.1 for the vector
// kernel, and 10 instead of 12 bytes per entry from HBM.
//
// A super-tile streams out of LDS when the pages of x its rows touch fit the window budget (otherwise the
// same pipeline gathers x from global memory: stream_global_super_tile); a TILE that holds more than
// kStreamTileNnz entries or a very long row is skipped here and left to csr_spmv_overflow (below).
constexpr int kStreamBlock = 256;
constexpr int kStreamWaves = kStreamBlock / kWave;                          // 4
// A tile holds RPT rows (64, 32, 24, 16, 12 or 8: one lane per row, the other lanes idle
// in the reduction; 128 or 256: two or four adjacent rows per lane, for rows of at most 8 / 4 entries) and at most kStreamTileNnz entries; narrower tiles let
// matrices with up to ~120 entries per row stream as well.
// rows of a super-tile when every wave owns TPW tiles of RPT rows (4 x 64: 1024)
constexpr int stream_rows(int tpw, int rpt = 64) { return kStreamWaves * tpw * rpt; }
constexpr int kStreamSteps = 8;                                             // 128 entries per step
constexpr int kStreamTileNnz = kStreamSteps * 128;                          // 1024 incl. alignment slack
// SKEW: the product strip in LDS is skewed by one entry per 128 bytes.  A lane sums its row out of the strip,
// and rows whose length is a multiple of 16 (f64; 32 f32) start in the same bank for every lane: 16 / 32 / 64 per
// row ran at 61 / 60 / 59 % of the HBM peak against 73 / 80 / 84 % for 15 / 31 / 63.  Skewed: 77 / 73 / 80 %, but
// 67 / 69 / 76 % for 15 / 31 / 63 (two LDS instructions where one did, address arithmetic) -- so it is a template
// parameter and the plan's choice (csr_plan_build: most rows a multiple of 128 bytes long).
template <bool SKEW>
constexpr int stream_strip() { return SKEW ? kStreamTileNnz + kStreamTileNnz / 16 : kStreamTileNnz; }   // entries of LDS per wave
template <typename T, bool SKEW>
__device__ __forceinline__ uint32_t strip_pos(uint32_t e) { return SKEW ? e + (e >> (sizeof(T) == 8 ? 4 : 5)) : e; }
constexpr int kStreamPad = 256;  // device arrays are over-allocated by this many entries


// Ablation builds only (-DSPAL_DIAG, tools/build_variant.sh; never in the shipped library): bits of the kernels'
// `flags` argument switch parts of the stream kernel off so that their share of the time can be measured
// (results are then WRONG): bit 8 lanes read the product strip at conflict-free addresses, bit 9 the x
// window is not staged, bit 10 no row sums (a lane stores one product).
#ifdef SPAL_DIAG
#define SPAL_DIAG_ON(flags, bit) ((((flags) >> (bit)) & 1u) != 0u)
#else
#define SPAL_DIAG_ON(flags, bit) false
#endif

template <typename T> struct Pair;
template <> struct Pair<double> { using type = __attribute__((ext_vector_type(2))) double; };
template <> struct Pair<float> { using type = __attribute__((ext_vector_type(2))) float; };

// A tile whose rows hold more than the strip's 1024 entries (a heavy row among light ones) is not
// streamed: the stream kernels skip it and csr_spmv_overflow, launched right after them over the
// plan's list of such tiles, computes its rows -- a wave per row.  The other tiles of the
// super-tile stream as usual.  (Handing the whole super-tile to the vector path made it the
// launch's straggler: one row of 3000 entries every 20 super-tiles cost +50 ... +130 %; running
// the vector rows inside the stream kernels cost their hot loop 6 ... 10 % in registers.)
// entries of the tile [b, e) as the strip sees them (from the even start)
__device__ __forceinline__ bool stream_tile_overflows(uint32_t b, uint32_t e) {
    return e - (b & ~1u) > (uint32_t)kStreamTileNnz;
}
// Which tiles those are is the plan's decision (more than 1024 entries, or a row so long that its lane
// would keep the other 63 waiting): a bit per tile of the super-tile, packed into the descriptor --
// desc.w = contiguous | bits 0..15 << 16, desc.z = mode | bits 16..31 << 16.
__device__ __forceinline__ uint32_t desc_mode(const uint4 &d) { return d.z & 0xffffu; }
// stream modes: desc.y = pages of the window (at most 64) | ulen << 8, ulen = 1 + the length every row of the
// super-tile has, or 0 when they differ (stream_row_bounds)
__device__ __forceinline__ uint32_t desc_ulen(const uint4 &d) { return d.y >> 8; }
__device__ __forceinline__ uint32_t desc_skip_bits(const uint4 &d) { return (d.w >> 16) | (d.z & 0xffff0000u); }
// the bits of one wave's TPW tiles; zero for almost every wave, which then runs the tile loop compiled
// without the test
template <int TPW>
__device__ __forceinline__ uint32_t wave_skip_mask(uint32_t bits, uint32_t wave) {
    return (bits >> (wave * TPW)) & ((1u << TPW) - 1u);
}

// A lane's left-to-right sum of the products of one row out of the wave's strip: the reads of four products go
// out together, the adds stay in stored order (the order is the contract, the batching is not).
template <typename T, bool SKEW>
__device__ __forceinline__ T strip_row_sum(const T *prod, uint32_t off, uint32_t len) {
    T acc = T(0);
    if (len) {
        acc = prod[strip_pos<T, SKEW>(off)];
        uint32_t k = 1;
        for (; k + 4 <= len; k += 4) {
            const T p0 = prod[strip_pos<T, SKEW>(off + k)], p1 = prod[strip_pos<T, SKEW>(off + k + 1)],
                    p2 = prod[strip_pos<T, SKEW>(off + k + 2)], p3 = prod[strip_pos<T, SKEW>(off + k + 3)];
            acc = acc + p0;
            acc = acc + p1;
            acc = acc + p2;
            acc = acc + p3;
        }
        for (; k < len; ++k) acc = acc + prod[strip_pos<T, SKEW>(off + k)];
    }
    return acc;
}
// ... and the stores of a tile's results: RPT <= 64 a row per lane, RPT = 128 / 256 the lane's two / four adjacent rows
template <typename T, int RPT, bool SKEW, typename Tile>
__device__ __forceinline__ void strip_sums_to_y(const Tile &t, const T *prod, T *__restrict__ y, uint32_t row0,
                                                uint32_t row1, uint32_t lane, uint32_t nt_store) {
    // nt_store: y stored non-temporally (the autotune's choice).  A third form -- written through at agent scope, an
    // atomic store -- gained 2-6 us in a micro-benchmark, but its mere PRESENCE as a third branch in these kernels cost
    // the sliding kernel 12 % (252.7 -> 283-290 us on one handle, never executed: profiles/r03/README.md): not built in.
    const uint32_t rlast = min(row0 + (uint32_t)RPT, row1);
    if constexpr (RPT > 64) {
        constexpr uint32_t RPL = RPT / 64;
        T acc[RPL];
#pragma unroll
        for (uint32_t i = 0; i < RPL; ++i)
            acc[i] = strip_row_sum<T, SKEW>(prod, t.rpl[i] - t.start, t.rpl[i + 1] - t.rpl[i]);
        __builtin_amdgcn_wave_barrier();  // the next tile's products overwrite this strip
#pragma unroll
        for (uint32_t i = 0; i < RPL; ++i) {
            const uint32_t r = row0 + RPL * lane + i;
            if (r < rlast) {
                if (nt_store) __builtin_nontemporal_store(acc[i], &y[r]);
                else y[r] = acc[i];
            }
        }
    } else {
        const T acc = strip_row_sum<T, SKEW>(prod, t.rp0 - t.start, t.rp1 - t.rp0);
        __builtin_amdgcn_wave_barrier();  // the next tile's products overwrite this strip
        if (row0 + lane < rlast) {
            if (nt_store) __builtin_nontemporal_store(acc, &y[row0 + lane]);  // y is written once, never re-read here
            else y[row0 + lane] = acc;
        }
    }
}

template <typename T>
struct StreamTile {
    typename Pair<T>::type v[kStreamSteps];
    uint32_t c[kStreamSteps];  // two 16-bit LDS-window positions (page slot * 256 + column inside the page)
    uint32_t rp0, rp1;         // rowptr[row], rowptr[row + 1] of this lane's row
    uint32_t rpl[5];           // RPT = 128 / 256 (two / four adjacent rows per lane): rowptr[first row + 0 ... 4]
    uint32_t start;            // first loaded entry (tile start rounded down to even); wave-uniform
    uint32_t steps;            // 128-entry steps that hold entries of the tile; wave-uniform
};

// The row bounds a lane needs for its sums.  ulen == 0: from rowptr (two loads per lane and tile; both
// unconditional: lanes past the tile's last row read rowptr[rlast] twice, i.e. an empty row -- no select,
// hence no wait here).  ulen > 0: the plan found every row of the super-tile to hold exactly ulen - 1 entries
// (structured / banded matrices): the bounds follow from the tile's first entry b, and rowptr -- 4 bytes per
// row, 2 % of config 3's traffic -- is not read at all.
template <typename Tile, int RPT>
__device__ __forceinline__ void stream_row_bounds(Tile &t, const uint32_t *__restrict__ rowptr, uint32_t row0,
                                                  uint32_t rlast, uint32_t b, uint32_t lane, uint32_t ulen) {
    if (ulen) {   // wave-uniform
        const uint32_t len = ulen - 1u, nrt = rlast - row0;
        if constexpr (RPT > 64) {
            constexpr uint32_t RPL = RPT / 64;
#pragma unroll
            for (uint32_t i = 0; i <= RPL; ++i) t.rpl[i] = b + min(RPL * lane + i, nrt) * len;
        } else {
            t.rp0 = b + min(lane, nrt) * len;
            t.rp1 = b + min(lane + 1u, nrt) * len;
        }
        return;
    }
    if constexpr (RPT > 64) {   // rows of very few entries: lane l owns RPT / 64 adjacent rows
        constexpr uint32_t RPL = RPT / 64;
#pragma unroll
        for (uint32_t i = 0; i <= RPL; ++i) t.rpl[i] = rowptr[min(row0 + RPL * lane + i, rlast)];
    } else {
        t.rp0 = rowptr[min(row0 + lane, rlast)];
        t.rp1 = rowptr[min(row0 + lane + 1, rlast)];
    }
}

// b, e: the tile's entry range [rowptr[row0], rowptr[last row + 1]), wave-uniform
template <typename T, int RPT>
__device__ __forceinline__ void stream_load(StreamTile<T> &t, const uint32_t *__restrict__ rowptr,
                                            const uint16_t *__restrict__ col16,
                                            const T *__restrict__ vals, uint32_t row0,
                                            uint32_t row1, uint32_t b, uint32_t e, uint32_t lane,
                                            uint32_t ulen = 0u) {
    using pair_t = typename Pair<T>::type;
    const uint32_t rlast = min(row0 + (uint32_t)RPT, row1);
    t.start = b & ~1u;
    t.steps = (e - t.start + 127u) >> 7;
    const uint32_t e0 = t.start + lane * 2;
    // the HBM stream first ...
#pragma unroll
    for (int j = 0; j < kStreamSteps; ++j) {
        if ((uint32_t)j < t.steps) {  // uniform
            t.v[j] = __builtin_nontemporal_load(reinterpret_cast<const pair_t *>(vals + e0 + j * 128));
            t.c[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(col16 + e0 + j * 128));
        }
    }
    // ... then this lane's row bounds (needed only by the reduction)
    stream_row_bounds<StreamTile<T>, RPT>(t, rowptr, row0, rlast, b, lane, ulen);
}

template <typename T, int RPT, bool SKEW>
__device__ __forceinline__ void stream_compute(const StreamTile<T> &t, const T *xw, uint32_t wmax,
                                               T *prod, T *__restrict__ y, uint32_t row0,
                                               uint32_t row1, uint32_t lane, uint32_t nt_store = 0u,
                                               uint32_t flags = 0u) {
    using pair_t = typename Pair<T>::type;
#pragma unroll
    for (int j = 0; j < kStreamSteps; ++j) {
        if ((uint32_t)j < t.steps) {  // uniform
            // entries past the tile's end belong to other tiles (or the padding):
            // clamp their column into the window, nobody reads their product
            const uint32_t c0 = min(t.c[j] & 0xffffu, wmax);
            const uint32_t c1 = min(t.c[j] >> 16, wmax);
            pair_t p;
            p.x = t.v[j].x * xw[c0];
            p.y = t.v[j].y * xw[c1];
            if constexpr (SKEW) {
                const uint32_t at = strip_pos<T, true>(2u * (j * 64 + lane));   // an even entry and its odd neighbour stay adjacent
                prod[at] = p.x;
                prod[at + 1] = p.y;
            } else {
                reinterpret_cast<pair_t *>(prod)[j * 64 + lane] = p;
            }
        }
    }
    // the wave's own LDS writes are read back by other lanes of the same wave:
    // LDS executes a wave's instructions in order, only the compiler must not
    // move the reads above the writes
    __builtin_amdgcn_wave_barrier();
#ifdef SPAL_DIAG
    if (SPAL_DIAG_ON(flags, 8) || SPAL_DIAG_ON(flags, 10)) {   // ablation: see SPAL_DIAG_ON
        T acc = prod[lane];
        if (SPAL_DIAG_ON(flags, 8)) {
            const uint32_t len = t.rp1 - t.rp0;
            for (uint32_t k = 1; k < len; ++k) acc = acc + prod[(k * 64u + lane) & (uint32_t)(kStreamTileNnz - 1)];
        }
        __builtin_amdgcn_wave_barrier();
        if (row0 + lane < min(row0 + (uint32_t)RPT, row1)) y[row0 + lane] = acc;
        return;
    }
#endif
    (void)flags;
    strip_sums_to_y<T, RPT, SKEW>(t, prod, y, row0, row1, lane, nt_store);
}

// ---- stream tiles whose x window does not fit LDS ------------------------------
// Same tiles, same lane-per-row sums (still bit-identical to the reference
// order); the columns come from the 32-bit array and x is gathered from global
// memory (L2 / Infinity Cache), all gathers of a tile in flight together.
template <typename T>
struct StreamTileG {
    typename Pair<T>::type v[kStreamSteps];
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    u32x2 c[kStreamSteps];     // two 32-bit columns
    uint32_t rp0, rp1, rpl[5], start, steps;
};

template <typename T, int RPT>
__device__ __forceinline__ void stream_load_g(StreamTileG<T> &t, const uint32_t *__restrict__ rowptr,
                                              const uint32_t *__restrict__ colind,
                                              const T *__restrict__ vals, uint32_t row0,
                                              uint32_t row1, uint32_t b, uint32_t e, uint32_t lane,
                                              uint32_t ulen = 0u) {
    using pair_t = typename Pair<T>::type;
    const uint32_t rlast = min(row0 + (uint32_t)RPT, row1);
    t.start = b & ~1u;
    t.steps = (e - t.start + 127u) >> 7;
    const uint32_t e0 = t.start + lane * 2;
#pragma unroll
    for (int j = 0; j < kStreamSteps; ++j) {
        if ((uint32_t)j < t.steps) {  // uniform
            t.v[j] = __builtin_nontemporal_load(reinterpret_cast<const pair_t *>(vals + e0 + j * 128));
            t.c[j] = __builtin_nontemporal_load(
                reinterpret_cast<const typename StreamTileG<T>::u32x2 *>(colind + e0 + j * 128));
        }
    }
    stream_row_bounds<StreamTileG<T>, RPT>(t, rowptr, row0, rlast, b, lane, ulen);
}

template <typename T, int RPT, bool SKEW>
__device__ __forceinline__ void stream_compute_g(const StreamTileG<T> &t, const T *__restrict__ x,
                                                 uint32_t cmax, T *prod, T *__restrict__ y,
                                                 uint32_t row0, uint32_t row1, uint32_t lane,
                                                 uint32_t nt_store) {
    using pair_t = typename Pair<T>::type;
    T xa[kStreamSteps], xb[kStreamSteps];
#pragma unroll
    for (int j = 0; j < kStreamSteps; ++j) {
        if ((uint32_t)j < t.steps) {  // uniform; entries past the tile's end: clamp into x
            xa[j] = x[min(t.c[j].x, cmax)];
            xb[j] = x[min(t.c[j].y, cmax)];
        }
    }
#pragma unroll
    for (int j = 0; j < kStreamSteps; ++j) {
        if ((uint32_t)j < t.steps) {
            pair_t p;
            p.x = t.v[j].x * xa[j];
            p.y = t.v[j].y * xb[j];
            if constexpr (SKEW) {
                const uint32_t at = strip_pos<T, true>(2u * (j * 64 + lane));   // an even entry and its odd neighbour stay adjacent
                prod[at] = p.x;
                prod[at + 1] = p.y;
            } else {
                reinterpret_cast<pair_t *>(prod)[j * 64 + lane] = p;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    strip_sums_to_y<T, RPT, SKEW>(t, prod, y, row0, row1, lane, nt_store);
}

// one super-tile in stream-global mode (no window, no workgroup barrier)
template <typename T, int TPW, int RPT, bool SKEW>
__device__ __forceinline__ void stream_global_super_tile(const uint32_t *__restrict__ rowptr,
                                                         const uint32_t *__restrict__ colind,
                                                         const T *__restrict__ vals,
                                                         const T *__restrict__ x, T *__restrict__ y,
                                                         T *prod, uint32_t row0, uint32_t row1,
                                                         uint32_t ncols, bool nt_store, uint32_t skip_bits,
                                                         uint32_t ulen) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t wrow = row0 + wave * (TPW * (uint32_t)RPT);
    if (wrow >= row1) return;  // wave-uniform
    const uint32_t tb_lane = rowptr[min(wrow + min(lane, (uint32_t)TPW) * (uint32_t)RPT, row1)];
    uint32_t tb[TPW + 1];
#pragma unroll
    for (int k = 0; k <= TPW; ++k) tb[k] = __builtin_amdgcn_readlane(tb_lane, k);
    StreamTileG<T> cur, nxt;
    stream_load_g<T, RPT>(cur, rowptr, colind, vals, wrow, row1, tb[0], tb[1], lane, ulen);
    const uint32_t ovmask = wave_skip_mask<TPW>(skip_bits, wave);
    auto tiles = [&](auto ov) {
        constexpr bool OV = decltype(ov)::value;
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
            const uint32_t r0 = wrow + k * (uint32_t)RPT;
            if (r0 >= row1) break;  // wave-uniform
            const uint32_t rn = r0 + (uint32_t)RPT;
            const bool more = (k + 1 < TPW) && rn < row1;
            if (more) stream_load_g<T, RPT>(nxt, rowptr, colind, vals, rn, row1, tb[k + 1],
                                            tb[k + 2 <= TPW ? k + 2 : k + 1], lane, ulen);
            if (!OV || !((ovmask >> k) & 1u))  // oversized tiles: csr_spmv_overflow
                stream_compute_g<T, RPT, SKEW>(cur, x, ncols - 1, prod, y, r0, row1, lane, nt_store);
            if (more) cur = nxt;
        }
    };
    if (ovmask == 0u) tiles(std::false_type{});
    else tiles(std::true_type{});
}

// desc[b] = Stream: {first page id or offset into pages[], number of pages, mode, 1 if the pages are a
// contiguous run}; VectorLds: {window base column, window length, mode, 0}
// PF: tiles of loads a wave keeps in flight ahead of the one it is working on (1 or 2).  Two workgroups per CU are
// 8 waves; with one tile (9 KB) ahead each, a CU has ~70 KB of loads in flight, about what it takes to cover the HBM
// latency at full rate and nothing to spare while a wave sits in its row sums; two tiles ahead cost 40 more registers.
template <typename T, int L, int U, bool USE_DPP, int TPW, int RPT, bool SKEW = false, int PF = 1>
__global__ __launch_bounds__(kStreamBlock, 2) void csr_spmv_stream(
    const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
    const uint16_t *__restrict__ col16, const T *__restrict__ vals, const T *__restrict__ x,
    T *__restrict__ y, const uint4 *__restrict__ desc, const uint32_t *__restrict__ pages, uint32_t nrows,
    uint32_t ncols, uint32_t nnz, uint32_t nblocks, uint32_t per_xcd, uint32_t flags, uint32_t ring) {
    extern __shared__ __attribute__((aligned(16))) unsigned char spal_smem[];
    const bool nt_store = flags & 1u;
    // [ products: 4 waves x kStreamTileNnz ][ x window ]
    T *prod_all = reinterpret_cast<T *>(spal_smem);
    T *xw = prod_all + kStreamWaves * stream_strip<SKEW>();

    const uint32_t b = xcd_block(blockIdx.x, per_xcd);
    if (b >= nblocks) return;
    constexpr uint32_t kRows = stream_rows(TPW, RPT);
    const uint32_t row0 = b * kRows;
    const uint32_t row1 = min(row0 + kRows, nrows);
    uint4 d = desc[b];  // block-uniform
    const uint32_t skip_bits = desc_skip_bits(d);
    d.z = desc_mode(d);
    if ((d.w & 2u) && (flags & 2u)) return;   // a wide band's super-tile: csr_spmv_panel (csr_panel.hpp) takes it
    d.w &= 1u;

    if (d.z == kModeStream) {
        const uint32_t ulen = desc_ulen(d);
        d.y &= 0xffu;
        const uint32_t lane = threadIdx.x & (kWave - 1);
        const uint32_t wave = threadIdx.x / kWave;
        T *prod = prod_all + wave * stream_strip<SKEW>();
        // this wave's tiles: rows row0 + (wave*4 + k) * 64
        const uint32_t wrow = row0 + wave * (TPW * (uint32_t)RPT);
        // entry offsets of this wave's tile boundaries, fetched once (lane k holds
        // boundary k) so that no tile's loads wait on a row-pointer round trip
        const uint32_t tb_lane = rowptr[min(wrow + min(lane, (uint32_t)TPW) * (uint32_t)RPT, row1)];
        uint32_t tb[TPW + 1];
#pragma unroll
        for (int k = 0; k <= TPW; ++k) tb[k] = __builtin_amdgcn_readlane(tb_lane, k);
        constexpr int NB = PF + 1;      // tiles a wave holds: the one it works on + PF in flight
        StreamTile<T> t[NB];
        const bool has0 = wrow < row1;  // wave-uniform
        // the first PF tiles' loads overlap the staging
#pragma unroll
        for (int q = 0; q < PF && q < TPW; ++q)
            if (wrow + q * (uint32_t)RPT < row1)
                stream_load<T, RPT>(t[q % NB], rowptr, col16, vals, wrow + q * (uint32_t)RPT, row1, tb[q], tb[q + 1], lane, ulen);
        if (!SPAL_DIAG_ON(flags, 9))
            stage_pages<T, kStreamBlock>(xw, x, pages, d.x, d.y, d.w != 0u, ncols, ring);   // d = {first, npages, mode, contiguous}
        __syncthreads();
        if (!has0) return;
        const uint32_t wmax = ((d.w && ring) ? ring : d.y) * kPageCols - 1u;
        const uint32_t ovmask = wave_skip_mask<TPW>(skip_bits, wave);
        auto tiles = [&](auto ov) {
            constexpr bool OV = decltype(ov)::value;
#pragma unroll
            for (int k = 0; k < TPW; ++k) {
                const uint32_t r0 = wrow + k * (uint32_t)RPT;
                if (r0 >= row1) break;  // wave-uniform
                const int kn = k + PF < TPW ? k + PF : TPW - 1;   // the tile PF ahead, if the wave has one
                const uint32_t rn = wrow + kn * (uint32_t)RPT;
                if (k + PF < TPW && rn < row1)
                    stream_load<T, RPT>(t[kn % NB], rowptr, col16, vals, rn, row1, tb[kn], tb[kn + 1], lane, ulen);
                if (!OV || !((ovmask >> k) & 1u))  // oversized tiles: csr_spmv_overflow
                    stream_compute<T, RPT, SKEW>(t[k % NB], xw, wmax, prod, y, r0, row1, lane, nt_store, flags);
            }
        };
        if (ovmask == 0u) tiles(std::false_type{});
        else tiles(std::true_type{});
        return;
    }
    if (d.z == kModeStreamGlobal) {
        stream_global_super_tile<T, TPW, RPT, SKEW>(rowptr, colind, vals, x, y,
                                              prod_all + (threadIdx.x / kWave) * stream_strip<SKEW>(), row0,
                                              row1, ncols, nt_store, skip_bits, desc_ulen(d));
        return;
    }
    const uint32_t last_nz = nnz - 1;
    if (d.z == kModeVectorLds) {
        stage_window<T, kStreamBlock>(xw, x, d.x, d.y);
        __syncthreads();
        vector_rows<T, L, U, true, USE_DPP, kStreamBlock, 4>(rowptr, colind, vals, x, xw, y, row0, row1,
                                                          d.x, last_nz);
        return;
    }
    vector_rows<T, L, U, false, USE_DPP, kStreamBlock, 4>(rowptr, colind, vals, x, nullptr, y, row0, row1,
                                                       0u, last_nz);
}

// ---- persistent form of the stream kernel --------------------------------------
// Same tiles, same arithmetic, same results as csr_spmv_stream; what changes is
// who walks them.  A fixed grid (two workgroups per CU) is launched; workgroups
// that share an XCD split that XCD's contiguous run of super-tiles into
// contiguous chunks and loop over them.  The wave's load pipeline never drains
// at a super-tile boundary: while the last tile of super-tile s is summed, the
// first tile of s+1 is already in flight, and it stays in flight across the two
// barriers that retire the old x window and publish the new one.  This removes
// the per-workgroup cold start (boundary load -> tile loads -> window) and the
// drain that the one-super-tile-per-workgroup form pays every 1024 rows.
template <typename T, int L, int U, bool USE_DPP, int TPW, int RPT, bool SKEW = false>
__global__ __launch_bounds__(kStreamBlock, 2) void csr_spmv_stream_persistent(
    const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
    const uint16_t *__restrict__ col16, const T *__restrict__ vals, const T *__restrict__ x,
    T *__restrict__ y, const uint4 *__restrict__ desc, const uint32_t *__restrict__ pages, uint32_t nrows,
    uint32_t ncols, uint32_t nnz, uint32_t nblocks, uint32_t per_xcd, uint32_t chunk, uint32_t flags, uint32_t ring) {
    extern __shared__ __attribute__((aligned(16))) unsigned char spal_smem[];
    const bool nt_store = flags & 1u;
    T *prod_all = reinterpret_cast<T *>(spal_smem);
    T *xw = prod_all + kStreamWaves * stream_strip<SKEW>();
    constexpr uint32_t kRows = stream_rows(TPW, RPT);

    // this workgroup's super-tiles: [s_begin, s_end) inside its XCD's run
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t run_end = min((xcd + 1) * per_xcd, nblocks);
    const uint32_t s_begin = xcd * per_xcd + slot * chunk;
    if (s_begin >= run_end) return;
    const uint32_t s_end = min(s_begin + chunk, run_end);

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = threadIdx.x / kWave;
    T *prod = prod_all + wave * stream_strip<SKEW>();
    const uint32_t last_nz = nnz - 1;

    // tile boundaries (entry offsets) of this wave's TPW tiles in super-tile s
    auto bounds_lane = [&](uint32_t s) {
        const uint32_t row0 = s * kRows, row1 = min(row0 + kRows, nrows);
        const uint32_t wrow = row0 + wave * (TPW * (uint32_t)RPT);
        return rowptr[min(wrow + min(lane, (uint32_t)TPW) * (uint32_t)RPT, row1)];
    };

    StreamTile<T> cur, nxt;
    bool cur_valid = false;          // cur holds the first tile of super-tile s (prefetched)
    uint32_t tbl = bounds_lane(s_begin);

    for (uint32_t s = s_begin; s < s_end; ++s) {
        const uint32_t row0 = s * kRows, row1 = min(row0 + kRows, nrows);
        uint4 d = desc[s];  // block-uniform
        const uint32_t skip_bits = desc_skip_bits(d);
        d.z = desc_mode(d);
        const uint32_t dw_raw = d.w;
        d.w &= 1u;
        // the next super-tile's boundaries: asked for now, needed at this one's last tile
        const uint4 d_next = desc[min(s + 1, s_end - 1)];
        const bool next_stream = (s + 1 < s_end) && desc_mode(d_next) == kModeStream;
        const uint32_t ulen_next = desc_ulen(d_next);
        const uint32_t tbl_next = (s + 1 < s_end) ? bounds_lane(s + 1) : 0u;

        const uint32_t ulen = desc_ulen(d);   // (stream modes)
        if (d.z == kModeStream) {
            d.y &= 0xffu;
            uint32_t tb[TPW + 1];
#pragma unroll
            for (int k = 0; k <= TPW; ++k) tb[k] = __builtin_amdgcn_readlane(tbl, k);
            const uint32_t wrow = row0 + wave * (TPW * (uint32_t)RPT);
            const bool has0 = wrow < row1;  // wave-uniform
            if (has0 && !cur_valid) stream_load<T, RPT>(cur, rowptr, col16, vals, wrow, row1, tb[0], tb[1], lane, ulen);
            __syncthreads();  // every wave is done with the previous window
            stage_pages<T, kStreamBlock>(xw, x, pages, d.x, d.y, d.w != 0u, ncols, ring);
            __syncthreads();
            const uint32_t wmax = ((d.w && ring) ? ring : d.y) * kPageCols - 1u;
            bool fetched_next = false;
            if (has0) {
                const uint32_t ovmask = wave_skip_mask<TPW>(skip_bits, wave);
                auto tiles = [&](auto ov) {
                    constexpr bool OV = decltype(ov)::value;
#pragma unroll
                    for (int k = 0; k < TPW; ++k) {
                        const uint32_t r0 = wrow + k * (uint32_t)RPT;
                        if (r0 >= row1) break;  // wave-uniform
                        const uint32_t rn = r0 + (uint32_t)RPT;
                        const bool more = (k + 1 < TPW) && rn < row1;
                        if (more) {
                            stream_load<T, RPT>(nxt, rowptr, col16, vals, rn, row1, tb[k + 1],
                                           tb[k + 2 <= TPW ? k + 2 : k + 1], lane, ulen);
                        } else if (next_stream) {
                            // last tile of this super-tile: start on the next one's first tile
                            const uint32_t nrow0 = (s + 1) * kRows, nrow1 = min(nrow0 + kRows, nrows);
                            const uint32_t nwrow = nrow0 + wave * (TPW * (uint32_t)RPT);
                            if (nwrow < nrow1) {
                                const uint32_t b0 = __builtin_amdgcn_readlane(tbl_next, 0);
                                const uint32_t b1 = __builtin_amdgcn_readlane(tbl_next, 1);
                                stream_load<T, RPT>(nxt, rowptr, col16, vals, nwrow, nrow1, b0, b1, lane, ulen_next);
                                fetched_next = true;
                            }
                        }
                        if (!OV || !((ovmask >> k) & 1u))  // oversized tiles: csr_spmv_overflow
                            stream_compute<T, RPT, SKEW>(cur, xw, wmax, prod, y, r0, row1, lane, nt_store);
                        if (more || fetched_next) cur = nxt;
                    }
                };
                if (ovmask == 0u) tiles(std::false_type{});
                else tiles(std::true_type{});
            }
            cur_valid = fetched_next;
        } else if (d.z == kModeStreamGlobal && (dw_raw & 2u) && (flags & 2u)) {
            cur_valid = false;   // a wide band's super-tile: csr_spmv_panel takes it
        } else if (d.z == kModeStreamGlobal) {
            cur_valid = false;   // (no window involved: no barrier needed)
            stream_global_super_tile<T, TPW, RPT, SKEW>(rowptr, colind, vals, x, y, prod, row0, row1, ncols,
                                                  nt_store, skip_bits, ulen);
        } else {
            cur_valid = false;
            __syncthreads();
            if (d.z == kModeVectorLds) {
                stage_window<T, kStreamBlock>(xw, x, d.x, d.y);
                __syncthreads();
                vector_rows<T, L, U, true, USE_DPP, kStreamBlock, 4>(rowptr, colind, vals, x, xw, y, row0,
                                                                  row1, d.x, last_nz);
            } else {
                vector_rows<T, L, U, false, USE_DPP, kStreamBlock, 4>(rowptr, colind, vals, x, nullptr, y,
                                                                   row0, row1, 0u, last_nz);
            }
        }
        tbl = tbl_next;
    }
}

// ---- the oversized tiles of a stream plan ----------------------------------------------
// tiles[i] = first row of a tile of rpt <= 64 rows that holds more than 1024 entries: typically one or
// a few heavy rows among light ones.  A workgroup per tile; every wave reads the tile's row bounds
// (a row per lane) and sorts the rows into three classes by length:
//   light  (<= 128 entries): 8 lanes per row, 32 rows per trip;
//   medium (<= 1024):        a wave per row, the four waves taking turns;
//   heavy:                   the whole workgroup per row, partial sums of the four waves folded in LDS.
// Each class keeps four (column, value) pairs per lane in flight; x is gathered from global memory.
// (A wave per row for every row: 156 000 mostly idle waves cost 50 us for 2400 such tiles; 16 lanes per row:
// a row of 3000 entries is 47 dependent trips.)
constexpr uint32_t kOverflowLight = 128, kOverflowMedium = 1024;

template <typename T, uint32_t S>   // partial sum of entries b + s, b + s + S, ... < e
__device__ __forceinline__ T strided_row_sum(const uint32_t *__restrict__ colind, const T *__restrict__ vals,
                                             const T *__restrict__ x, uint32_t b, uint32_t e, uint32_t s) {
    T acc = T(0);
    uint32_t k = b + s;
    for (; k + 3 * S < e; k += 4 * S) {
        const uint32_t c0 = __builtin_nontemporal_load(colind + k), c1 = __builtin_nontemporal_load(colind + k + S),
                       c2 = __builtin_nontemporal_load(colind + k + 2 * S),
                       c3 = __builtin_nontemporal_load(colind + k + 3 * S);
        const T v0 = __builtin_nontemporal_load(vals + k), v1 = __builtin_nontemporal_load(vals + k + S),
                v2 = __builtin_nontemporal_load(vals + k + 2 * S), v3 = __builtin_nontemporal_load(vals + k + 3 * S);
        const T x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
        acc += v0 * x0;
        acc += v1 * x1;
        acc += v2 * x2;
        acc += v3 * x3;
    }
    for (; k < e; k += S) acc += __builtin_nontemporal_load(vals + k) * x[__builtin_nontemporal_load(colind + k)];
    return acc;
}

template <typename T>
__global__ __launch_bounds__(kStreamBlock) void csr_spmv_overflow(
    const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind, const T *__restrict__ vals,
    const T *__restrict__ x, T *__restrict__ y, const uint32_t *__restrict__ tiles, uint32_t ntiles,
    uint32_t rpt, uint32_t nrows) {
    __shared__ T part[kStreamWaves];
    const uint32_t t = blockIdx.x;
    if (t >= ntiles) return;
    const uint32_t r0 = tiles[t];
    const uint32_t r1 = min(r0 + rpt, nrows);
    const uint32_t nrow = r1 - r0;                       // <= 64
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    // lane i: bounds of row r0 + i (lanes past the tile: an empty row)
    const uint32_t b_l = rowptr[min(r0 + lane, r1)];
    const uint32_t e_l = rowptr[min(r0 + lane + 1, r1)];
    const uint32_t len_l = e_l - b_l;
    const uint64_t medium = __ballot(len_l > kOverflowLight && len_l <= kOverflowMedium);
    uint64_t heavy = __ballot(len_l > kOverflowMedium);

    // light rows: 8 lanes per row
    {
        const uint32_t g = threadIdx.x >> 3, s = threadIdx.x & 7u;   // 32 row groups
        for (uint32_t i = g; i < 64u; i += 32u) {
            const uint32_t b = __shfl(b_l, (int)i), e = __shfl(e_l, (int)i);
            T acc = T(0);
            if (e - b <= kOverflowLight) acc = strided_row_sum<T, 8>(colind, vals, x, b, e, s);
            acc += __shfl_xor(acc, 4);
            acc += __shfl_xor(acc, 2);
            acc += __shfl_xor(acc, 1);
            if (s == 0 && i < nrow && e - b <= kOverflowLight) y[r0 + i] = acc;
        }
    }
    // medium rows: a wave per row
    {
        uint64_t m = medium;
        uint32_t turn = 0;
        while (m) {   // wave-uniform
            const uint32_t i = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            if ((turn++ & (kStreamWaves - 1)) != wave) continue;
            const uint32_t b = __shfl(b_l, (int)i), e = __shfl(e_l, (int)i);
            T acc = strided_row_sum<T, kWave>(colind, vals, x, b, e, lane);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0) y[r0 + i] = acc;
        }
    }
    // heavy rows: the workgroup per row
    while (heavy) {   // block-uniform
        const uint32_t i = (uint32_t)__builtin_ctzll(heavy);
        heavy &= heavy - 1;
        const uint32_t b = __shfl(b_l, (int)i), e = __shfl(e_l, (int)i);
        T acc = strided_row_sum<T, kStreamBlock>(colind, vals, x, b, e, threadIdx.x);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) part[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) y[r0 + i] = (part[0] + part[1]) + (part[2] + part[3]);
        __syncthreads();
    }
}

// ---- the LONG rows of a row-split plan (spal_csr.hip: csr_try_row_split) -----------------------------------------
// rows[i], i < nlong: rows of more than the split's threshold, longest first; the first nheavy (more than 1024 entries) take a
// whole workgroup each and start first.  Otherwise a wave per row: consecutive lanes read consecutive entries
// (coalesced, non-temporal), x gathered through L2, lane-partial sums folded by a shuffle tree -- rounded like every row the
// vector kernels compute (1e-10 parity, not bit-identical).  Rows of a power-law tail, a few thousand entries at most:
// the wave loops; four rows in flight per workgroup.
// partial sum of entries b + s, b + s + S, ... < e with EIGHT entries per lane in flight: all eight (column, value) pairs are
// requested before the first x is gathered (clamped, masked: no branch between the loads), so a row of up to 8 S entries costs
// two dependent round trips, not four
template <typename T, uint32_t S>
__device__ __forceinline__ T strided_row_sum8(const uint32_t *__restrict__ colind, const T *__restrict__ vals,
                                              const T *__restrict__ x, uint32_t b, uint32_t e, uint32_t s) {
    T acc = T(0);
    for (uint32_t k0 = b + s; k0 < e; k0 += 8 * S) {   // (lanes past the row's end skip the trip; e > b here)
        uint32_t c[8];
        T v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t k = min(k0 + u * S, e - 1u);
            c[u] = __builtin_nontemporal_load(colind + k);
            v[u] = __builtin_nontemporal_load(vals + k);
        }
        T xv[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) xv[u] = x[c[u]];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u)
            if (k0 + u * S < e) acc += v[u] * xv[u];
    }
    return acc;
}

template <typename T>
__global__ __launch_bounds__(kStreamBlock) void csr_spmv_row_list(
    const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind, const T *__restrict__ vals,
    const T *__restrict__ x, T *__restrict__ y, const uint32_t *__restrict__ rows, uint32_t nlong, uint32_t nheavy) {
    // rows[3 i ...] = {row, its first entry, one past its last}: the bounds come with the list, not by a second round trip
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (blockIdx.x < nheavy) {   // block-uniform: one of the longest rows (listed first), the whole workgroup on it
        __shared__ T part[kStreamWaves];
        const uint32_t r = rows[3u * blockIdx.x], b = rows[3u * blockIdx.x + 1u], e = rows[3u * blockIdx.x + 2u];
        T acc = strided_row_sum8<T, kStreamBlock>(colind, vals, x, b, e, threadIdx.x);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) part[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) y[r] = (part[0] + part[1]) + (part[2] + part[3]);
        return;
    }
    const uint32_t i = nheavy + (blockIdx.x - nheavy) * kStreamWaves + wave;
    if (i >= nlong) return;   // wave-uniform
    const uint32_t r = rows[3u * i], b = rows[3u * i + 1u], e = rows[3u * i + 2u];
    T acc = strided_row_sum8<T, kWave>(colind, vals, x, b, e, lane);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) y[r] = acc;
}

// setup: the short rows' entries copied into the compacted arrays of the split's short part (a thread per row: setup time)
template <typename T>
__global__ __launch_bounds__(256) void csr_split_copy(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ rowptr_s,
                                                      const uint32_t *__restrict__ colind, const T *__restrict__ vals,
                                                      uint32_t *__restrict__ colind_s, T *__restrict__ vals_s, uint32_t nrows) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const uint32_t b = rowptr[r], bs = rowptr_s[r], n = rowptr_s[r + 1] - bs;   // n = the row's length, or 0 for a long row
    for (uint32_t k = 0; k < n; ++k) {
        colind_s[bs + k] = colind[b + k];
        vals_s[bs + k] = vals[b + k];
    }
}

// y[i] = 0 for an all-empty matrix slice (nnz == 0): nothing to stream.
template <typename T>
__global__ void fill_zero(T *y, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = T(0);
}

}  // namespace spal
