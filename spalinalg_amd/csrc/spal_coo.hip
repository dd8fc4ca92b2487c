// spal_coo.hip -- COO upload, device-side COO -> CSR assembly, and the stable
// radix sort it is built on.
//
// Contract (reference src/csr/conv/coo.rs:4-115):
//   order entries by (row, col), STABLY w.r.t. insertion order;
//   sum every run of equal (row, col) left to right (separately rounded adds);
//   drop sums that compare equal to zero (-0.0 dropped, NaN kept);
//   emit CSR (columns strictly increasing inside a row).
// rowptr / colind / values are bit-identical to the reference's result: every
// reordering step is stable and each run is summed by ONE thread in insertion
// order.
//
// Pipeline (device only; one read-back at the end: the output size and the groups' column spans):
//   0. at upload (functions of the uploaded indices alone; by rows at upload, by columns with the first CSC
//      assembly): the first radix pass's
//      per-tile digit offsets; the offsets of the groups of 2^gbits consecutive rows (about a thousand entries
//      each) in the row-sorted order, and the fullest group, which decides the LDS capacity of step 2
//      (512 ... 2048 entries).
//   1. stable LSD radix sort by the ROW BITS ABOVE gbits only (8 bits per pass: 2 passes at config 5), carrying
//      (col, value) as payload -- the entries of a group end up contiguous, still in insertion order.  Each pass:
//      per-tile digit histogram -> scan -> scatter that first reorders the tile in LDS so every digit leaves as
//      one contiguous, coalesced run.
//   2. one workgroup per group, everything in LDS and entry-parallel: counting sort by the low row bits, stable
//      rank by column inside each row, run heads sum their runs in insertion order, zeros dropped; the group's
//      place in the result comes from a decoupled look-back over the groups before it, and the survivors and
//      the rowptr of the group's rows are written once, at their final offsets.
// If some group holds more than 2048 entries the assembly runs the general
// route -- LSD passes over the column bits first, then all row bits -- and a
// lane-sequential run summation, which is correct for any input, only slower.
#include "spal_internal.hpp"

#include <numeric>

namespace spal {

// --------------------------------------------------------------------------
// exclusive scan of u32 (generic, two levels)
// --------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;  // 2048

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= (uint32_t)o) v += t;
    }
    return v;
}

// Counters in LDS that the lanes of ONE wave hand to each other between two rounds of a ranking loop (the lowest lane of
// a digit publishes the new count, the next round's lanes read it).  The compiler must re-read them every round; declared
// `volatile` it did -- but through FLAT instructions (address-space inference leaves volatile accesses alone), each followed
// by s_waitcnt vmcnt(0): 32 serialised flat round trips per tile in radix_scatter, and in the group kernel a wait for every
// load in flight.  Relaxed atomics at wavefront scope are plain ds_read / ds_write, re-read every time, and LDS
// instructions of one wave execute in order.
__device__ __forceinline__ uint32_t lds_peek(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void lds_poke(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// block-wide exclusive scan of one value per thread (256 threads); returns the
// exclusive prefix, *total receives the block sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *total) {
    __shared__ uint32_t wsum[kScanThreads / 64];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t i = 0; i < kScanThreads / 64; ++i) {
        const uint32_t s = wsum[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(kScanThreads) void scan_tile_sums(const uint32_t *__restrict__ in,
                                                               uint64_t n,
                                                               uint32_t *__restrict__ sums) {
    const uint64_t t0 = (uint64_t)blockIdx.x * kScanTile;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        const uint64_t i = t0 + (uint64_t)j * kScanThreads + threadIdx.x;
        if (i < n) acc += in[i];
    }
    uint32_t total;
    (void)block_exclusive_scan(acc, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// single workgroup: exclusive scan of `sums` in place; total -> *grand (may be NULL)
__global__ __launch_bounds__(kScanThreads) void scan_sums_inplace(uint32_t *sums, uint32_t m,
                                                                  uint32_t *grand) {
    uint32_t carry = 0;
    for (uint32_t b = 0; b < m; b += kScanThreads) {
        const uint32_t i = b + threadIdx.x;
        const uint32_t v = i < m ? sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, &total);
        if (i < m) sums[i] = carry + ex;
        carry += total;
    }
    if (grand && threadIdx.x == 0) *grand = carry;
}

// out[i] = prefix; when `closing` is set out[n] = grand total as well
__global__ __launch_bounds__(kScanThreads) void scan_apply(const uint32_t *__restrict__ in,
                                                           uint32_t *__restrict__ out, uint64_t n,
                                                           const uint32_t *__restrict__ sums,
                                                           int closing) {
    // thread owns kScanItems CONSECUTIVE elements so the scan order is the array order
    const uint64_t t0 = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        v[j] = (t0 + j < n) ? in[t0 + j] : 0u;
        acc += v[j];
    }
    uint32_t total;
    uint32_t ex = block_exclusive_scan(acc, &total) + sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        if (t0 + j < n) out[t0 + j] = ex;
        ex += v[j];
        if (closing && t0 + j + 1 == n) out[n] = ex;
    }
}

// out[i] = sum in[0..i); *d_total (device, may be NULL) = sum of all; with
// `closing`, out must have n + 1 entries and out[n] = the total.  `sums` must
// hold ceil(n / kScanTile) u32.  in == out allowed.
static hipError_t exclusive_scan_u32(const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *sums,
                                     uint32_t *d_total, hipStream_t st, bool closing = false) {
    if (n == 0) {
        hipError_t e = hipSuccess;
        if (d_total) e = hipMemsetAsync(d_total, 0, sizeof(uint32_t), st);
        if (e == hipSuccess && closing) e = hipMemsetAsync(out, 0, sizeof(uint32_t), st);
        return e;
    }
    const uint32_t tiles = (uint32_t)((n + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL(scan_tile_sums, dim3(tiles), dim3(kScanThreads), 0, st, in, n, sums);
    hipLaunchKernelGGL(scan_sums_inplace, dim3(1), dim3(kScanThreads), 0, st, sums, tiles, d_total);
    hipLaunchKernelGGL(scan_apply, dim3(tiles), dim3(kScanThreads), 0, st, in, out, n, sums,
                       closing ? 1 : 0);
    return hipGetLastError();
}

// --------------------------------------------------------------------------
// stable LSD radix sort: (u32 key, u32 aux, T value), 8 bits per pass
// --------------------------------------------------------------------------
#ifndef SPAL_SORT_THREADS
#define SPAL_SORT_THREADS 256
#endif
#ifndef SPAL_SORT_XCD
#define SPAL_SORT_XCD 1
#endif
#ifndef SPAL_SORT_ITEMS
#define SPAL_SORT_ITEMS 16
#endif
constexpr int kSortThreads = SPAL_SORT_THREADS;           // scatter workgroup
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortItems = SPAL_SORT_ITEMS;               // per thread
constexpr int kSortTile = kSortThreads * kSortItems;      // 4096 entries per workgroup
constexpr int kWaveChunk = 64 * kSortItems;               // consecutive entries per wave
constexpr int kHistThreads = 256;
constexpr int kHistItems = kSortTile / kHistThreads;

// The digit counts of a pass and what the scatter derives its offsets from (round 4: no scan over all 256 x tiles counts --
// three launches and 24 us per pass at config 5 -- any more):
//   raw[d * stride + t]   keys of tile t with digit d, as counted (stride = tiles rounded up to whole groups of 16);
//   gt[d * groups + g]    the total of digit d over group g's 16 tiles; after digit_scan: the digit's keys in the groups
//                         BEFORE g (exclusive, inside the digit);
//   dt[d]                 all keys with digit d.
// Where tile t's keys with digit d go:  (sum of dt over smaller digits: 256 values, scanned by the scatter workgroup itself)
//   + gt[d][t / 16] + the raw counts of the tiles of t's group before t (at most 15 words of one 64-byte line).
struct PassCounts {
    uint32_t *raw = nullptr, *gt = nullptr, *dt = nullptr;
};
// radix_hist: the order inside a tile does not matter here: 16-byte loads, 4 keys per lane.  A workgroup counts kHistGroup
// consecutive tiles and writes, per digit, their counts as ONE run of kHistGroup words (one tile per workgroup wrote 107 MB
// for 12.5 MB of counts at config 5: profiles/r03/pmc_traffic.txt), and the run's total.
constexpr int kHistGroup = 16;
__global__ __launch_bounds__(kHistThreads) void radix_hist(const uint32_t *__restrict__ keys,
                                                           uint64_t len, uint32_t shift,
                                                           uint32_t *__restrict__ raw, uint32_t *__restrict__ gt,
                                                           uint32_t nblk, uint32_t stride, uint32_t groups) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t h[kHistGroup][257];   // (257: the transposed read below walks a column)
    for (int j = 0; j < kHistGroup; ++j) h[j][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t blk0 = blockIdx.x * kHistGroup;
    static_assert(kSortTile % (4 * kHistThreads) == 0, "tile = whole rounds of 4 keys per thread");
    for (int g = 0; g < kHistGroup; ++g) {   // uniform
        const uint32_t blk = blk0 + (uint32_t)g;
        if (blk >= nblk) break;
        const uint64_t t0 = (uint64_t)blk * kSortTile;  // multiple of 4: 16-byte aligned
        if (t0 + kSortTile <= len) {
            u32x4 k[kHistItems / 4];
#pragma unroll
            for (int j = 0; j < kHistItems / 4; ++j)
                k[j] = *reinterpret_cast<const u32x4 *>(keys + t0 + ((uint64_t)j * kHistThreads + threadIdx.x) * 4);
#pragma unroll
            for (int j = 0; j < kHistItems / 4; ++j) {
                atomicAdd(&h[g][(k[j].x >> shift) & 0xffu], 1u);
                atomicAdd(&h[g][(k[j].y >> shift) & 0xffu], 1u);
                atomicAdd(&h[g][(k[j].z >> shift) & 0xffu], 1u);
                atomicAdd(&h[g][(k[j].w >> shift) & 0xffu], 1u);
            }
        } else {
#pragma unroll
            for (int j = 0; j < kHistItems; ++j) {
                const uint64_t i = t0 + (uint64_t)j * kHistThreads + threadIdx.x;
                if (i < len) atomicAdd(&h[g][(keys[i] >> shift) & 0xffu], 1u);
            }
        }
    }
    __syncthreads();
    // sixteen lanes write one digit's run of sixteen counts (64 contiguous, aligned bytes; zeros for tiles beyond the last),
    // a wave four digits' runs
    const uint32_t g = threadIdx.x % kHistGroup;
    for (uint32_t d = threadIdx.x / kHistGroup; d < 256; d += kHistThreads / kHistGroup)
        raw[(uint64_t)d * stride + blk0 + g] = h[g][d];
    {   // thread d: the group's total of digit d
        const uint32_t d = threadIdx.x;
        uint32_t tot = 0;
#pragma unroll
        for (int j = 0; j < kHistGroup; ++j) tot += h[j][d];
        gt[(uint64_t)d * groups + blockIdx.x] = tot;
    }
}

// Workgroup d: gt[d][.] -> its exclusive prefix in place (the digit's keys in earlier groups), dt[d] = the digit's total.
__global__ __launch_bounds__(256) void digit_scan(uint32_t *__restrict__ gt, uint32_t *__restrict__ dt, uint32_t groups) {
    uint32_t *row = gt + (uint64_t)blockIdx.x * groups;
    uint32_t carry = 0;
    for (uint32_t b = 0; b < groups; b += 256) {
        const uint32_t i = b + threadIdx.x;
        const uint32_t v = i < groups ? row[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, &total);
        if (i < groups) row[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) dt[blockIdx.x] = carry;
}

// Stable scatter of one tile.  Wave w owns the tile's entries [w*1024, (w+1)*1024)
// and walks them 64 at a time, so tile order = (wave, round, lane).  The rank
// of an entry among the tile's entries with the same digit is
//   entries of earlier waves + entries of earlier rounds of this wave +
//   earlier lanes of this round,
// computed from ballots and per-wave counters without atomics: deterministic
// and stable.  The tile is first written to LDS in digit order, then copied out
// linearly, so each digit leaves the workgroup as ONE contiguous run
// (coalesced stores) that starts at the scanned global offset of (digit, tile).
// PACK (the last pass before the group kernel, when the minor index and the row inside its group fit one word): the
// key is not written at all and the payload leaves as aux | (key & (2^pack_bits - 1)) << (32 - pack_bits) -- the group a
// sorted entry belongs to is its position, all the group kernel still needs of the row are its low bits: 12 instead of 16
// bytes per entry written here and read there.
template <typename T, bool PACK = false>
__global__ __launch_bounds__(kSortThreads) void radix_scatter(
    const uint32_t *__restrict__ kin, const uint32_t *__restrict__ ain, const T *__restrict__ vin,
    uint32_t *__restrict__ kout, uint32_t *__restrict__ aout, T *__restrict__ vout, uint64_t len,
    uint32_t shift, const uint32_t *__restrict__ raw, const uint32_t *__restrict__ gt, const uint32_t *__restrict__ dt,
    uint32_t nblk, uint32_t stride, uint32_t groups, uint32_t per_xcd, uint32_t pack_bits = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char spal_sort_smem[];
    T *s_val = reinterpret_cast<T *>(spal_sort_smem);                       // kSortTile
    uint32_t *s_key = reinterpret_cast<uint32_t *>(s_val + kSortTile);      // kSortTile
    uint32_t *s_aux = s_key + kSortTile;                                    // kSortTile
    // lanes of a wave hand counts to each other through this array between two rounds (lds_peek / lds_poke)
    uint32_t *cnt = s_aux + kSortTile;                                       // [kSortWaves][256]
    uint32_t *s_start = cnt + kSortWaves * 256;                              // [256] tile-local digit start
    uint32_t *s_delta = s_start + 256;                                       // [256] global - local
    uint32_t *s_wsum = s_delta + 256;                                        // [4] + [4] digit-scan wave sums (tile-local starts, digit bases)

    // tiles that run side by side on one XCD are neighbours in tile order, so the
    // partial cache lines they leave at the end of each digit's run meet in one L2
#if SPAL_SORT_XCD
    const uint32_t tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
#else
    const uint32_t tile = blockIdx.x;
#endif
    if (tile >= nblk) return;  // block-uniform
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < kSortWaves * 256; i += kSortThreads) cnt[i] = 0;
    __syncthreads();

    const uint64_t tile0 = (uint64_t)tile * kSortTile;
    const uint64_t w0 = tile0 + (uint64_t)w * kWaveChunk;
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t key[kSortItems], aux[kSortItems], rank[kSortItems];
    T val[kSortItems];
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = w0 + (uint64_t)j * 64 + lane;
        const bool ok = i < len;
        key[j] = ok ? kin[i] : 0u;
        aux[j] = ok ? ain[i] : 0u;
        val[j] = ok ? vin[i] : T(0);
    }
    // thread d: where this tile's keys with digit d go, apart from the digits' bases (PassCounts) -- requested behind the
    // tile's entries, summed when the ranks are done (asked for first and summed at once they held the entries' loads back:
    // 379 instead of 351 us for the first pass at config 5)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t my_dt = 0, my_gt = 0;
    u32x4 my_line[kHistGroup / 4];
    if (threadIdx.x < 256) {
        const uint32_t d = threadIdx.x, grp = tile / (uint32_t)kHistGroup;
        my_dt = dt[d];
        my_gt = gt[(uint64_t)d * groups + grp];
        const u32x4 *line = reinterpret_cast<const u32x4 *>(raw + (uint64_t)d * stride + (uint64_t)grp * kHistGroup);
#pragma unroll
        for (int q = 0; q < kHistGroup / 4; ++q) my_line[q] = line[q];
    }
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = w0 + (uint64_t)j * 64 + lane;
        const bool ok = i < len;
        const uint32_t d = (key[j] >> shift) & 0xffu;
        // lanes of this round with the same digit (inactive tail lanes excluded)
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = ok ? lds_peek(&cnt[w * 256 + d]) : 0u;
        rank[j] = before + (uint32_t)__popcll(peers & lt);
        // the lowest peer lane publishes the new count (one writer per digit)
        if (ok && (peers & lt) == 0) lds_poke(&cnt[w * 256 + d], before + (uint32_t)__popcll(peers));
    }
    __syncthreads();
    // per digit: exclusive prefix over the waves; tile-local start of the digit;
    // distance between the digit's global run and its place in the tile
    {
        const uint32_t d = threadIdx.x;  // the first 256 threads (whole waves) take the 256 digits
        uint32_t run = 0, inc = 0, inc_dt = 0;
        if (d < 256) {
#pragma unroll
            for (int ww = 0; ww < kSortWaves; ++ww) {
                const uint32_t c = cnt[ww * 256 + d];
                cnt[ww * 256 + d] = run;
                run += c;
            }
            inc = wave_inclusive_scan(run);
            inc_dt = wave_inclusive_scan(my_dt);
            if (lane == 63) { s_wsum[w] = inc; s_wsum[4 + w] = inc_dt; }
        }
        __syncthreads();
        if (d < 256) {
            uint32_t base = 0, base_dt = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i)
                if (i < w) { base += s_wsum[i]; base_dt += s_wsum[4 + i]; }
            const uint32_t start = base + inc - run;
            s_start[d] = start;
            uint32_t my_before = my_gt;
            const uint32_t in_grp = tile % (uint32_t)kHistGroup;
#pragma unroll
            for (int q = 0; q < kHistGroup / 4; ++q)
                my_before += ((uint32_t)(4 * q) < in_grp ? my_line[q].x : 0u) + ((uint32_t)(4 * q + 1) < in_grp ? my_line[q].y : 0u) +
                             ((uint32_t)(4 * q + 2) < in_grp ? my_line[q].z : 0u) + ((uint32_t)(4 * q + 3) < in_grp ? my_line[q].w : 0u);
            s_delta[d] = (base_dt + inc_dt - my_dt) + my_before - start;   // keys with smaller digits + digit d's keys in earlier tiles
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = w0 + (uint64_t)j * 64 + lane;
        if (i < len) {
            const uint32_t d = (key[j] >> shift) & 0xffu;
            const uint32_t lp = s_start[d] + cnt[w * 256 + d] + rank[j];
            s_key[lp] = key[j];
            s_aux[lp] = aux[j];
            s_val[lp] = val[j];
        }
    }
    __syncthreads();
    const uint32_t n_tile = (uint32_t)min((uint64_t)kSortTile, len - tile0);
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const uint32_t lp = j * kSortThreads + threadIdx.x;
        if (lp < n_tile) {
            const uint32_t k = s_key[lp];
            const uint32_t gp = s_delta[(k >> shift) & 0xffu] + lp;
            if (PACK) {
                aout[gp] = pack_bits ? (s_aux[lp] | ((k & ((1u << pack_bits) - 1u)) << (32u - pack_bits))) : s_aux[lp];
            } else {
                kout[gp] = k;
                aout[gp] = s_aux[lp];
            }
            vout[gp] = s_val[lp];
        }
    }
}

template <typename T>
struct SortBuffers {
    uint32_t *key[2] = {nullptr, nullptr};
    uint32_t *aux[2] = {nullptr, nullptr};
    T *val[2] = {nullptr, nullptr};
    PassCounts counts;           // raw 256 * stride, gt 256 * groups, dt 256
    PassCounts counts2;          // the same again: the second pass's, when the first pass's counts must survive it
    uint32_t *sums = nullptr;    // scan scratch (general route)
};
static uint32_t sort_tiles(uint64_t len) { return (uint32_t)((len + kSortTile - 1) / kSortTile); }
static uint32_t sort_groups(uint64_t len) { return (sort_tiles(len) + kHistGroup - 1) / kHistGroup; }
static uint32_t sort_stride(uint64_t len) { return sort_groups(len) * kHistGroup; }

template <typename T>
static size_t sort_lds_bytes() {
    return (size_t)kSortTile * (sizeof(T) + 8) + (size_t)(kSortWaves * 256 + 512 + 8) * 4;
}

// Sorts by bits [lo_bit, lo_bit + nbits) of key, stably.  The first pass reads
// (k_in, a_in, v_in) when given (the caller's arrays, left untouched), else
// buffer set `cur`; `cur` is updated to the set that holds the result.
// `two_counts`: the second pass counts into b.counts2, so that the first pass's scanned counts (the digit buckets'
// starts) are still there afterwards.  `pack_bits` >= 0: the LAST pass writes the packed payload (radix_scatter<T, true>)
// and no keys.
template <typename T>
static hipError_t radix_sort_bits(SortBuffers<T> &b, uint64_t len, uint32_t lo_bit, uint32_t nbits,
                                  int &cur, hipStream_t st, const uint32_t *k_in = nullptr,
                                  const uint32_t *a_in = nullptr, const T *v_in = nullptr,
                                  bool two_counts = false, int pack_bits = -1) {
    if (len == 0) return hipSuccess;
    const uint32_t nblk = sort_tiles(len), groups = sort_groups(len), stride = sort_stride(len);
    const size_t lds = sort_lds_bytes<T>();
    {  // > 64 KiB of dynamic LDS needs the cap raised (per device; cheap, so every call)
        hipError_t e = hipFuncSetAttribute((const void *)radix_scatter<T, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess && pack_bits >= 0)
            e = hipFuncSetAttribute((const void *)radix_scatter<T, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int pass = 0;
    for (uint32_t shift = lo_bit; shift < lo_bit + nbits; shift += 8, ++pass) {
        const uint32_t *ki = k_in ? k_in : b.key[cur];
        const uint32_t *ai = k_in ? a_in : b.aux[cur];
        const T *vi = k_in ? v_in : b.val[cur];
        const int dst = k_in ? cur : (cur ^ 1);
        const PassCounts &pc = (two_counts && pass == 1) ? b.counts2 : b.counts;
        hipLaunchKernelGGL(radix_hist, dim3(groups), dim3(kHistThreads), 0, st, ki, len, shift, pc.raw, pc.gt, nblk, stride, groups);
        hipLaunchKernelGGL(digit_scan, dim3(256), dim3(256), 0, st, pc.gt, pc.dt, groups);
        const uint32_t per_xcd = (nblk + 7) / 8;
        const bool last = shift + 8 >= lo_bit + nbits;
        if (last && pack_bits >= 0)
            hipLaunchKernelGGL((radix_scatter<T, true>), dim3(SPAL_SORT_XCD ? per_xcd * 8 : nblk), dim3(kSortThreads),
                               lds, st, ki, ai, vi, b.key[dst], b.aux[dst], b.val[dst], len, shift, pc.raw, pc.gt, pc.dt,
                               nblk, stride, groups, per_xcd, (uint32_t)pack_bits);
        else
            hipLaunchKernelGGL((radix_scatter<T, false>), dim3(SPAL_SORT_XCD ? per_xcd * 8 : nblk), dim3(kSortThreads),
                               lds, st, ki, ai, vi, b.key[dst], b.aux[dst], b.val[dst], len, shift, pc.raw, pc.gt, pc.dt,
                               nblk, stride, groups, per_xcd, 0u);
        cur = dst;
        k_in = nullptr;
    }
    return hipGetLastError();
}

// The groups' offsets after exactly TWO passes, from the passes' own counts (round 4; round 3 read all sorted keys once
// more for them, rows_boundaries: 206 MB and 45 us at config 5).  Group g = d2 << 8 | d1 (d1 = the first pass's digit, d2
// = the second's).  The second pass's input is ordered by d1: bucket d1 begins at B[d1] = the keys with a smaller first
// digit, inside tile t* = B[d1] / tile.  Entries ordered before group g in the result: every entry with a smaller d2, and
// of those with the same d2 the ones in buckets before d1 -- that is what the second pass's counts say about (d2, tiles
// before t*) (PassCounts) plus the entries with digit d2 inside tile t* that lie before B[d1], which workgroup d1 counts
// here (at most one tile of keys).
__global__ __launch_bounds__(256) void group_offsets(const uint32_t *__restrict__ dt1, const uint32_t *__restrict__ raw2,
                                                     const uint32_t *__restrict__ gt2, const uint32_t *__restrict__ dt2,
                                                     const uint32_t *__restrict__ keys1, uint32_t len, uint32_t nblk,
                                                     uint32_t stride, uint32_t groups, uint32_t shift2, uint32_t ngroups,
                                                     uint32_t *__restrict__ gstart) {
    __shared__ uint32_t h[256], s_b;
    const uint32_t d1 = blockIdx.x, t = threadIdx.x;
    h[t] = 0;
    uint32_t total;
    const uint32_t b_mine = block_exclusive_scan(dt1[t], &total);      // B[t]
    if (t == d1) s_b = b_mine;
    const uint32_t my_dt2 = dt2[t];
    const uint32_t base2 = block_exclusive_scan(my_dt2, &total);       // keys with a second digit below t (has barriers: s_b, h are set)
    const uint32_t b = s_b;
    const uint32_t tstar = b / (uint32_t)kSortTile, t0 = tstar * (uint32_t)kSortTile;
    for (uint32_t i = t0 + t; i < b; i += 256) atomicAdd(&h[(keys1[i] >> shift2) & 0xffu], 1u);
    __syncthreads();
    const uint32_t d2 = t, g = d2 << 8 | d1;
    if (g < ngroups) {
        uint32_t v = base2;
        if (tstar < nblk) {
            const uint32_t grp = tstar / (uint32_t)kHistGroup, in_grp = tstar % (uint32_t)kHistGroup;
            v += gt2[(uint64_t)d2 * groups + grp] + h[d2];
            for (uint32_t j = 0; j < in_grp; ++j) v += raw2[(uint64_t)d2 * stride + (uint64_t)grp * kHistGroup + j];
        } else {
            v += my_dt2;   // (the bucket begins at the very end: nothing of it exists, every key with this second digit lies before)
        }
        gstart[g] = v;
    }
    if (d1 == 0 && t == 0) gstart[ngroups] = len;
}

static uint32_t bits_for(uint64_t n) {  // bits needed for values in [0, n)
    uint32_t b = 0;
    while (b < 64 && (1ull << b) < n) ++b;
    return b ? b : 1;
}

// --------------------------------------------------------------------------
// after the row sort
// --------------------------------------------------------------------------
// Row starts of an array of keys that is sorted by (key >> shift): with shift = 0
// rows, otherwise groups of 2^shift consecutive rows ("row" below = key >> shift).
// start[r] = first sorted entry whose row is >= r   (r in [0, nrows])
__global__ __launch_bounds__(256) void rows_lower_bound(const uint32_t *__restrict__ sorted_row,
                                                        uint32_t n, uint32_t nrows, uint32_t shift,
                                                        uint32_t *__restrict__ start) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r > nrows) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)(sorted_row[mid] >> shift) < r) lo = mid + 1; else hi = mid;
    }
    start[r] = lo;
}

// start[r] = first sorted entry whose row is >= r, r in [0, nrows], by ONE
// streaming pass over the sorted keys: entry i with row[i] != row[i-1] is the
// first of its row and of every empty row in between.  (The binary search above
// costs 26 dependent loads per row; this reads every key once.)
__global__ __launch_bounds__(256) void rows_boundaries(const uint32_t *__restrict__ sorted_row,
                                                       uint32_t n, uint32_t nrows, uint32_t shift,
                                                       uint32_t *__restrict__ start) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // four consecutive entries per thread (one 16-byte load) + the key before them
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 > n) return;
    // virtual row -1 before the first entry (0xffffffff + 1 == 0), nrows after the last one
    uint32_t prev = i0 == 0 ? 0xffffffffu : sorted_row[i0 - 1] >> shift;
    uint32_t k[4];
    if (i0 + 4 <= n) {
        const u32x4 q = *reinterpret_cast<const u32x4 *>(sorted_row + i0);
        k[0] = q.x >> shift; k[1] = q.y >> shift; k[2] = q.z >> shift; k[3] = q.w >> shift;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = (i0 + j < n) ? sorted_row[i0 + j] >> shift : nrows;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t i = i0 + j;
        if (i > n) break;
        // rows prev + 1 .. k[j] start at i (empty unless the row changes here)
        for (uint32_t r = prev + 1u; r <= k[j]; ++r) start[r] = (uint32_t)i;
        prev = k[j];
    }
}

// start[] of a sorted key array: the streaming pass, unless rows outnumber entries
// so much that one thread of it would fill long stretches of empty rows
static void launch_row_starts(const uint32_t *sorted_row, uint32_t n, uint32_t nrows, uint32_t *start,
                              hipStream_t st, uint32_t shift = 0) {
    if ((uint64_t)nrows > 8ull * n + 1024)
        hipLaunchKernelGGL(rows_lower_bound, dim3((uint32_t)(((uint64_t)nrows + 1 + 255) / 256)), dim3(256), 0,
                           st, sorted_row, n, nrows, shift, start);
    else
        hipLaunchKernelGGL(rows_boundaries, dim3((uint32_t)(((uint64_t)n / 4 + 1 + 255) / 256)), dim3(256), 0,
                           st, sorted_row, n, nrows, shift, start);
}

constexpr int kGroupCap = 2048;  // entries a group of rows may hold for the LDS local sort

// *fullest = max(*fullest, entries of the fullest group): one atomicMax per workgroup (a few hundred at most -- thousands
// of waves raising one shared maximum would serialise on it)
__global__ __launch_bounds__(256) void groups_check(const uint32_t *__restrict__ gstart, uint32_t ngroups,
                                                    uint32_t *__restrict__ fullest) {
    __shared__ uint32_t s_max[4];
    uint32_t v = 0;
#pragma unroll 4
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (uint64_t)gridDim.x * 256)
        v = max(v, gstart[g + 1] - gstart[g]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(fullest, max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));
}

// The local sort.  The radix passes order the entries by the row bits ABOVE gbits
// only, so a group of 2^gbits consecutive rows is one contiguous segment
// [gstart[grp], gstart[grp + 1]) that still holds its entries in insertion order.
// One workgroup (4 waves) per group finishes the job in LDS -- in effect the last
// radix pass, the per-row column sort, the duplicate sums and the zero drop in
// one kernel, with one read and one write of the data.  All phases are
// entry-parallel:
//   0. one batch of global loads: (row, col, val) of every entry -> registers.
//      Wave w owns the entries [w * chunk, (w + 1) * chunk) and walks them 64 at
//      a time, so insertion order = (wave, round, lane).
//   1. stable counting sort by the low row bits: the rank of an entry among the
//      group's entries of the same row = entries of earlier waves + earlier
//      rounds of this wave + earlier lanes of this round (ballots and per-wave
//      counters, no atomics); a scan of the row totals gives the row starts rs[].
//      The column goes to c1[rs[row] + rank]: rows contiguous, insertion order
//      inside each row.
//   2. the stable rank of an entry by column inside its row = number of entries
//      j of the row with col_j < col_i, or col_j == col_i and j before i.  The
//      entry is scattered to position rs[row] + rank of c2 / v2 / r2: the group
//      is now sorted by (row, col), equal (row, col) in insertion order.
//      (Quadratic in the row length, which the group capacity bounds.)
//   3. a sorted position starts a run when its (row, col) differs from its
//      predecessor's; the head sums its run left to right = insertion order
//      (coo.rs:42-46); sums that compare equal to zero are dropped (coo.rs:64).
//   4. survivors are numbered in sorted order (ballots + a scan of the 4 x K
//      wave counts); the group learns how many survivors the groups before it
//      hold (group_lookback) and writes its own at their FINAL offsets of
//      colind / values, and rowptr of its rows (LDS counters + a scan).
// state[g] of the look-back below: (status << 32) | count, status 0 = nothing yet, 1 = the group's own number of
// survivors, 2 = survivors of groups 0 ... g inclusive.
constexpr unsigned long long kGroupOwn = 1ull << 32, kGroupUpTo = 2ull << 32;
#ifndef SPAL_COO_LBW
#define SPAL_COO_LBW 1
#endif
constexpr uint32_t kLookbackSpins = 1u << 21;   // (seconds: a bound, so that every wave reaches its exit; SPAL_COO_LOOKBACK_SPINS overrides)

// Survivors in all groups before `grp`, for the group that holds `total` of its own: decoupled look-back over the
// groups' 8-byte state words (wave 0 of the workgroup, all 64 lanes: 64 predecessors per round).  The count travels
// IN the word that flags it (relaxed agent-scope stores / loads: written through, read past L1), so no release /
// acquire fence is paid -- with fences (an L2 write-back per group) this form lost to a separate pack kernel.
// Progress: a group waits only for groups with a SMALLER id, and ids are handed out by a device ticket (one atomicAdd
// per workgroup, coo_group_sort) in the order in which workgroups actually start: whoever holds id g started after the
// holders of 0 ... g - 1, which are therefore resident or finished -- the lowest unfinished group waits for nobody,
// whatever order the dispatcher takes the workgroups in.  The spin bound stays as a backstop (a wave that gives up
// raises *err bit 0, publishes nothing further and the host repeats the assembly on the general route; forced by
// SPAL_COO_LOOKBACK_SPINS=0 in tests/test_gpu_csc_coo.py).  (A resident grid whose workgroups walk the
// groups b, b + grid, ... with the next group's loads in flight during the look-back needs no ticket either; it
// was measured and is slower: 1.93 vs 1.68 ms per assembly, the static order keeps a fast workgroup from running ahead.)
// Memory order: the only data a successor reads from a predecessor is the count, and it travels in the SAME 8-byte
// word as the status (single-copy atomic 8-byte store / load at agent scope: sc1, written through to / read from the
// memory side of the per-XCD L2s) -- there is no second location whose visibility would have to be ordered against
// the flag, hence relaxed suffices and no release / acquire fence (an L2 write-back per group) is paid.
// Measured (config 5, profiles/r02/coo_lookback.txt): the wait costs 174 us of coo_group_sort's 750 (groups finish in
// order, so a workgroup also waits out every slower predecessor still in flight) against 193 + 32 us for the pack
// kernel and row scan it replaces, and 1.2 GB less traffic.  Polling 128 or 512 predecessors per round trip is slower
// (1.86 / 1.98 vs 1.75 ms per assembly), the sleep between polls does not matter (1 ... 64: 1.75 - 1.79 ms).
__device__ __forceinline__ uint32_t group_lookback(unsigned long long *state, uint32_t grp, uint32_t total,
                                                   uint32_t lane, uint32_t *err, uint32_t spin_bound,
                                                   uint32_t *dbg = nullptr) {
    constexpr int W = SPAL_COO_LBW;
    if (lane == 0)
        __hip_atomic_store(&state[grp], (grp ? kGroupOwn : kGroupUpTo) | total, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (grp == 0) return 0;
#ifdef SPAL_COO_FAKE_LOOKBACK   // lab builds: what the kernel costs WITHOUT the wait (wrong offsets, results discarded)
    if (lane == 0) __hip_atomic_store(&state[grp], kGroupUpTo | (unsigned long long)(grp * 1264u + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return grp * 1264u;
#endif
    uint32_t mine = 0, spins = 0;                         // lane-local part of the sum
    int64_t base = (int64_t)grp - 1;                      // lane 0 looks at the nearest predecessor
    for (;;) {
        // one round trip covers W windows of 64 predecessors (nearest first): W loads per lane issued back to back
        unsigned long long sv[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const int64_t idx = base - (int64_t)lane - 64 * j;
            sv[j] = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : kGroupUpTo;                // before group 0: nothing
        }
        uint32_t part = 0;
        bool wait = false, done = false;
#pragma unroll
        for (int j = 0; j < W; ++j) {                     // (all tests wave-uniform)
            if (wait || done) continue;
            const uint32_t status = (uint32_t)(sv[j] >> 32);
            const uint64_t missing = __ballot(status == 0), upto = __ballot(status == 2);
            // the nearest predecessor that knows its inclusive count ends the walk; everyone nearer must have reported
            const uint64_t need = upto ? ((2ull << __builtin_ctzll(upto)) - 1ull) : ~0ull;
            if (missing & need) { wait = true; continue; }
            if ((need >> lane) & 1ull) part += (uint32_t)sv[j];
            if (upto) done = true;
        }
        if (wait) {
            if (++spins > spin_bound) {
                if (lane == 0) atomicOr(err, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        mine += part;
        if (done) break;
        base -= 64 * W;
    }
    if (dbg) { dbg[0] = spins; dbg[1] = (uint32_t)(((int64_t)grp - 1 - base) / (64 * W)) + 1u; }   // (lab builds: polls that waited, windows walked)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += (uint32_t)__shfl_xor((int)mine, o, 64);
    if (lane == 0)
        __hip_atomic_store(&state[grp], kGroupUpTo | (unsigned long long)(uint32_t)(mine + total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    return mine;
}

// Bitonic sorting network on N registers (N a power of two, fully unrolled: every compare-exchange is one v_min_u32 and
// one v_max_u32 with compile-time directions) -- a row's columns, one row per thread (coo_group_sort, step 2).
template <int N>
__device__ __forceinline__ void bitonic_sort_regs(uint32_t (&k)[N]) {
#pragma unroll
    for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool up = (i & size) == 0;
                    const uint32_t lo = min(k[i], k[j]), hi = max(k[i], k[j]);
                    k[i] = up ? lo : hi;
                    k[j] = up ? hi : lo;
                }
            }
        }
    }
}
// One thread sorts ONE row's columns: reads the row's L <= N columns out of c1[a ...) (row order = insertion order), sorts
// the keys column << 5 | place-in-row (unique, so ties between equal columns fall in insertion order: stable), and writes
// to every entry's slot its place in (row, col) order instead of its column (the entries keep their columns in registers).
template <int N>
__device__ __forceinline__ void sort_row_in_regs(uint32_t *c1, uint32_t a, uint32_t L) {
    uint32_t key[N];
#pragma unroll
    for (int u = 0; u < N; ++u) key[u] = (uint32_t)u < L ? (c1[a + u] << 5 | (uint32_t)u) : 0xffffffffu;
    bitonic_sort_regs<N>(key);
#pragma unroll
    for (int u = 0; u < N; ++u)
        if ((uint32_t)u < L) c1[a + (key[u] & 31u)] = a + (uint32_t)u;
}

// ---- the group kernel (round 4) ------------------------------------------------------------------------------------
// One workgroup per group, as in round 3; what changed:
//  * ids come from kTicketClasses = 8 counters: workgroup b draws from counter b & 7 and takes id = 8 * ticket + (b & 7)
//    (class c has exactly as many workgroups as ids).  One device atomic on ONE address per workgroup is served at 87 M/s
//    (tools/micro/ticket.hip): 39 063 tickets cost 447 us whatever else happens, +0.13 ms per assembly in round 3; eight
//    addresses are served side by side (72 us per 39 063, spread over the launch).
//  * the per-wave counters of the counting sort are read and written as LDS (lds_peek / lds_poke), not as volatile flat
//    accesses with a full wait each.
//  * PACKED: the last radix pass left column | row-in-group << (32 - gbits) in ONE word (12 instead of 16 bytes per entry).
// Progress.  A group waits only for groups with SMALLER ids.  Class c hands its ids out in the order in which its
// workgroups actually start, so inside a class the holder of an id started after the holders of all smaller ids of that
// class.  Let g* be the lowest unfinished id, of class c.  If a workgroup holds it, it waits for nobody.  If nobody holds
// it yet, every class-c workgroup that has started holds a smaller id and is therefore finished and gone; g* goes to the
// next class-c workgroup the dispatcher starts.  That this workgroup does start is where the single counter of round 3
// assumed nothing and this form assumes something: workgroups b & 7 == c run on XCD c (round-robin dispatch), whose slots
// are only ever taken by class-c workgroups -- all finished, so free; what is assumed is that the dispatcher hands XCD c
// its next workgroup while other XCDs are full of waiting workgroups (no head-of-line blocking across XCDs beyond what
// blockIdx order already implies: under strictly ordered dispatch ids equal blockIdx and no workgroup ever waits for one
// dispatched after it).  The spin bound is the backstop it always was: a look-back that gives up raises the error flag and
// the host repeats the assembly on the general route (tested: SPAL_COO_LOOKBACK_SPINS=0); SPAL_COO_TICKET=1 takes the
// single counter again, 0 takes blockIdx.
// Measured and not kept (profiles/r04/coo_assembly.txt): a RESIDENT grid of occupancy x CUs workgroups walking through
// dynamically drawn groups with the next group's entries in flight in a second register set, its bounds and the ticket
// after that in flight too -- 1.39-1.59 ms for this kernel instead of 0.74, whatever the occupancy (3, 4, 5 workgroups per
// CU): every workgroup holds the ids of its next groups while it works on (or waits in the look-back of) the current one,
// every other workgroup's look-back needs those ids' counts, and whoever falls behind by one iteration stalls everyone by
// one iteration.  One group per workgroup lets the dispatcher start the next group the moment a slot is free; a waiting
// workgroup holds nothing anybody needs.
#ifndef SPAL_COO_PRIO
#define SPAL_COO_PRIO 3   // wave priority of the group kernel's phases before its count is published (0: none); -15 us per assembly
#endif
#ifndef SPAL_COO_LB_1536
#define SPAL_COO_LB_1536 7
#endif
constexpr uint32_t kTicketClasses = 8;
// -DSPAL_COO_STAMPS (lab builds): thread 0 of every workgroup of coo_group_sort records wall_clock64() (100 MHz) at its
// phase boundaries into g_coo_stamps[group][8]; the host writes the phases' mean durations to stderr after the assembly
#ifdef SPAL_COO_STAMPS
__device__ unsigned long long *g_coo_stamps = nullptr;
#define SPAL_STAMP(i) do { if (threadIdx.x == 0 && g_coo_stamps) g_coo_stamps[(size_t)stamp_slot * 16 + (i)] = wall_clock64(); } while (0)
#else
#define SPAL_STAMP(i) do { } while (0)
#endif
// state[] tail behind the groups' look-back words: {error flags, fullest group, single ticket, -, tickets[kTicketClasses]}
// The class counters lie kTicketStride words apart: atomics on ONE line are served one after the other whatever their address in
// the line (87 M/s; eight counters in consecutive words were one hot line -- the 1 792 workgroups of the launch's first round waited
// 18 us for their ids, and the steady 62 M tickets/s kept that line 70 % busy), counters 256 bytes apart are served side by side.
#ifndef SPAL_COO_TICKET_STRIDE
#define SPAL_COO_TICKET_STRIDE 64
#endif
constexpr uint32_t kTicketStride = SPAL_COO_TICKET_STRIDE;
constexpr uint32_t kTailWords = 4 + kTicketClasses * kTicketStride;

// ROWSORT: step 2 by the per-row network and the wave-per-long-row pass (columns below 2^27, rows of at most 256 entries: a
// group with a longer row raises *err bit 2 and the host runs the kernel again with ROWSORT = false, where every entry
// counts its place for itself as in rounds 1-3 -- two kernels rather than two paths in one: the unused path's registers
// were spilled by the used one).
template <typename T, int CAP, bool PACKED, bool ROWSORT>
// (workgroups per CU the LDS footprint allows; eight at CAP = 1536 measured behind seven)
__global__ __launch_bounds__(256, CAP == 1536 ? SPAL_COO_LB_1536 : CAP == 2048 ? 5 : 8) void coo_group_sort(const uint32_t *__restrict__ gstart,
                                                      const uint32_t *__restrict__ sorted_row,
                                                      const uint32_t *__restrict__ cols, const T *__restrict__ vals,
                                                      uint32_t nrows, uint32_t gbits, uint32_t ngroups,
                                                      unsigned long long *__restrict__ state, uint32_t *__restrict__ err,
                                                      uint32_t *__restrict__ tickets, uint32_t ticket_classes, uint32_t spin_bound,
                                                      uint32_t *__restrict__ rowptr, uint32_t *__restrict__ out_col,
                                                      T *__restrict__ out_val, uint2 *__restrict__ gwin) {
    constexpr int K = CAP / 256;  // rounds per wave = sorted positions per thread
    // LDS: the sorted values (written only after every rank is known) share their space with the per-wave row counters
    // and the row starts of the counting sort, which are dead by then -- 21 instead of 26 KB at CAP = 1536 (f64): seven
    // workgroups per CU instead of six.  The survivors' row counters of step 4 live there as well.
    constexpr size_t kCntBytes = 4 * 256 * sizeof(uint32_t), kRsBytes = 260 * sizeof(uint32_t);
    constexpr size_t kRegion = CAP * sizeof(T) > kCntBytes + kRsBytes ? CAP * sizeof(T) : kCntBytes + kRsBytes;
    __shared__ __attribute__((aligned(16))) unsigned char s_region[kRegion];
    T *s_v2 = reinterpret_cast<T *>(s_region);
    // lanes of a wave hand counts to each other through this array between two rounds (lds_peek / lds_poke)
    uint32_t (*s_cnt)[256] = reinterpret_cast<uint32_t (*)[256]>(s_region);
    uint32_t *s_rs = reinterpret_cast<uint32_t *>(s_region + kCntBytes);   // 257 row starts
    __shared__ uint32_t s_c1[CAP];
    uint32_t *s_c2 = s_c1;   // (row, col) order replaces the row order in place (a barrier in between)
    uint32_t *s_rk = reinterpret_cast<uint32_t *>(s_region);   // survivors per row: counted when the sorted values are dead too
    __shared__ uint32_t s_wsum[4], s_wlong[4];
    __shared__ uint32_t s_wc[K * 4];
    __shared__ uint32_t s_cmin, s_cmax;   // columns of the survivors (the CSR planner's window input)
    __shared__ uint32_t s_base, s_total;  // survivors of the groups before this one / of this one
    __shared__ uint8_t s_r2[CAP];
    __shared__ uint32_t s_nlong;          // rows of more than 16 entries, listed for the wave-per-row pass of step 2
    __shared__ uint8_t s_long[256];

    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
#if SPAL_COO_PRIO
    // everything up to the group's published count is what OTHER workgroups wait for in their look-back: those phases run at a
    // raised wave priority, the stores behind the look-back at the normal one
    __builtin_amdgcn_s_setprio(SPAL_COO_PRIO);
#endif
    uint32_t row_len = 0, row_a = 0;      // thread t's row of the group: entries, first place in row order
#ifdef SPAL_COO_STAMPS
    const uint32_t stamp_slot = blockIdx.x;
#endif
    SPAL_STAMP(0);
    // The group this workgroup takes: its ticket (start order inside its class), not its blockIdx (see above).
    if (t == 0) {
        uint32_t id = blockIdx.x;
        if (tickets) {
            if (ticket_classes > 1) {
                const uint32_t cls = blockIdx.x & (kTicketClasses - 1);
                id = atomicInc(&tickets[cls * kTicketStride], 0xffffffffu) * kTicketClasses + cls;
            } else {
                id = atomicInc(tickets, 0xffffffffu);
            }
        }
        s_base = id;
    }
    __syncthreads();
    const uint32_t grp = s_base;                          // < ngroups (ngroups workgroups; every class has as many workgroups as ids)
    SPAL_STAMP(1);
    const uint32_t e0 = gstart[grp], e1 = gstart[grp + 1];
    __syncthreads();                                      // (s_base is written again below)
    SPAL_STAMP(2);
    const uint32_t r0 = grp << gbits;                     // < nrows (there are ceil(nrows / 2^gbits) groups)
    const uint32_t nr = min(1u << gbits, nrows - r0);     // rows of this group, <= 256
    uint32_t n = e1 - e0;
    const bool last = grp + 1 == ngroups;
    // The capacity is the host's guess (the last assembly's fullest group, or mean + 6 sigma): a group that does not
    // fit raises *err bit 1, takes part in the look-back as an empty group (nobody waits for it) and the host runs
    // the kernel again at the capacity the fullest group needs -- the device computes that beside (group_offsets / groups_check).
    if (n > (uint32_t)CAP) {
        if (t == 0) atomicOr(err, 2u);
        n = 0;
    }
    if (n == 0) {  // block-uniform: no entries, but the group's rows start where the groups before it end
        if (w == 0) {
            const uint32_t before = group_lookback(state, grp, 0u, lane, err, spin_bound);
            if (lane == 0) s_base = before;
        }
        __syncthreads();
        const uint32_t before = s_base;
        if (t < nr) rowptr[r0 + t] = before;
        if (last) {
            if (t == 0) rowptr[nrows] = before;
            out_col[before + t] = 0u;                     // the stream kernel's over-read margin (256 entries)
            out_val[before + t] = T(0);
        }
        if (t == 0) gwin[grp] = make_uint2(0xffffffffu, 0u);
        return;
    }
    // 0. one batch of loads (clamped lanes re-read the last entry: every load is issued unconditionally, back to
    // back); wave w owns the entries [w * chunk, (w + 1) * chunk)
    const uint32_t chunk = ((n + 255) / 256) * 64;        // entries per wave, a multiple of 64, <= 64 K
    // (the VALUES are requested later, behind step 2: nothing before the scatter into (row, col) order looks at them, and
    //  their registers -- 12 of 72 for f64 -- are what the row-sorting network of step 2 needs; the kernel is bound by its
    //  VALU instructions, so the other workgroups of the CU cover the wait)
    uint32_t rc[K], pr[K];   // column; (row inside the group) << 16 | position (step 1: among the row's entries, then in the group)
    // (lanes beyond the group's last entry read the next group's entries, or up to 255 entries past the end of the sorted
    //  arrays, which lie inside the workspace -- never looked at: one base address and immediate offsets instead of a clamp
    //  and an address per load.  Price: 3 KB per group that its neighbour fetches again, 0.12 of the 4.85 GB per assembly at
    //  config 5; clamped, the network form spills 37 - 66 registers at seven workgroups per CU.)
    const size_t my0 = (size_t)e0 + w * chunk + lane;
    {
        const uint32_t cmask = gbits ? (0xffffffffu >> gbits) : 0xffffffffu, rshift = 32u - gbits;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (PACKED) {
                const uint32_t q = cols[my0 + 64u * k];
                rc[k] = q & cmask;
                pr[k] = (gbits ? (q >> rshift) : 0u) << 16;
            } else {
                rc[k] = cols[my0 + 64u * k];
                pr[k] = (sorted_row[my0 + 64u * k] - r0) << 16;
            }
        }
    }
    {
        for (uint32_t i = t; i < 4 * 256; i += 256) s_cnt[i >> 8][i & 255] = 0;
        if (t == 0) { s_cmin = 0xffffffffu; s_cmax = 0u; s_nlong = 0u; }
        __syncthreads();
        // 1. stable counting sort by row inside the group
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (64u * k >= chunk) break;  // block-uniform
            const bool ok = w * chunk + 64u * k + lane < n;
            const uint32_t d = pr[k] >> 16;
            uint64_t peers = __ballot(ok);   // lanes of this round with the same row
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint64_t m = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
            const uint32_t before = ok ? lds_peek(&s_cnt[w][d]) : 0u;
            pr[k] |= before + (uint32_t)__popcll(peers & lt);
            // the lowest peer lane publishes the new count (one writer per row)
            if (ok && (peers & lt) == 0) lds_poke(&s_cnt[w][d], before + (uint32_t)__popcll(peers));
        }
        __syncthreads();
        SPAL_STAMP(3);
        {   // thread d: exclusive prefix of row d's counts over the waves, then the row starts
            uint32_t run = 0;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const uint32_t c = s_cnt[ww][t];
                s_cnt[ww][t] = run;
                run += c;
            }
            const uint32_t inc = wave_inclusive_scan(run);
            uint32_t longest = run;                             // the longest row of this wave's 64 rows
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, o, 64));
            if (lane == 63) { s_wsum[w] = inc; s_wlong[w] = longest; }
            __syncthreads();
            uint32_t base = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i)
                if (i < w) base += s_wsum[i];
            s_rs[t] = base + inc - run;
            if (t == 255) s_rs[256] = base + inc;  // = n
            row_len = run;
            row_a = base + inc - run;
        }
        __syncthreads();
        SPAL_STAMP(12);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (64u * k >= chunk) break;
            if (w * chunk + 64u * k + lane < n) {
                pr[k] += s_rs[pr[k] >> 16] + s_cnt[w][pr[k] >> 16];   // place in row order, insertion order inside the row
                s_c1[pr[k] & 0xffffu] = rc[k];
            }
        }
        __syncthreads();
        SPAL_STAMP(13);
        // 2. rank by column inside the row -> (row, col) order.  The kernel is bound by its VALU instructions (8 400 wave
        // instructions per group of 1 280 entries, a wave instruction holds a SIMD for four cycles: profiles/r04/
        // coo_assembly.txt), and the entry-parallel count -- every entry walks its row -- was a third of them.  Now:
        //  * a row of at most kRowNet = 16 entries is sorted by ONE thread in registers (thread t: row t; 80 compare-exchanges
        //    of two instructions) which leaves every entry's place in the entry's slot of c1;
        //  * longer rows (up to 256 entries) are listed and taken by a whole wave each, a lane per entry (four at most), the row
        //    read as broadcasts -- a wave that met one such entry used to walk the loop for all its lanes;
        //  * the entries pick their places up.
        // Rows beyond 256 entries, or columns that leave no 5 bits free: the kernel's other form (ROWSORT = false).
        constexpr uint32_t kRowNet = 16;
        if (ROWSORT) {
            const uint32_t group_longest = max(max(s_wlong[0], s_wlong[1]), max(s_wlong[2], s_wlong[3]));
            if (group_longest > 256u && t == 0) atomicOr(err, 4u);   // (the result is discarded: the host takes the other kernel)
            if (row_len > kRowNet) s_long[atomicAdd(&s_nlong, 1u)] = (uint8_t)t;     // (order of the list does not matter)
            else if (s_wlong[w] > 1u) sort_row_in_regs<16>(s_c1, row_a, row_len);    // (wave-uniform: some row of this wave holds two or more)
            else if (row_len) s_c1[row_a] = row_a;
            __syncthreads();
            for (uint32_t li = w; li < s_nlong; li += 4) {   // wave-uniform
                const uint32_t d = s_long[li], a = s_rs[d], L = min(s_rs[d + 1] - a, 256u);
                uint32_t ci[4], rank[4] = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < 4; ++c) ci[c] = s_c1[a + min(lane + 64u * c, L - 1)];
                for (uint32_t j = 0; j < L; ++j) {
                    const uint32_t q = s_c1[a + j];               // one address for the wave: a broadcast
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (64u * c < L) rank[c] += (uint32_t)((q < ci[c]) | ((q == ci[c]) & (j < lane + 64u * c)));   // (wave-uniform test)
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)                       // (after the wave's last read of the row: LDS keeps a wave's order)
                    if (lane + 64u * c < L) s_c1[a + lane + 64u * c] = a + rank[c];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (64u * k >= chunk) break;
                if (w * chunk + 64u * k + lane < n) pr[k] = (pr[k] & 0xffff0000u) | s_c1[pr[k] & 0xffffu];
            }
        } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (64u * k >= chunk) break;
            if (w * chunk + 64u * k + lane < n) {
                const uint32_t d = pr[k] >> 16, a = s_rs[d], b = s_rs[d + 1], ci = rc[k], i = pr[k] & 0xffffu;
                uint32_t rank = 0, j = a;
                for (; j + 4 <= b; j += 4) {
                    const uint32_t q0 = s_c1[j], q1 = s_c1[j + 1], q2 = s_c1[j + 2], q3 = s_c1[j + 3];
                    rank += (uint32_t)((q0 < ci) | ((q0 == ci) & (j < i)));
                    rank += (uint32_t)((q1 < ci) | ((q1 == ci) & (j + 1 < i)));
                    rank += (uint32_t)((q2 < ci) | ((q2 == ci) & (j + 2 < i)));
                    rank += (uint32_t)((q3 < ci) | ((q3 == ci) & (j + 3 < i)));
                }
                for (; j < b; ++j) {
                    const uint32_t q = s_c1[j];
                    rank += (uint32_t)((q < ci) | ((q == ci) & (j < i)));
                }
                pr[k] = (pr[k] & 0xffff0000u) | (a + rank);
            }
        }
        }
        T rv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) rv[k] = vals[my0 + 64u * k];
        __syncthreads();   // every rank is known (and picked up): the row-ordered columns may be overwritten
        SPAL_STAMP(14);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (64u * k >= chunk) break;
            if (w * chunk + 64u * k + lane < n) {
                s_c2[pr[k] & 0xffffu] = rc[k];
                s_v2[pr[k] & 0xffffu] = rv[k];
                s_r2[pr[k] & 0xffffu] = (uint8_t)(pr[k] >> 16);
            }
        }
        __syncthreads();
        SPAL_STAMP(15);
        const uint32_t cur_n = n;
        // 3. heads and run sums; thread t takes the sorted positions t, t + 256, ...
        T acc[K];
        uint32_t kinfo[K];   // bit 31: survivor; low bits: survivors of the same wave round before it
#pragma unroll
        for (int k = 0; k < K; ++k) {
            kinfo[k] = 0;
            acc[k] = T(0);
            if (256u * k >= cur_n) continue;  // block-uniform
            const uint32_t p = 256u * k + t;
            const bool live = p < cur_n;
            const uint32_t pc = live ? p : cur_n - 1, pp = pc ? pc - 1 : 0, pn = min(pc + 1, cur_n - 1);
            const uint32_t cp = s_c2[pc], cprev = s_c2[pp], cnext = s_c2[pn];
            const uint32_t rp = s_r2[pc], rprev = s_r2[pp], rnext = s_r2[pn];
            T a = s_v2[pc];
            const bool head = live && (pc == 0 || cprev != cp || rprev != rp);
            const bool dup = head && pn != pc && cnext == cp && rnext == rp;
            if (__any(dup)) {  // duplicates are rare: most waves skip this
                if (dup) {
                    for (uint32_t q = pc + 1; q < cur_n && s_c2[q] == cp && s_r2[q] == rp; ++q) a = a + s_v2[q];
                }
            }
            const bool keep = head && a != T(0);
            acc[k] = a;
            const uint64_t km = __ballot(keep);
            if (keep) kinfo[k] = 0x80000000u | (uint32_t)__popcll(km & lt);
            if (lane == 0) s_wc[k * 4 + w] = (uint32_t)__popcll(km);
        }
        __syncthreads();
        SPAL_STAMP(4);
        // 4. numbering in sorted order = (round, wave, lane); the group's place in the result (look-back over the
        // groups before it); survivors written at their FINAL offsets; rowptr of the group's rows
        s_rk[t] = 0;   // (in the sorted values' space: the run sums above were their last readers)
        if (t < 64) {  // K * 4 <= 64 wave counts: one wave scans them
            const uint32_t kk = min(t, (uint32_t)(K * 4 - 1));
            const uint32_t c = (t < (uint32_t)(K * 4) && 256u * (kk >> 2) < cur_n) ? s_wc[kk] : 0u;
            const uint32_t inc = wave_inclusive_scan(c);
            if (t < (uint32_t)(K * 4)) s_wc[t] = inc - c;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
#ifdef SPAL_COO_STAMPS
            uint32_t dbg[2] = {0, 0};
            const uint32_t before = group_lookback(state, grp, total, lane, err, spin_bound, dbg);
            if (t == 0 && g_coo_stamps) {
                g_coo_stamps[(size_t)stamp_slot * 16 + 8] = grp;
                g_coo_stamps[(size_t)stamp_slot * 16 + 9] = dbg[0];
                g_coo_stamps[(size_t)stamp_slot * 16 + 10] = dbg[1];
            }
#else
            const uint32_t before = group_lookback(state, grp, total, lane, err, spin_bound);
#endif
            if (t == 0) { s_base = before; s_total = total; }
        }
#if SPAL_COO_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __syncthreads();
        SPAL_STAMP(5);
        const uint32_t before = s_base;
        uint32_t cmin = 0xffffffffu, cmax = 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (256u * k >= cur_n) continue;
            if (kinfo[k] >> 31) {
                const uint32_t p = 256u * k + t;
                const uint32_t o = before + s_wc[k * 4 + w] + (kinfo[k] & 0x7fffffffu);
                const uint32_t cp = s_c2[p];
                out_col[o] = cp;
                out_val[o] = acc[k];
                atomicAdd(&s_rk[s_r2[p]], 1u);
                cmin = min(cmin, cp);
                cmax = max(cmax, cp + 1u);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cmin = min(cmin, (uint32_t)__shfl_xor((int)cmin, o, 64));
            cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, o, 64));
        }
        if (lane == 0) { atomicMin(&s_cmin, cmin); atomicMax(&s_cmax, cmax); }
        __syncthreads();
        if (t == 0) gwin[grp] = make_uint2(s_cmin, s_cmax);
        {   // rowptr[r0 + i] = survivors before the group + those of its rows before row i
            const uint32_t c = s_rk[t];                       // (0 beyond the group's rows)
            const uint32_t inc = wave_inclusive_scan(c);
            if (lane == 63) s_wsum[w] = inc;
            __syncthreads();
            uint32_t pre = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i)
                if (i < w) pre += s_wsum[i];
            if (t < nr) rowptr[r0 + t] = before + pre + inc - c;
        }
        if (last) {
            const uint32_t nnz = before + s_total;
            if (t == 0) rowptr[nrows] = nnz;
            out_col[nnz + t] = 0u;                            // the stream kernel's over-read margin (256 entries)
            out_val[nnz + t] = T(0);
        }
        SPAL_STAMP(6);
#ifdef SPAL_COO_STAMPS
        __builtin_amdgcn_s_waitcnt(0);   // (the stores have drained)
        SPAL_STAMP(7);
#endif
    }
}

// ---- general route (any row length): entries fully sorted by (row, col) ------
template <typename T>
__global__ __launch_bounds__(256) void coo_run_sums(const uint32_t *__restrict__ row,
                                                    const uint32_t *__restrict__ col,
                                                    const T *__restrict__ vals, uint64_t len,
                                                    T *__restrict__ runsum,
                                                    uint32_t *__restrict__ keep) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    const uint32_t r = row[i], c = col[i];
    uint32_t flag = 0;
    if (i == 0 || row[i - 1] != r || col[i - 1] != c) {
        // coo.rs:42-46: colval[prev] += val, one entry after the other
        T acc = vals[i];
        for (uint64_t j = i + 1; j < len && row[j] == r && col[j] == c; ++j) acc = acc + vals[j];
        runsum[i] = acc;
        flag = (acc != T(0)) ? 1u : 0u;  // coo.rs:64  `colval[ptr] != T::zero()`
    }
    keep[i] = flag;
}

template <typename T>
__global__ __launch_bounds__(256) void coo_compact(const uint32_t *__restrict__ row,
                                                   const uint32_t *__restrict__ col,
                                                   const T *__restrict__ runsum,
                                                   const uint32_t *__restrict__ keep,
                                                   const uint32_t *__restrict__ pos, uint64_t len,
                                                   uint32_t *__restrict__ out_row,
                                                   uint32_t *__restrict__ out_col,
                                                   T *__restrict__ out_val) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len || !keep[i]) return;
    const uint32_t q = pos[i];
    out_row[q] = row[i];
    out_col[q] = col[i];
    out_val[q] = runsum[i];
}

// a typed view into the workspace (same accessors as DevBuf, owns nothing)
struct DevView {
    char *p;
    template <typename U> U *as() { return reinterpret_cast<U *>(p); }
};

// One allocation for everything the assembly needs besides its output, made
// when the COO matrix is uploaded (setup, not the timed path).
struct CooWorkspace {
    size_t bytes = 0;
    size_t off_key[2], off_aux[2], off_val[2], off_raw[2], off_gt[2], off_dt[2], off_sums, off_state, off_total, off_gstart;
};

#ifndef SPAL_COO_GROUP_TARGET
#define SPAL_COO_GROUP_TARGET 1400
#endif
// rows of a group the local sort finishes in LDS: about a thousand entries on average
static uint32_t coo_group_bits(uint64_t len, uint64_t n_major) {
    const double mean = (double)len / (double)n_major;
    const uint32_t rbits = bits_for(n_major);
    uint32_t gbits = 8;
    while (gbits > 0 && mean * (double)(1u << gbits) > (double)SPAL_COO_GROUP_TARGET) --gbits;
    if (gbits >= rbits) gbits = rbits - 1;  // at least one pass: it also brings the triplets into the workspace
    return gbits;
}
static uint32_t coo_group_count(uint64_t len, uint64_t n_major) {
    const uint32_t gbits = coo_group_bits(len, n_major);
    return (uint32_t)((n_major + (1ull << gbits) - 1) >> gbits);
}

static CooWorkspace coo_workspace_layout(uint64_t len, uint64_t nrows, size_t elem) {
    CooWorkspace w;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 255) & ~(size_t)255; return r; };
    const uint64_t scan_n = std::max<uint64_t>(len, 1);   // (the general route scans one flag per entry)
    const uint64_t ngroups = len ? coo_group_count(len, nrows) : 1;
    for (int i = 0; i < 2; ++i) {
        w.off_key[i] = take(len * 4);
        w.off_aux[i] = take(len * 4);
        w.off_val[i] = take(len * elem);
    }
    for (int i = 0; i < 2; ++i) {   // PassCounts of the two passes
        w.off_raw[i] = take(256ull * std::max<uint32_t>(sort_stride(len), kHistGroup) * 4);
        w.off_gt[i] = take(256ull * std::max<uint32_t>(sort_groups(len), 1) * 4);
        w.off_dt[i] = take(256 * 4);
    }
    w.off_sums = take(((scan_n + kScanTile - 1) / kScanTile) * 4);
    w.off_state = take(ngroups * 8 + kTailWords * 4);   // the look-back words of coo_group_sort, then {error flags, fullest group, -, -, tickets[8]}
    w.off_total = take(4);
    w.off_gstart = take((ngroups + 1) * 4); // first sorted entry of every group
    (void)take(4096);                       // (the group kernel's last lanes read up to 255 entries past the sorted arrays' end)
    w.bytes = o;
    return w;
}

// The assembly on (major, minor): for CSR major = rows, for CSC major = columns
// (`From<&CooMatrix> for CscMatrix`, src/csc/conv/coo.rs:4-115, is the same code
// with the two exchanged).  Produces the compressed arrays; the caller wraps
// them in a handle.
struct Assembled {
    uint32_t *ptr = nullptr, *ind = nullptr;
    void *val = nullptr;
    uint64_t nnz = 0, cap = 0;
    // {first, one past last} minor index of every 256 majors of the result, when the local
    // sort produced it on the way (saves the CSR planner its own pass over the matrix)
    std::vector<uint2> win256;
    // ... or, still on the device, per group of 2^gwin_bits majors (ownership passes to whoever takes the result)
    uint2 *d_gwin = nullptr;
    uint32_t gwin_n = 0, gwin_bits = 0;
};

template <typename T>
static int coo_assemble_t(spal_coo *c, bool by_cols, hipStream_t st, Assembled &res) {
    const uint64_t len = c->len;
    const uint64_t n_major = by_cols ? c->ncols : c->nrows, n_minor = by_cols ? c->nrows : c->ncols;
    const uint32_t *d_major = by_cols ? c->d_cols : c->d_rows;
    const uint32_t *d_minor = by_cols ? c->d_rows : c->d_cols;
    const uint32_t nrows = (uint32_t)n_major;  // "rows" below = the major index
    const uint32_t cbits = bits_for(n_minor), rbits = bits_for(n_major);
    DevBuf rowptr;
    SPAL_HIP_TRY(rowptr.alloc(((size_t)nrows + 1) * 4));
    if (len == 0) {  // no entries at all: an empty CSR matrix
        DevBuf ocol, oval;
        SPAL_HIP_TRY(ocol.alloc(4));
        SPAL_HIP_TRY(oval.alloc(sizeof(T)));
        SPAL_HIP_TRY(hipMemsetAsync(rowptr.p, 0, ((size_t)nrows + 1) * 4, st));
        SPAL_HIP_TRY(hipStreamSynchronize(st));
        res.ptr = (uint32_t *)rowptr.release(); res.ind = (uint32_t *)ocol.release();
        res.val = oval.release(); res.nnz = 0; res.cap = 0;
        return SPAL_OK;
    }

    std::lock_guard<std::mutex> lock(c->mu);  // one assembly at a time per handle (shared workspace)
    const CooWorkspace ws = coo_workspace_layout(len, nrows, sizeof(T));
    if (!c->d_work || c->work_bytes < ws.bytes) {
        if (c->d_work) { (void)dev_free(c->d_work); c->d_work = nullptr; }
        SPAL_HIP_TRY(dev_alloc((void **)&c->d_work, ws.bytes));
        c->work_bytes = ws.bytes;
    }
    char *wb = (char *)c->d_work;
    DevView total{wb + ws.off_total}, sums{wb + ws.off_sums};
    SortBuffers<T> sb;
    for (int i = 0; i < 2; ++i) {
        sb.key[i] = (uint32_t *)(wb + ws.off_key[i]);
        sb.aux[i] = (uint32_t *)(wb + ws.off_aux[i]);
        sb.val[i] = (T *)(wb + ws.off_val[i]);
    }
    sb.counts = PassCounts{(uint32_t *)(wb + ws.off_raw[0]), (uint32_t *)(wb + ws.off_gt[0]), (uint32_t *)(wb + ws.off_dt[0])};
    sb.counts2 = PassCounts{(uint32_t *)(wb + ws.off_raw[1]), (uint32_t *)(wb + ws.off_gt[1]), (uint32_t *)(wb + ws.off_dt[1])};
    sb.sums = sums.as<uint32_t>();

    // The groups of 2^gbits rows (about a thousand entries on average) that are finished in LDS.  EVERYTHING that
    // depends on the triplets is computed here, in the assembly (the reference's `from` counts and scans inside the
    // call too, src/csr/conv/coo.rs:9-22; a `push` would invalidate anything kept from an earlier one): both passes'
    // digit counts, the groups' offsets (from the sorted keys) and the fullest group.  Only a HINT survives on the
    // handle: the fullest group of the last assembly, which picks the LDS capacity of the group kernel without a host
    // round trip in the middle; the kernel checks it (a group that does not fit raises a flag) and the device computes
    // the true maximum beside, so a wrong hint costs a second launch of that kernel, never a wrong result.
    // LDS of the group kernel is 13 B per entry of capacity: the smallest capacity that holds the fullest group
    // (more workgroups per CU); none -> general route
    const double mean = (double)len / (double)nrows;
    const uint32_t gbits = coo_group_bits(len, n_major);
    const uint32_t ngroups = coo_group_count(len, n_major);
    const int o = by_cols ? 1 : 0;
    auto cap_for = [](uint32_t fullest) {
        return fullest <= 512 ? 512 : fullest <= 1024 ? 1024 : fullest <= 1536 ? 1536 : fullest <= (uint32_t)kGroupCap ? kGroupCap : 0;
    };
    uint32_t guess = c->cap_hint[o];
    if (!guess) {   // first assembly: indices spread evenly would give Poisson counts per group -- mean + 6 sigma
        const double gmean = (double)len / (double)ngroups;
        guess = (uint32_t)std::min<double>(gmean + 6.0 * std::sqrt(gmean) + 16.0, (double)kGroupCap);
    }
    int group_cap = cap_for(guess);
    if (const char *e = getenv("SPAL_COO_ROUTE")) if (!strcmp(e, "general")) group_cap = 0;
    uint32_t spin_bound = kLookbackSpins;
    if (const char *e = getenv("SPAL_COO_LOOKBACK_SPINS")) spin_bound = (uint32_t)strtoul(e, nullptr, 10);
    int ticket_mode = 8;   // SPAL_COO_TICKET: 0 = blockIdx, 1 = one counter (round 3), anything else = the 8 class counters
    if (const char *e = getenv("SPAL_COO_TICKET")) ticket_mode = e[0] == '0' ? 0 : (e[0] == '1' && !e[1]) ? 1 : 8;
    c->last_group_rows = 0;
    c->last_group_cap = 0;
    c->last_relaunches = 0;

    uint32_t nnz = 0;
    DevBuf ocol, oval;
    int cur = 0;
    if (group_cap) {
        uint32_t *d_gstart = reinterpret_cast<uint32_t *>(wb + ws.off_gstart);
        // (the groups' column spans stay on the device, in a block of their own that goes with the result: the CSR planner
        //  fetches them when -- if -- a plan is built; round 3 copied 312 KB back inside every assembly)
        DevBuf gwin_buf;
        SPAL_HIP_TRY(gwin_buf.alloc((size_t)ngroups * sizeof(uint2)));
        uint2 *d_gwin = gwin_buf.as<uint2>();
        unsigned long long *d_state = reinterpret_cast<unsigned long long *>(wb + ws.off_state);
        uint32_t *d_err = reinterpret_cast<uint32_t *>(d_state + ngroups);   // {flags, fullest, -, -, tickets[8]}
        SPAL_HIP_TRY(hipMemsetAsync(d_state, 0, (size_t)ngroups * 8 + kTailWords * 4, st));
        // ---- 1. stable sort by the row bits above gbits, (col, value) carried along; the first pass reads the
        // uploaded triplets directly (they stay untouched).  Exactly two passes (config 5: 16 bits): the groups' offsets
        // come out of the passes' scanned counts (group_offsets), and when a column and the row inside its group fit one
        // word the second pass writes that word instead of key + column (radix_scatter<T, true>).
        const uint32_t sort_bits = rbits - gbits;
        const bool two_pass = sort_bits > 8 && sort_bits <= 16 && !getenv("SPAL_COO_NO_OFFSETS");
        const bool packed = two_pass && cbits + gbits <= 32 && !getenv("SPAL_COO_NO_PACK");
        SPAL_HIP_TRY(radix_sort_bits<T>(sb, len, gbits, sort_bits, cur, st, d_major, d_minor, (const T *)c->d_vals,
                                        two_pass, packed ? (int)gbits : -1));
        // ---- 2. the groups' offsets in the sorted triplets
        if (two_pass) {
            hipLaunchKernelGGL(group_offsets, dim3(256), dim3(256), 0, st, sb.counts.dt, sb.counts2.raw, sb.counts2.gt, sb.counts2.dt,
                               sb.key[cur ^ 1], (uint32_t)len, sort_tiles(len), sort_stride(len), sort_groups(len), gbits + 8,
                               ngroups, d_gstart);
        } else {
            launch_row_starts(sb.key[cur], (uint32_t)len, ngroups, d_gstart, st, gbits);   // (one streaming pass over the sorted keys)
        }
        hipLaunchKernelGGL(groups_check, dim3(std::max<uint32_t>(std::min<uint32_t>((ngroups + 255) / 256, 1024u), 1u)), dim3(256), 0, st,
                           d_gstart, ngroups, d_err + 1);   // the fullest group (the kernel's capacity is a guess: see above)
        // ---- 3. per group: rows, columns, run sums, zero drop in LDS; its place in the result by look-back over
        // the groups before it; survivors and rowptr written at their final offsets.  The result arrays are sized
        // for no entry dropped (the count is only known afterwards) and trimmed when a quarter or more is unused.
        uint64_t cap = len + 256;  // + the stream kernel's over-read margin
        SPAL_HIP_TRY(ocol.alloc(cap * 4));
        SPAL_HIP_TRY(oval.alloc(cap * sizeof(T)));
        // what comes back: the last group's state word (survivors of all groups) and {flags, fullest} -- into PINNED host
        // memory kept on the handle (copies into pageable memory cost 0.12 ms of the call)
        const size_t back_bytes = 16;
        if (!c->h_back || c->h_back_bytes < back_bytes) {
            if (c->h_back) { (void)hipHostFree(c->h_back); c->h_back = nullptr; c->h_back_bytes = 0; }
            SPAL_HIP_TRY(hipHostMalloc(&c->h_back, back_bytes, hipHostMallocDefault));
            c->h_back_bytes = back_bytes;
        }
        unsigned long long *tail = reinterpret_cast<unsigned long long *>(c->h_back);
        tail[0] = tail[1] = 0;
        // step 2 of the group kernel by the per-row network (columns << 5 | place-in-row must fit a word; a row beyond 256
        // entries sends the assembly to the kernel's other form: remembered on the handle like the capacity)
        bool row_sort = cbits <= 27 && !c->loop_hint[o] && !getenv("SPAL_COO_LOOP_RANKS");
        for (int attempt = 0; attempt < 3 && group_cap; ++attempt) {
            typedef void (*group_kernel_t)(const uint32_t *, const uint32_t *, const uint32_t *, const T *, uint32_t, uint32_t,
                                           uint32_t, unsigned long long *, uint32_t *, uint32_t *, uint32_t, uint32_t,
                                           uint32_t *, uint32_t *, T *, uint2 *);
            group_kernel_t k_sort;
#define SPAL_GROUP_KERNEL(P, R) (group_cap == 512 ? coo_group_sort<T, 512, P, R> : group_cap == 1024 ? coo_group_sort<T, 1024, P, R> \
                                 : group_cap == 1536 ? coo_group_sort<T, 1536, P, R> : coo_group_sort<T, kGroupCap, P, R>)
            if (packed) k_sort = row_sort ? SPAL_GROUP_KERNEL(true, true) : SPAL_GROUP_KERNEL(true, false);
            else k_sort = row_sort ? SPAL_GROUP_KERNEL(false, true) : SPAL_GROUP_KERNEL(false, false);
#undef SPAL_GROUP_KERNEL
            // ids: 8 class counters (default), the single counter of round 3 (SPAL_COO_TICKET=1) or blockIdx (=0)
            uint32_t *d_tickets = ticket_mode == 0 ? nullptr : ticket_mode == 1 ? d_err + 2 : d_err + 4;
#ifdef SPAL_COO_STAMPS
            static unsigned long long *d_stamps = nullptr;   // (lab builds: one buffer per process, never freed)
            static uint32_t stamps_for = 0;
            if (stamps_for < ngroups) {
                if (d_stamps) (void)hipFree(d_stamps);
                SPAL_HIP_TRY(hipMalloc((void **)&d_stamps, (size_t)ngroups * 16 * 8));
                stamps_for = ngroups;
                SPAL_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_coo_stamps), &d_stamps, sizeof(d_stamps)));
            }
            SPAL_HIP_TRY(hipMemsetAsync(d_stamps, 0, (size_t)ngroups * 16 * 8, st));
#endif
            hipLaunchKernelGGL(k_sort, dim3(ngroups), dim3(256), 0, st, d_gstart, sb.key[cur], sb.aux[cur],
                               sb.val[cur], nrows, gbits, ngroups, d_state, d_err, d_tickets, ticket_mode == 1 ? 1u : kTicketClasses,
                               spin_bound, rowptr.as<uint32_t>(), ocol.as<uint32_t>(), oval.as<T>(), d_gwin);
            SPAL_HIP_TRY(hipGetLastError());
            SPAL_HIP_TRY(hipMemcpyAsync(tail, d_state + (ngroups - 1), 16, hipMemcpyDeviceToHost, st));
            SPAL_HIP_TRY(hipStreamSynchronize(st));
#ifdef SPAL_COO_STAMPS
            {
                std::vector<unsigned long long> hs((size_t)ngroups * 16);
                SPAL_HIP_TRY(hipMemcpy(hs.data(), d_stamps, hs.size() * 8, hipMemcpyDeviceToHost));
                // order of the stamps in time
                static const int order[12] = {0, 1, 2, 3, 12, 13, 14, 15, 4, 5, 6, 7};
                static const char *name[12] = {"", "ticket", "bounds", "loads+rowsort", "row starts", "placed in row order", "ranks", "sorted arrays",
                                               "heads+sums", "look-back", "stores issued", "stores drained"};
                double sum[12] = {0};
                unsigned long long first = ~0ull, lastt = 0, cnt = 0;
                for (uint32_t g = 0; g < ngroups; ++g) {
                    const unsigned long long *q = &hs[(size_t)g * 16];
                    if (!q[7]) continue;
                    for (int i = 1; i < 12; ++i) sum[i] += (double)(q[order[i]] - q[order[i - 1]]);
                    first = std::min(first, q[0]); lastt = std::max(lastt, q[7]); ++cnt;
                }
                if (cnt) {
                    fprintf(stderr, "[spal coo stamps] %llu workgroups, kernel %.1f us; mean us per phase:", cnt, (double)(lastt - first) / 100.0);
                    double tot = 0;
                    for (int i = 1; i < 12; ++i) { fprintf(stderr, " %s %.2f,", name[i], sum[i] / cnt / 100.0); tot += sum[i] / cnt / 100.0; }
                    fprintf(stderr, " residence %.2f (= %.0f workgroups in flight on average)\n", tot, tot * (double)cnt / ((double)(lastt - first) / 100.0));
                    // who waits for whom: time from start to the published count and the wait behind it, by percentile; the wait
                    // a group cannot avoid is the time until the LAST of its predecessors (by id) has published its count
                    std::vector<double> pub, wait, spins, wins;
                    std::vector<unsigned long long> pub_at(ngroups, 0), ready_at(ngroups, 0);
                    for (uint32_t g = 0; g < ngroups; ++g) {
                        const unsigned long long *q = &hs[(size_t)g * 16];
                        if (!q[7] || q[8] >= ngroups) continue;
                        pub.push_back((double)(q[4] - q[0]) / 100.0);
                        wait.push_back((double)(q[5] - q[4]) / 100.0);
                        spins.push_back((double)q[9]);
                        wins.push_back((double)q[10]);
                        pub_at[q[8]] = q[4];
                        ready_at[q[8]] = q[5];
                    }
                    double natural = 0, measured = 0;
                    unsigned long long latest = 0;
                    for (uint32_t id = 0; id < ngroups; ++id) {
                        if (!pub_at[id]) continue;
                        if (latest > pub_at[id]) natural += (double)(latest - pub_at[id]) / 100.0;
                        measured += (double)(ready_at[id] - pub_at[id]) / 100.0;
                        latest = std::max(latest, pub_at[id]);
                    }
                    {   // which phase the slow groups are slow in: per phase p50 / p99, and the phases' means over the slowest 1 % to publish
                        std::vector<std::vector<double>> ph(12);
                        std::vector<std::pair<double, uint32_t>> by_pub;
                        for (uint32_t g = 0; g < ngroups; ++g) {
                            const unsigned long long *q = &hs[(size_t)g * 16];
                            if (!q[7]) continue;
                            for (int i = 1; i < 12; ++i) ph[i].push_back((double)(q[order[i]] - q[order[i - 1]]) / 100.0);
                            by_pub.push_back({(double)(q[4] - q[0]) / 100.0, g});
                        }
                        fprintf(stderr, "[spal coo stamps] p50 / p99 per phase:");
                        for (int i = 1; i < 12; ++i) {
                            std::sort(ph[i].begin(), ph[i].end());
                            fprintf(stderr, " %s %.1f / %.1f,", name[i], ph[i][ph[i].size() / 2], ph[i][(size_t)(0.99 * (ph[i].size() - 1))]);
                        }
                        std::sort(by_pub.begin(), by_pub.end());
                        const size_t n1 = std::max<size_t>(by_pub.size() / 100, 1);
                        double slow[12] = {0};
                        unsigned long long t_lo = ~0ull, t_hi = 0;
                        for (size_t k = by_pub.size() - n1; k < by_pub.size(); ++k) {
                            const unsigned long long *q = &hs[(size_t)by_pub[k].second * 16];
                            for (int i = 1; i < 12; ++i) slow[i] += (double)(q[order[i]] - q[order[i - 1]]) / 100.0 / (double)n1;
                            t_lo = std::min(t_lo, q[0]); t_hi = std::max(t_hi, q[0]);
                        }
                        fprintf(stderr, "\n[spal coo stamps] the slowest 1 %% to publish (started between %.1f and %.1f us of the kernel), mean us per phase:",
                                (double)(t_lo - first) / 100.0, (double)(t_hi - first) / 100.0);
                        for (int i = 1; i < 12; ++i) fprintf(stderr, " %s %.1f,", name[i], slow[i]);
                        // start times of the slowest 1 % by decile of the kernel
                        int dec[10] = {0};
                        for (size_t k = by_pub.size() - n1; k < by_pub.size(); ++k) {
                            const unsigned long long *q = &hs[(size_t)by_pub[k].second * 16];
                            dec[std::min<int>(9, (int)(10.0 * (double)(q[0] - first) / (double)(lastt - first)))]++;
                        }
                        fprintf(stderr, "\n[spal coo stamps] their starts by tenth of the kernel:");
                        for (int i = 0; i < 10; ++i) fprintf(stderr, " %d", dec[i]);
                        fprintf(stderr, "\n");
                    }
                    auto pct = [](std::vector<double> &v, double p) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[(size_t)(p * (v.size() - 1))]; };
                    fprintf(stderr, "[spal coo stamps] start -> count published us: p50 %.1f p90 %.1f p99 %.1f max %.1f; look-back us: p50 %.1f p90 %.1f p99 %.1f max %.1f; "
                            "polls that waited: mean %.1f p99 %.0f; windows walked: mean %.2f p99 %.0f max %.0f; mean wait %.2f us of which until the last predecessor had published %.2f us\n",
                            pct(pub, 0.5), pct(pub, 0.9), pct(pub, 0.99), pct(pub, 1.0), pct(wait, 0.5), pct(wait, 0.9), pct(wait, 0.99), pct(wait, 1.0),
                            std::accumulate(spins.begin(), spins.end(), 0.0) / std::max<size_t>(spins.size(), 1), pct(spins, 0.99),
                            std::accumulate(wins.begin(), wins.end(), 0.0) / std::max<size_t>(wins.size(), 1), pct(wins, 0.99), pct(wins, 1.0),
                            measured / cnt, natural / cnt);
                }
            }
#endif
            const uint32_t flags = (uint32_t)tail[1], fullest = (uint32_t)(tail[1] >> 32);
            c->cap_hint[o] = std::max<uint32_t>(fullest, 1u);
            c->last_ticket = ticket_mode;
            c->last_packed = packed ? 1 : 0;
            c->last_offsets = two_pass ? 1 : 0;
            if (getenv("SPAL_COO_DEBUG"))
                fprintf(stderr, "[spal coo] %.2f entries/row -> groups of %u rows, guessed %u, fullest %u, capacity %d, flags %u, ticket mode %d, %s, %s\n",
                        mean, 1u << gbits, guess, fullest, group_cap, flags, ticket_mode, packed ? "packed" : "key + column",
                        two_pass ? "offsets from the counts" : "offsets from the sorted keys");
            c->last_row_sort = row_sort ? 1 : 0;
            if (!(flags & 6u)) break;              // every group fitted, no row too long for the kernel's form
            // the guess was too small: once more at the capacity the fullest group needs (the sorted triplets and
            // the groups' offsets stand), or the general route when no capacity holds it; a row beyond the network
            // form's reach: once more with the other form
            group_cap = (flags & 1u) ? 0 : (flags & 2u) ? cap_for(fullest) : group_cap;
            if (flags & 4u) { row_sort = false; c->loop_hint[o] = 1; }
            c->last_relaunches++;
            if (group_cap) {   // states, flags, tickets (the fullest group stands: it is a property of the sorted triplets)
                SPAL_HIP_TRY(hipMemsetAsync(d_state, 0, (size_t)ngroups * 8 + 4, st));
                SPAL_HIP_TRY(hipMemsetAsync(d_err + 2, 0, (kTailWords - 2) * 4, st));
            }
        }
        if (group_cap && (uint32_t)tail[1] == 0 && (tail[0] >> 32) == 2) {   // no flag raised, the last group knows its inclusive count
            nnz = (uint32_t)tail[0];
            c->last_group_rows = (int)(1u << gbits);
            c->last_group_cap = group_cap;
            res.d_gwin = (uint2 *)gwin_buf.release(); res.gwin_n = ngroups; res.gwin_bits = gbits;
            if (((uint64_t)nnz + 256) * 4 <= cap * 3) {   // many duplicates summed: do not keep len-sized arrays
                DevBuf tcol, tval;
                const uint64_t tcap = (uint64_t)nnz + 256;
                SPAL_HIP_TRY(tcol.alloc(tcap * 4));
                SPAL_HIP_TRY(tval.alloc(tcap * sizeof(T)));
                SPAL_HIP_TRY(hipMemcpyAsync(tcol.p, ocol.p, tcap * 4, hipMemcpyDeviceToDevice, st));
                SPAL_HIP_TRY(hipMemcpyAsync(tval.p, oval.p, tcap * sizeof(T), hipMemcpyDeviceToDevice, st));
                SPAL_HIP_TRY(hipStreamSynchronize(st));
                std::swap(tcol.p, ocol.p);
                std::swap(tval.p, oval.p);
                cap = tcap;
            }
            res.ptr = (uint32_t *)rowptr.release(); res.ind = (uint32_t *)ocol.release();
            res.val = oval.release(); res.nnz = nnz; res.cap = cap;
            return SPAL_OK;
        }
        // a group beyond every capacity, or the look-back gave up waiting (its backstop: see group_lookback)
        // -> the general route below
        if (getenv("SPAL_COO_DEBUG")) fprintf(stderr, "[spal coo] flags %u: general route\n", (uint32_t)tail[1]);
        c->last_lookback_gave_up += ((uint32_t)tail[1] & 1u) ? 1 : 0;
        (void)dev_free(ocol.release());
        (void)dev_free(oval.release());
    }

    // ---- general route: sort by column bits, then by row bits (LSD), with the
    // column as key first (key <-> aux swapped for the column passes)
    cur = 0;
    SPAL_HIP_TRY(radix_sort_bits<T>(sb, len, 0, cbits, cur, st, d_minor, d_major,
                                    (const T *)c->d_vals));
    std::swap(sb.key[0], sb.aux[0]);  // now key = row, aux = col
    std::swap(sb.key[1], sb.aux[1]);
    SPAL_HIP_TRY(radix_sort_bits<T>(sb, len, 0, rbits, cur, st));
    uint32_t *s_row = sb.key[cur], *s_col = sb.aux[cur];
    T *s_val = sb.val[cur];
    uint32_t *d_keep = sb.key[cur ^ 1], *d_pos = sb.aux[cur ^ 1];  // scratch
    T *runsum = sb.val[cur ^ 1];
    const uint32_t g256 = (uint32_t)((len + 255) / 256);
    hipLaunchKernelGGL(coo_run_sums<T>, dim3(g256), dim3(256), 0, st, s_row, s_col, s_val, len, runsum,
                       d_keep);
    SPAL_HIP_TRY(exclusive_scan_u32(d_keep, d_pos, len, sums.as<uint32_t>(), total.as<uint32_t>(), st));
    SPAL_HIP_TRY(hipMemcpyAsync(&nnz, total.p, 4, hipMemcpyDeviceToHost, st));
    SPAL_HIP_TRY(hipStreamSynchronize(st));
    {
        DevBuf orow;
        const uint64_t cap = (uint64_t)nnz + 256;
        SPAL_HIP_TRY(orow.alloc((size_t)nnz * 4));
        SPAL_HIP_TRY(ocol.alloc(cap * 4));
        SPAL_HIP_TRY(oval.alloc(cap * sizeof(T)));
        SPAL_HIP_TRY(hipMemsetAsync((char *)ocol.p + (size_t)nnz * 4, 0, 256 * 4, st));
        SPAL_HIP_TRY(hipMemsetAsync((char *)oval.p + (size_t)nnz * sizeof(T), 0, 256 * sizeof(T), st));
        hipLaunchKernelGGL(coo_compact<T>, dim3(g256), dim3(256), 0, st, s_row, s_col, runsum, d_keep,
                           d_pos, len, orow.as<uint32_t>(), ocol.as<uint32_t>(), oval.as<T>());
        launch_row_starts(orow.as<uint32_t>(), nnz, nrows, rowptr.as<uint32_t>(), st);
        SPAL_HIP_TRY(hipGetLastError());
        SPAL_HIP_TRY(hipStreamSynchronize(st));
        res.ptr = (uint32_t *)rowptr.release(); res.ind = (uint32_t *)ocol.release();
        res.val = oval.release(); res.nnz = nnz; res.cap = cap;
    }
    return SPAL_OK;
}

static int coo_assemble(spal_coo *c, bool by_cols, hipStream_t st, Assembled &res) {
    return c->elem_size == 8 ? coo_assemble_t<double>(c, by_cols, st, res)
                             : coo_assemble_t<float>(c, by_cols, st, res);
}

// --------------------------------------------------------------------------
// compressed-by-major -> compressed-by-minor (CSR <-> CSC, transpose)
// Device twin of the counting sort of src/csr.rs:358-406 /
// src/csr/conv/csc.rs:4-52 / src/csc/conv/csr.rs:4-52: a stable sort of the
// entries by their minor index keeps the major indices ascending inside every
// minor slice, so the result is exactly the reference's (same order, same
// values -- entries are only moved).
// --------------------------------------------------------------------------
// major index of every entry (one thread per major slice; slices are short)
__global__ __launch_bounds__(256) void expand_major(const uint32_t *__restrict__ ptr, uint32_t nmajor,
                                                    uint32_t *__restrict__ major) {
    const uint64_t m = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= nmajor) return;
    for (uint32_t p = ptr[m]; p < ptr[m + 1]; ++p) major[p] = (uint32_t)m;
}

template <typename T>
static int transpose_t(int device, uint64_t nmajor, uint64_t nminor, uint64_t nnz,
                       const uint32_t *d_ptr, const uint32_t *d_ind, const T *d_val, hipStream_t st,
                       uint32_t **out_ptr, uint32_t **out_ind, T **out_val, uint64_t *out_cap) {
    (void)device;
    const uint64_t cap = nnz + 256;  // the stream kernel's over-read margin
    DevBuf optr, oind, oval;
    SPAL_HIP_TRY(optr.alloc((nminor + 1) * 4));
    SPAL_HIP_TRY(oind.alloc(cap * 4));
    SPAL_HIP_TRY(oval.alloc(cap * sizeof(T)));
    SPAL_HIP_TRY(hipMemsetAsync((char *)oind.p + nnz * 4, 0, 256 * 4, st));
    SPAL_HIP_TRY(hipMemsetAsync((char *)oval.p + nnz * sizeof(T), 0, 256 * sizeof(T), st));
    if (nnz == 0) {
        SPAL_HIP_TRY(hipMemsetAsync(optr.p, 0, (nminor + 1) * 4, st));
    } else {
        const CooWorkspace ws = coo_workspace_layout(nnz, nminor, sizeof(T));
        DevBuf work, major;
        SPAL_HIP_TRY(work.alloc(ws.bytes));
        SPAL_HIP_TRY(major.alloc(nnz * 4));
        char *wb = (char *)work.p;
        SortBuffers<T> sb;
        for (int i = 0; i < 2; ++i) {
            sb.key[i] = (uint32_t *)(wb + ws.off_key[i]);
            sb.aux[i] = (uint32_t *)(wb + ws.off_aux[i]);
            sb.val[i] = (T *)(wb + ws.off_val[i]);
        }
        sb.counts = PassCounts{(uint32_t *)(wb + ws.off_raw[0]), (uint32_t *)(wb + ws.off_gt[0]), (uint32_t *)(wb + ws.off_dt[0])};
        sb.sums = (uint32_t *)(wb + ws.off_sums);
        hipLaunchKernelGGL(expand_major, dim3((uint32_t)((nmajor + 255) / 256)), dim3(256), 0, st, d_ptr,
                           (uint32_t)nmajor, major.as<uint32_t>());
        int cur = 0;
        SPAL_HIP_TRY(radix_sort_bits<T>(sb, nnz, 0, bits_for(nminor), cur, st, d_ind,
                                        major.as<uint32_t>(), d_val));
        launch_row_starts(sb.key[cur], (uint32_t)nnz, (uint32_t)nminor, optr.as<uint32_t>(), st);
        SPAL_HIP_TRY(hipMemcpyAsync(oind.p, sb.aux[cur], nnz * 4, hipMemcpyDeviceToDevice, st));
        SPAL_HIP_TRY(hipMemcpyAsync(oval.p, sb.val[cur], nnz * sizeof(T), hipMemcpyDeviceToDevice, st));
        SPAL_HIP_TRY(hipGetLastError());
        SPAL_HIP_TRY(hipStreamSynchronize(st));  // `work` is freed on return
    }
    SPAL_HIP_TRY(hipStreamSynchronize(st));
    *out_ptr = (uint32_t *)optr.release();
    *out_ind = (uint32_t *)oind.release();
    *out_val = (T *)oval.release();
    *out_cap = cap;
    return SPAL_OK;
}

int transpose_device(int device, int elem_size, uint64_t nmajor, uint64_t nminor, uint64_t nnz,
                     const uint32_t *d_ptr, const uint32_t *d_ind, const void *d_val, hipStream_t st,
                     uint32_t **out_ptr, uint32_t **out_ind, void **out_val, uint64_t *out_cap) {
    if (elem_size == 8) {
        double *v = nullptr;
        SPAL_TRY(transpose_t<double>(device, nmajor, nminor, nnz, d_ptr, d_ind, (const double *)d_val, st,
                                     out_ptr, out_ind, &v, out_cap));
        *out_val = v;
    } else {
        float *v = nullptr;
        SPAL_TRY(transpose_t<float>(device, nmajor, nminor, nnz, d_ptr, d_ind, (const float *)d_val, st,
                                    out_ptr, out_ind, &v, out_cap));
        *out_val = v;
    }
    return SPAL_OK;
}

static void coo_free(spal_coo *c) {
    if (!c) return;
    if (c->h_back) (void)hipHostFree(c->h_back);
    (void)dev_free(c->d_work);
    (void)dev_free(c->d_rows);
    (void)dev_free(c->d_cols);
    (void)dev_free(c->d_vals);
    delete c;
}

template <typename T>
static int coo_upload(int device, uint64_t nrows, uint64_t ncols, uint64_t len, const uint64_t *rows,
                      const uint64_t *cols, const T *vals, spal_coo_t *out) {
    if (!out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_coo_upload: out is NULL");
    *out = nullptr;
    if (len && (!rows || !cols || !vals))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_coo_upload: null array");
    // CooMatrix::new asserts (src/coo.rs:105-106)
    if (!(nrows > 0)) return fail(SPAL_ERR_INVARIANT, "CooMatrix::new would panic: assertion failed: nrows > 0");
    if (!(ncols > 0)) return fail(SPAL_ERR_INVARIANT, "CooMatrix::new would panic: assertion failed: ncols > 0");
    // A COO handle assembles to CSR (pointer array over nrows + 1) and to CSC (over ncols + 1): whichever becomes
    // the MAJOR dimension needs dim + 1 to fit 32 bits, so both are bounded alike here (spal_csr_create / spal_csc_create
    // know their major dimension and allow the minor one to be exactly 2^32 - 1).
    if (nrows >= 0xffffffffull || ncols >= 0xffffffffull || len > kMaxEntries)
        return fail(SPAL_ERR_UNSUPPORTED, "COO shape does not fit 32-bit device indices (nrows, ncols < 2^32 - 1)");
    // every entry inside the matrix (push asserts, src/coo.rs:432-433)
    std::vector<uint32_t> r32(len), c32(len);
    std::vector<uint64_t> bad(host_threads(), UINT64_MAX);  // first offending entry of every thread's range
    parallel_for(len, [&](uint64_t b, uint64_t e, unsigned t) {
        for (uint64_t i = b; i < e; ++i) {
            if (rows[i] >= nrows || cols[i] >= ncols) { bad[t] = i; return; }
            r32[i] = (uint32_t)rows[i];
            c32[i] = (uint32_t)cols[i];
        }
    });
    uint64_t first_bad = UINT64_MAX;
    for (uint64_t f : bad) first_bad = std::min(first_bad, f);
    if (first_bad != UINT64_MAX) {  // the entry a sequence of push() calls would have panicked on (row is asserted first)
        const bool row_bad = rows[first_bad] >= nrows;
        return fail(SPAL_ERR_INDEX_OUT_OF_BOUNDS,
                    "CooMatrix::push would panic: assertion failed: %s (entry %llu: %s %llu)",
                    row_bad ? "row < nrows" : "col < ncols", (unsigned long long)first_bad, row_bad ? "row" : "col",
                    (unsigned long long)(row_bad ? rows[first_bad] : cols[first_bad]));
    }
    DeviceGuard guard(device);
    if (guard.status != SPAL_OK) return guard.status;
    spal_coo *c = new spal_coo;
    c->device = device; c->elem_size = (int)sizeof(T);
    c->nrows = nrows; c->ncols = ncols; c->len = len;
    hipError_t e = dev_alloc((void **)&c->d_rows, std::max<uint64_t>(len, 1) * 4);
    if (e == hipSuccess) e = dev_alloc((void **)&c->d_cols, std::max<uint64_t>(len, 1) * 4);
    if (e == hipSuccess) e = dev_alloc((void **)&c->d_vals, std::max<uint64_t>(len, 1) * sizeof(T));
    if (e == hipSuccess && len) e = hipMemcpy(c->d_rows, r32.data(), len * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && len) e = hipMemcpy(c->d_cols, c32.data(), len * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && len) e = hipMemcpy(c->d_vals, vals, len * sizeof(T), hipMemcpyHostToDevice);
    if (e == hipSuccess && len) {  // the assembly's workspace: setup, not the timed path
        c->work_bytes = coo_workspace_layout(len, nrows, sizeof(T)).bytes;
        e = dev_alloc((void **)&c->d_work, c->work_bytes);
    }
    if (e != hipSuccess) {
        coo_free(c);
        return fail(e == hipErrorOutOfMemory ? SPAL_ERR_OUT_OF_MEMORY : SPAL_ERR_HIP,
                    "spal_coo_upload: upload failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return SPAL_OK;
}

template <typename T>
static int coo_to_csr(int device, uint64_t nrows, uint64_t ncols, uint64_t len, const uint64_t *rows,
                      const uint64_t *cols, const T *vals, spal_csr_t *out) {
    if (!out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_coo_to_csr: out is NULL");
    *out = nullptr;
    spal_coo_t c = nullptr;
    SPAL_TRY(coo_upload<T>(device, nrows, ncols, len, rows, cols, vals, &c));
    int st = spal_coo_assemble_csr(c, nullptr, out);
    spal_coo_destroy(c);
    return st;
}

}  // namespace spal

using namespace spal;

extern "C" {

int spal_coo_upload_f64(int device, uint64_t nrows, uint64_t ncols, uint64_t len, const uint64_t *rows,
                        const uint64_t *cols, const double *vals, spal_coo_t *out) {
    return coo_upload<double>(device, nrows, ncols, len, rows, cols, vals, out);
}
int spal_coo_upload_f32(int device, uint64_t nrows, uint64_t ncols, uint64_t len, const uint64_t *rows,
                        const uint64_t *cols, const float *vals, spal_coo_t *out) {
    return coo_upload<float>(device, nrows, ncols, len, rows, cols, vals, out);
}
int spal_coo_destroy(spal_coo_t c) {
    if (!c) return SPAL_OK;
    DeviceGuard guard(c->device);
    coo_free(c);
    return SPAL_OK;
}
int spal_coo_describe(spal_coo_t c, char *buf, size_t buf_len) {
    if (!c || !buf || !buf_len) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_coo_describe: null argument");
    snprintf(buf, buf_len,
             "{\"format\": \"coo\", \"dtype\": \"%s\", \"nrows\": %llu, \"ncols\": %llu, \"len\": %llu, "
             "\"last_route\": \"%s\", \"group_rows\": %d, \"group_cap\": %d, \"group_relaunches\": %d, "
             "\"lookback_gave_up\": %d, \"ticket_mode\": %d, \"packed_payload\": %d, \"offsets_from_counts\": %d, \"row_sort\": %d}",
             c->elem_size == 8 ? "f64" : "f32", (unsigned long long)c->nrows, (unsigned long long)c->ncols,
             (unsigned long long)c->len, c->last_group_rows ? "local_sort" : "general", c->last_group_rows,
             c->last_group_cap, c->last_relaunches, c->last_lookback_gave_up, c->last_ticket, c->last_packed, c->last_offsets, c->last_row_sort);
    return SPAL_OK;
}
int spal_coo_assemble_csr(spal_coo_t c, void *stream, spal_csr_t *out) {
    if (!c || !out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_coo_assemble_csr: null argument");
    *out = nullptr;
    DeviceGuard guard(c->device);
    if (guard.status != SPAL_OK) return guard.status;
    Assembled r;
    SPAL_TRY(coo_assemble(c, false, (hipStream_t)stream, r));
    int st = csr_adopt_device(c->device, c->elem_size, c->nrows, c->ncols, r.nnz, r.cap, r.ptr, r.ind,
                              r.val, out, r.win256.empty() ? nullptr : &r.win256, false,
                              !(getenv("SPAL_COO_EAGER_PLAN") && getenv("SPAL_COO_EAGER_PLAN")[0] == '1'),
                              r.d_gwin, r.gwin_n, r.gwin_bits);   // (takes the spans' block, also when it fails)
    if (st != SPAL_OK) { (void)dev_free(r.ptr); (void)dev_free(r.ind); (void)dev_free(r.val); }
    return st;
}
int spal_coo_assemble_csc(spal_coo_t c, void *stream, spal_csc_t *out) {
    if (!c || !out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_coo_assemble_csc: null argument");
    *out = nullptr;
    DeviceGuard guard(c->device);
    if (guard.status != SPAL_OK) return guard.status;
    Assembled r;
    SPAL_TRY(coo_assemble(c, true, (hipStream_t)stream, r));
    (void)dev_free(r.d_gwin);   // (the groups' ROW spans: the CSC planner has no use for them)
    r.d_gwin = nullptr;
    if (r.cap < r.nnz + 256) {  // csc handles expect the over-read margin too
        uint32_t *ind = nullptr;
        void *val = nullptr;
        hipError_t e = dev_alloc((void **)&ind, (r.nnz + 256) * 4);
        if (e == hipSuccess) e = dev_alloc((void **)&val, (r.nnz + 256) * (size_t)c->elem_size);
        if (e == hipSuccess) e = hipMemset(ind, 0, (r.nnz + 256) * 4);
        if (e == hipSuccess) e = hipMemset(val, 0, (r.nnz + 256) * (size_t)c->elem_size);
        if (e == hipSuccess && r.nnz) e = hipMemcpy(ind, r.ind, r.nnz * 4, hipMemcpyDeviceToDevice);
        if (e == hipSuccess && r.nnz) e = hipMemcpy(val, r.val, r.nnz * (size_t)c->elem_size, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            (void)dev_free(ind); (void)dev_free(val); (void)dev_free(r.ptr); (void)dev_free(r.ind); (void)dev_free(r.val);
            return fail(SPAL_ERR_HIP, "spal_coo_assemble_csc: %s", hipGetErrorString(e));
        }
        (void)dev_free(r.ind); (void)dev_free(r.val);
        r.ind = ind; r.val = val;
    }
    int st = csc_adopt_device(c->device, c->elem_size, c->nrows, c->ncols, r.nnz, r.ptr, r.ind, r.val, out);
    if (st != SPAL_OK) { (void)dev_free(r.ptr); (void)dev_free(r.ind); (void)dev_free(r.val); }
    return st;
}
int spal_coo_to_csr_f64(int device, uint64_t nrows, uint64_t ncols, uint64_t len, const uint64_t *rows,
                        const uint64_t *cols, const double *vals, spal_csr_t *out) {
    return coo_to_csr<double>(device, nrows, ncols, len, rows, cols, vals, out);
}
int spal_coo_to_csr_f32(int device, uint64_t nrows, uint64_t ncols, uint64_t len, const uint64_t *rows,
                        const uint64_t *cols, const float *vals, spal_csr_t *out) {
    return coo_to_csr<float>(device, nrows, ncols, len, rows, cols, vals, out);
}
int spal_coo_to_csc_f64(int device, uint64_t nrows, uint64_t ncols, uint64_t len, const uint64_t *rows,
                        const uint64_t *cols, const double *vals, spal_csc_t *out) {
    if (!out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_coo_to_csc: out is NULL");
    *out = nullptr;
    spal_coo_t c = nullptr;
    SPAL_TRY(coo_upload<double>(device, nrows, ncols, len, rows, cols, vals, &c));
    int st = spal_coo_assemble_csc(c, nullptr, out);
    spal_coo_destroy(c);
    return st;
}
int spal_coo_to_csc_f32(int device, uint64_t nrows, uint64_t ncols, uint64_t len, const uint64_t *rows,
                        const uint64_t *cols, const float *vals, spal_csc_t *out) {
    if (!out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_coo_to_csc: out is NULL");
    *out = nullptr;
    spal_coo_t c = nullptr;
    SPAL_TRY(coo_upload<float>(device, nrows, ncols, len, rows, cols, vals, &c));
    int st = spal_coo_assemble_csc(c, nullptr, out);
    spal_coo_destroy(c);
    return st;
}

}  // extern "C"
