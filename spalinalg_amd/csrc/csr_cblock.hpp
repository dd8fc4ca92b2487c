// csr_cblock.hpp -- y = A*x for matrices whose columns are NOT local (uniform-random columns, the assembled
// config-5 matrix, power-law graphs): the column-blocked kernel family.
//
// What bounds the other kernels on such matrices is the gather of x: every 8-byte read pulls a 128-byte line from
// beyond L2 (config 5's result: 6.87 GB fetched for 0.65 GB algorithmic, profiles/r02).  Here the matrix is kept a
// second time in TILES: rows in blocks of RB, columns in blocks of CB (x slice of CB columns = 1 MB: it
// stays in every XCD's 4 MB L2 while the workgroups of that XCD work on it), entries ordered (row block, column
// block, row, column).  One workgroup per row block walks its column blocks in ascending order; per tile
//   * the tile's entries are read entry-parallel (coalesced 8 + 4 + 2 bytes per lane: value, column, row inside the
//     row block), x gathered (L2 hits), the PRODUCT, rounded, goes to an LDS strip in entry order;
//   * the entry that heads its row's run in the tile adds the run's products to the row's running sum ONE AFTER THE
//     OTHER.
// The running sum of a row lives in LDS from the first column block to the last, so a row's products are
// added in ascending column order, the product rounded before the add, the first product taken as it is: exactly
// the reference's order (src/csr/ops/mul.rs:31-38) -- rows are bit-identical to the sequential CPU result.
// (The running sums start at -0.0: -0.0 + p == p bit for bit for every p, which makes "add" the reference's
//  "assign" for the first product; rows without entries store +0.0.)
// Two kernels share the tiled copy: csr_spmv_cblock (entry-parallel, producer and consumer waves: runs of about one
// entry per row and column block) and csr_spmv_cblock_rows (threads own rows: runs of several entries); the plan counts
// the runs and chooses (spal_csr_cblock.hip).
#pragma once
#include "csr_kernels.hpp"

namespace spal {

constexpr int kCbThreads = 256;
constexpr uint32_t kCbStrip = 4096;      // most products of one tile (entries)
constexpr uint32_t kCbMaxRows = 8192;    // most rows of a row block (their running sums: 64 KiB of f64; the plan keeps a workgroup under half a CU's LDS)
constexpr uint32_t kCbMaxBlocks = 128;   // column blocks per matrix

__device__ __forceinline__ uint32_t cb_wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= (uint32_t)o) v += t;
    }
    return v;
}

// ---- the entry-parallel form ------------------------------------------------------------------------------------------
// tile_ptr[rb * nbc + cb] = first entry of tile (rb, cb); rows16[e] = row of entry e inside its row block; a tile holds
// at most S entries (plan).  The running sums of the row block live in LDS (acc), every entry's product goes to a strip;
// an entry whose predecessor in the tile belongs to another row HEADS its row's run and adds the run to acc[row] one
// product after the other -- acc[row], then + first product, then + second ...: the reference's order.  A row has one
// head per tile, so no two threads touch one acc[row] between two barriers.
// The two phases of a tile run on DIFFERENT WAVES: NP waves PRODUCE (entries, gathers of x, products and rows into one
// of TWO strips), the other 4 - NP CONSUME (head-of-run sums) the tile before -- producers work on tile k + 1 while
// consumers sum tile k; one barrier per tile hands a strip over in each direction.  What bounds the kernel is the L1's
// queue of outstanding gathers: with all waves alternating between the phases (round 3's first form) it
// idled while the sums were taken (pending-stall counter 65 % of the cycles; 488 us on the config-5 result), now it is
// fed all the time (383 us).  A producer thread keeps two batches of entries in registers: the batch whose x it gathers
// and the next, requested a batch ahead (unconditional clamped loads; past the last batch it re-reads the last: no
// traffic).  Measured: NP = 2, U = 8 (3 producers and one consumer wave: 586-965 us; U = 4 / 12 / 16: 420 / 544 / 570);
// exactly TWO workgroups per CU (one: 619, three: 631, four: 731 us -- the launch pads its LDS request to stay at two).
// RB (rows of a row block, <= kCbMaxRows) and S (strip entries) are the plan's: the row blocks are sized so that the
// launch is a whole number of rounds of the workgroups the device holds (1536 workgroups = 3.0 rounds: 412 us, 1571 =
// 3.07 rounds: 485 us).
// LDS (dynamic): strip T[2][S] | acc T[RB rounded up to 2] | srow u16[2][S + 2].
template <typename T, int NP, int U>
__global__ __launch_bounds__(kCbThreads) void csr_spmv_cblock(const T *__restrict__ vals, const uint32_t *__restrict__ cols,
                                                                 const uint16_t *__restrict__ rows16,
                                                                 const uint32_t *__restrict__ tile_ptr,
                                                                 const uint32_t *__restrict__ rowptr, const T *__restrict__ x,
                                                                 T *__restrict__ y, uint32_t nrows, uint32_t nbc, uint32_t RB,
                                                                 uint32_t S) {
    constexpr uint32_t PT = 64u * NP, CT = kCbThreads - PT, BATCH = PT * U;   // producer / consumer threads, entries per batch
    extern __shared__ __attribute__((aligned(16))) unsigned char spal_smem[];
    T *strip = reinterpret_cast<T *>(spal_smem);                              // [2][S]
    T *acc = strip + 2u * S;
    uint16_t *srow = reinterpret_cast<uint16_t *>(acc + ((RB + 1u) & ~1u));   // [2][S + 2]: srow[1 + i] = row of entry i; sentinels either side
    __shared__ uint32_t s_tp[kCbMaxBlocks + 1];
    const uint32_t tid = threadIdx.x;
    const uint32_t rb = blockIdx.x;
    for (uint32_t i = tid; i <= nbc; i += kCbThreads) s_tp[i] = tile_ptr[(size_t)rb * nbc + i];
    for (uint32_t i = tid; i < RB; i += kCbThreads) acc[i] = -T(0);
    if (tid < 2) srow[tid * (S + 2u)] = 0xffffu;
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // (scalar: the roles are uniform control flow)
    auto tp = [&](uint32_t i) { return (uint32_t)__builtin_amdgcn_readfirstlane(s_tp[i]); };
    auto next_tile = [&](uint32_t cb) { while (cb < nbc && tp(cb + 1) == tp(cb)) ++cb; return cb; };   // the next tile with entries, or nbc

    if (wave < (uint32_t)NP) {
        // ---- producers
        struct Batch { uint32_t cb = 0, b0 = 0, e0 = 0, cnt = 0, tile_n = 0; };   // cnt == 0: past the last batch
        auto first = [&]() {
            Batch b;
            b.cb = next_tile(0u);
            if (b.cb < nbc) { b.e0 = tp(b.cb); b.tile_n = tp(b.cb + 1) - b.e0; b.cnt = min(BATCH, b.tile_n); }
            return b;
        };
        auto after = [&](const Batch &a) {                                     // the batch after a (a itself when a is past the end)
            Batch b = a;
            if (a.cnt == 0u) return b;
            if (a.b0 + a.cnt < a.tile_n) { b.b0 = a.b0 + a.cnt; b.e0 = a.e0 + a.cnt; b.cnt = min(BATCH, a.tile_n - b.b0); return b; }
            b.cb = next_tile(a.cb + 1u); b.b0 = 0u;
            if (b.cb < nbc) { b.e0 = tp(b.cb); b.tile_n = tp(b.cb + 1) - b.e0; b.cnt = min(BATCH, b.tile_n); }
            else { b.cnt = 0u; }                                               // (e0, tile_n stay: the clamped loads re-read the last batch)
            return b;
        };
        T va[U], vb[U];
        uint32_t ca[U], cb2[U];
        uint16_t ra[U], rb2[U];
        auto request = [&](T (&v)[U], uint32_t (&c)[U], uint16_t (&r)[U], const Batch &b, const Batch &fallback) {
            const uint32_t e0 = b.cnt ? b.e0 : fallback.e0, cnt = b.cnt ? b.cnt : fallback.cnt;   // uniform
            const uint32_t lim = cnt ? cnt - 1u : 0u;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = e0 + min(tid + (uint32_t)u * PT, lim);
                v[u] = load_stream(vals + i);
                c[u] = load_stream(cols + i);
                r[u] = load_stream(rows16 + i);
            }
        };
        uint32_t buf = 0;
        auto produce = [&](T (&v)[U], uint32_t (&c)[U], uint16_t (&r)[U], const Batch &b) {
            T xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = x[c[u]];
            T *st = strip + buf * S;
            uint16_t *sr = srow + buf * (S + 2u);
            T pr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pr[u] = v[u] * xv[u];                  // (outside the lanes' branch: the wait for x is every path's)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t k = tid + (uint32_t)u * PT;
                if (k < b.cnt) { st[b.b0 + k] = pr[u]; sr[1u + b.b0 + k] = r[u]; }
            }
            if (b.cnt && b.b0 + b.cnt == b.tile_n) {                           // uniform: the tile is complete
                if (tid == 0) sr[1u + b.tile_n] = 0xffffu;                     // no run continues past the tile
                __syncthreads();                                               // hand it over; the other strip is free again
                buf ^= 1u;
            }
        };
        Batch cur = first();
        if (cur.cnt) {
            request(va, ca, ra, cur, cur);
            bool more;
            do {
                const Batch nxt = after(cur);
                request(vb, cb2, rb2, nxt, cur);
                produce(va, ca, ra, cur);
                cur = after(nxt);
                Batch fb = cur;                                                // (by value: a reference to either of two structs goes through memory)
                if (nxt.cnt) { fb.e0 = nxt.e0; fb.cnt = nxt.cnt; }
                request(va, ca, ra, cur, fb);
                produce(vb, cb2, rb2, nxt);
                more = cur.cnt != 0u;
            } while (more);
        }
    } else {
        // ---- consumers: tile k's sums while tile k + 1 is produced
        const uint32_t ctid = tid - PT;
        uint32_t buf = 0;
        for (uint32_t cb = next_tile(0u); cb < nbc; cb = next_tile(cb + 1u)) {
            const uint32_t n = tp(cb + 1) - tp(cb);
            __syncthreads();                                                   // the tile's products and rows are in strip[buf]
            const T *st = strip + buf * S;
            const uint16_t *sr = srow + buf * (S + 2u);
            for (uint32_t i = ctid; i < n; i += CT) {
                const uint32_t r = sr[1u + i];
                if (sr[i] != r) {                                              // the row's first entry in this tile: its run, in order
                    T a = acc[r] + st[i];
                    for (uint32_t k = i + 1u; sr[1u + k] == r; ++k) a = a + st[k];
                    acc[r] = a;
                }
            }
            buf ^= 1u;
        }
    }
    __syncthreads();                                                           // every tile is summed
    // y: rows without entries are +0.0 (the running sums started at -0.0: -0.0 + p == p for the first product)
    const uint32_t r0 = rb * RB;
    for (uint32_t i = tid; i < RB; i += kCbThreads) {
        const uint32_t r = r0 + i;
        if (r < nrows) y[r] = rowptr[r + 1] != rowptr[r] ? acc[i] : T(0);
    }
}

// ---- the rows form: rows that hold SEVERAL entries per column block --------------------------------------------------
// (14 per row over 8 column blocks, wide bands.)  The entry-parallel kernel lets the thread that heads a row's run
// add the whole run while the run's other threads idle, and pays six LDS accesses per entry to find the heads; here
// thread t OWNS the RPT consecutive rows t * RPT ... of the row block: their running sums live in its registers from
// the first column block to the last, the rows' entry counts in the tile (one byte per row and column block: cnt8, one
// vector load per thread) give, by a block scan, where its products start in the strip, and it adds them one after the
// other -- the same order, so the same bits.  Same tiled copy (values, columns, tile_ptr); cnt8 instead of rows16.
template <int RPT> struct CbCounts;      // RPT bytes as one vector
template <> struct CbCounts<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <> struct CbCounts<8> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct CbCounts<4> { typedef uint32_t type; };
template <> struct CbCounts<2> { typedef uint16_t type; };
template <> struct CbCounts<1> { typedef uint8_t type; };

template <int RPT>
__device__ __forceinline__ void cb_unpack(const typename CbCounts<RPT>::type &v, uint32_t (&c)[RPT]) {
    if constexpr (RPT >= 8) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) c[r] = (v[r >> 2] >> (8 * (r & 3))) & 0xffu;
    } else {
#pragma unroll
        for (int r = 0; r < RPT; ++r) c[r] = ((uint32_t)v >> (8 * r)) & 0xffu;
    }
}

// cnt8[(rb * nbc + cb) * RB + r] = entries of row r of the row block in column block cb (<= 255, checked by the plan);
// RB = 256 * RPT; every tile holds at most kCbStrip entries (plan).
template <typename T, int RPT, int U>
__global__ __launch_bounds__(kCbThreads) void csr_spmv_cblock_rows(const T *__restrict__ vals, const uint32_t *__restrict__ cols,
                                                                   const uint32_t *__restrict__ tile_ptr,
                                                                   const uint8_t *__restrict__ cnt8, const T *__restrict__ x,
                                                                   T *__restrict__ y, uint32_t nrows, uint32_t nbc) {
    constexpr uint32_t RB = kCbThreads * RPT;
    __shared__ __attribute__((aligned(16))) T strip[kCbStrip > RB ? kCbStrip : RB];
    __shared__ uint32_t s_tp[kCbMaxBlocks + 1];
    __shared__ uint32_t s_wsum[kCbThreads / 64];
    using cnt_t = typename CbCounts<RPT>::type;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t rb = blockIdx.x;
    for (uint32_t i = tid; i <= nbc; i += kCbThreads) s_tp[i] = tile_ptr[(size_t)rb * nbc + i];
    __syncthreads();
    T acc[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) acc[r] = -T(0);
    uint32_t seen = 0;                                    // bit r: row r of this thread holds an entry
    const cnt_t *cnt_base = reinterpret_cast<const cnt_t *>(cnt8 + ((size_t)rb * nbc) * RB) + tid;

    // the first non-empty tile's entries are requested here, every later tile's while the one before it is summed
    T v[U];
    uint32_t c[U];
    uint32_t cb = 0;
    while (cb < nbc && s_tp[cb + 1] == s_tp[cb]) ++cb;    // uniform
    auto request = [&](uint32_t e0, uint32_t n) {         // one batch: entries e0 + tid + 256 * u (clamped: unconditional loads)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = e0 + min(tid + (uint32_t)u * kCbThreads, n - 1u);
            v[u] = load_stream(vals + i);
            c[u] = load_stream(cols + i);
        }
    };
    auto products = [&](uint32_t b0, uint32_t n) {        // the batch in hand -> strip
        T xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = x[c[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = b0 + tid + (uint32_t)u * kCbThreads;
            if (i < n) strip[i] = v[u] * xv[u];
        }
    };
    if (cb < nbc) request(s_tp[cb], s_tp[cb + 1] - s_tp[cb]);
    while (cb < nbc) {
        const uint32_t e0 = s_tp[cb], n = s_tp[cb + 1] - e0;   // 1 <= n <= kCbStrip
        const cnt_t packed = cnt_base[(size_t)cb * (RB / RPT)];
        products(0u, n);
        for (uint32_t b0 = kCbThreads * U; b0 < n; b0 += kCbThreads * U) {   // uniform: tiles above 256 * U entries
            request(e0 + b0, n - b0);
            products(b0, n);
        }
        // the next non-empty tile's first batch travels while this one is summed
        uint32_t nb = cb + 1;
        while (nb < nbc && s_tp[nb + 1] == s_tp[nb]) ++nb;
        if (nb < nbc) request(s_tp[nb], s_tp[nb + 1] - s_tp[nb]);
        // where this thread's rows start in the strip: exclusive scan of the threads' entry counts
        uint32_t cnt[RPT];
        cb_unpack<RPT>(packed, cnt);
        uint32_t mine = 0;
#pragma unroll
        for (int r = 0; r < RPT; ++r) mine += cnt[r];
        const uint32_t inc = cb_wave_inclusive_scan(mine, lane);
        if (lane == 63) s_wsum[w] = inc;
        __syncthreads();                                   // (also: every product is in the strip)
        uint32_t pos = inc - mine;
#pragma unroll
        for (uint32_t i = 0; i < kCbThreads / 64; ++i)
            if (i < w) pos += s_wsum[i];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            for (uint32_t k = 0; k < cnt[r]; ++k) acc[r] = acc[r] + strip[pos++];
            seen |= (cnt[r] ? 1u : 0u) << r;
        }
        __syncthreads();                                   // the strip is written again
        cb = nb;
    }
    // y: through the strip, so that a wave stores 64 consecutive rows; rows without entries are +0.0
#pragma unroll
    for (int r = 0; r < RPT; ++r) strip[tid * RPT + r] = ((seen >> r) & 1u) ? acc[r] : T(0);
    __syncthreads();
    const uint32_t r0 = rb * RB;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const uint32_t i = tid + (uint32_t)r * kCbThreads;
        if (r0 + i < nrows) y[r0 + i] = strip[i];
    }
}

// ---- plan-time kernels ------------------------------------------------------------------------------------------
// cnt8[(rb * nbc + cb) * RB + row in block] = entries of the row in column block cb (columns ascend inside a row);
// *flag |= 1 when a count exceeds 255.  cnt8 is zeroed by the caller.
__global__ __launch_bounds__(256) void cb_count(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
                                                uint32_t nrows, uint32_t RB, uint32_t nbc, uint32_t shift,
                                                uint8_t *__restrict__ cnt8, uint32_t *__restrict__ flag) {
    const uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= nrows) return;
    const uint32_t rb = (uint32_t)(row / RB), rl = (uint32_t)(row % RB);
    uint32_t p = rowptr[row];
    const uint32_t p1 = rowptr[row + 1];
    while (p < p1) {
        const uint32_t cb = colind[p] >> shift;
        uint32_t n = 1;
        ++p;
        while (p < p1 && (colind[p] >> shift) == cb) { ++p; ++n; }
        if (n > 255u) { atomicOr(flag, 1u); n = 255u; }
        cnt8[((size_t)rb * nbc + cb) * RB + rl] = (uint8_t)n;
    }
}

// tile_n[t] = entries of tile t = sum of its RB counts (one workgroup per tile); *runs += rows with entries in the tile
__global__ __launch_bounds__(256) void cb_tile_totals(const uint8_t *__restrict__ cnt8, uint32_t RB,
                                                      uint32_t *__restrict__ tile_n, unsigned long long *__restrict__ runs) {
    __shared__ uint32_t s[4], s_runs[4];
    const uint8_t *c = cnt8 + (size_t)blockIdx.x * RB;
    uint32_t v = 0, nz = 0;
    for (uint32_t i = threadIdx.x; i < RB; i += 256) { v += c[i]; nz += c[i] ? 1u : 0u; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v += (uint32_t)__shfl_xor((int)v, o, 64); nz += (uint32_t)__shfl_xor((int)nz, o, 64); }
    if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6] = v; s_runs[threadIdx.x >> 6] = nz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_n[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
        atomicAdd(runs, (unsigned long long)(s_runs[0] + s_runs[1] + s_runs[2] + s_runs[3]));
    }
}

// the tiled copy: one workgroup per row block; thread t owns the rpt = ceil(RB / 256) consecutive rows t * rpt ... of it
// and moves their entries tile by tile (plan time: bandwidth is not the point)
template <typename T>
__global__ __launch_bounds__(kCbThreads) void cb_fill(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
                                                      const T *__restrict__ values, const uint32_t *__restrict__ tile_ptr,
                                                      const uint8_t *__restrict__ cnt8, uint32_t nrows, uint32_t nbc, uint32_t RB,
                                                      uint32_t *__restrict__ out_col, T *__restrict__ out_val,
                                                      uint16_t *__restrict__ out_row) {
    constexpr int kMaxRpt = (int)(kCbMaxRows / kCbThreads);
    __shared__ uint32_t s_wsum[kCbThreads / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, rb = blockIdx.x;
    const uint32_t rpt = (RB + kCbThreads - 1) / kCbThreads;
    uint32_t src[kMaxRpt];
#pragma unroll
    for (int r = 0; r < kMaxRpt; ++r) {
        const uint32_t rl = tid * rpt + (uint32_t)r;
        const uint64_t row = (uint64_t)rb * RB + rl;
        src[r] = ((uint32_t)r < rpt && rl < RB && row < nrows) ? rowptr[row] : 0u;
    }
    for (uint32_t cb = 0; cb < nbc; ++cb) {
        const uint8_t *c8 = cnt8 + ((size_t)rb * nbc + cb) * RB;
        uint32_t cnt[kMaxRpt], mine = 0;
#pragma unroll
        for (int r = 0; r < kMaxRpt; ++r) {
            const uint32_t rl = tid * rpt + (uint32_t)r;
            cnt[r] = ((uint32_t)r < rpt && rl < RB) ? c8[rl] : 0u;
            mine += cnt[r];
        }
        const uint32_t inc = cb_wave_inclusive_scan(mine, lane);
        if (lane == 63) s_wsum[w] = inc;
        __syncthreads();
        uint32_t pos = tile_ptr[(size_t)rb * nbc + cb] + inc - mine;
#pragma unroll
        for (uint32_t i = 0; i < kCbThreads / 64; ++i)
            if (i < w) pos += s_wsum[i];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kMaxRpt; ++r)
            for (uint32_t k = 0; k < cnt[r]; ++k) {
                out_col[pos] = colind[src[r]];
                out_val[pos] = values[src[r]];
                if (out_row) out_row[pos] = (uint16_t)(tid * rpt + (uint32_t)r);
                ++pos;
                ++src[r];
            }
    }
}

}  // namespace spal
