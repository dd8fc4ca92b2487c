// spal_csr.hip -- CSR handle: create / plan / launch / download.
// C ABI entry points documented in include/spal.h.
#include "csr_kernels.hpp"
#include "csr_slide.hpp"
#include "spal_internal.hpp"

namespace spal {

DeviceGuard::DeviceGuard(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        status = fail(SPAL_ERR_NO_DEVICE, "no HIP device available (%s)",
                      e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return;
    }
    if (device < 0 || device >= count) {
        status = fail(SPAL_ERR_INVALID_ARGUMENT, "device %d out of range (0..%d)", device, count - 1);
        return;
    }
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) {
        e = hipSetDevice(device);
        if (e != hipSuccess) status = fail(SPAL_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    }
}
DeviceGuard::~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
}

// ---- caching allocator for device blocks ------------------------------------------
// hipMalloc / hipFree cost tens of microseconds (hipFree synchronises the device)
// and, for the 600 MB blocks of an assembly, milliseconds on some hosts.  Blocks
// are handed back to a per-process cache instead: large ones (>= 1 MiB) are
// reused for requests up to 25 % smaller, small ones are rounded up to a power of
// two (>= 256 B) and reused for the same class.
namespace {
struct DevCache {
    std::mutex mu;
    struct Block { void *p; size_t bytes; int device; };
    std::vector<Block> free_blocks;
    std::vector<Block> live;      // blocks handed out by dev_alloc (for their size at free time)
    size_t cached_bytes = 0;
    // default: a quarter of the device's memory (72 GB of 288), at least 8 GiB -- the 27 GB of output an
    // assembly of 2.3e9 triplets allocates must be reusable or every call pays hipMalloc / hipFree again
    // (SPAL_CACHE_BYTES overrides; when the device cannot be asked, or is small, the cache stays small: at most
    //  half of what was free at first use)
    size_t limit = [] {
        if (const char *e = getenv("SPAL_CACHE_BYTES")) return (size_t)strtoull(e, nullptr, 10);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return (size_t)256 << 20; }
        return std::min(free_b / 2, std::max((size_t)8 << 30, total_b / 4));
    }();
    ~DevCache() {}  // the process is going away; the driver reclaims device memory
};
DevCache &dev_cache() { static DevCache c; return c; }
constexpr size_t kCacheLargeBytes = 1u << 20;
size_t size_class(size_t bytes) {  // what is actually allocated for a request
    if (bytes >= kCacheLargeBytes) return bytes;
    size_t c = 256;
    while (c < bytes) c <<= 1;
    return c;
}

struct StreamPool {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> idle;   // (device, stream)
};
StreamPool &stream_pool() { static StreamPool p; return p; }
}  // namespace

// Non-blocking streams are pooled: creating one costs ~100 us, and every handle
// (including each assembled CSR result) owns one.
// ---- placement blocks ------------------------------------------------------------------------------------------------
namespace {
struct PlaceArenaBlock {
    void *base = nullptr;
    size_t size = 0;
    std::vector<std::pair<size_t, size_t>> used;   // {offset, bytes}, sorted by offset
};
struct PlaceArena {
    std::mutex mu;
    std::vector<PlaceArenaBlock> blocks[64];        // per device
    bool walked[64] = {};
};
PlaceArena &place_arena() {
    static PlaceArena *a = new PlaceArena;          // (never destroyed: handles may outlive static destructors)
    return *a;
}
}  // namespace
int place_block_count(int device) {
    PlaceArena &a = place_arena();
    std::lock_guard<std::mutex> lock(a.mu);
    return (int)a.blocks[device & 63].size();
}
PlaceBlock place_block(int device, int index) {
    PlaceArena &a = place_arena();
    std::lock_guard<std::mutex> lock(a.mu);
    const auto &v = a.blocks[device & 63];
    if (index < 0 || index >= (int)v.size()) return PlaceBlock{nullptr, 0};
    return PlaceBlock{v[(size_t)index].base, v[(size_t)index].size};
}
void *place_alloc(int device, int index, size_t bytes) {
    PlaceArena &a = place_arena();
    std::lock_guard<std::mutex> lock(a.mu);
    auto &v = a.blocks[device & 63];
    if (index < 0 || index >= (int)v.size() || bytes == 0) return nullptr;
    PlaceArenaBlock &b = v[(size_t)index];
    bytes = (bytes + 4095) & ~(size_t)4095;
    size_t at = 0;
    size_t pos = 0;
    for (; pos < b.used.size(); ++pos) {            // first fit
        if (b.used[pos].first - at >= bytes) break;
        at = b.used[pos].first + b.used[pos].second;
    }
    if (pos == b.used.size() && b.size - at < bytes) return nullptr;
    b.used.insert(b.used.begin() + (long)pos, std::make_pair(at, bytes));
    return (char *)b.base + at;
}
void place_free(int device, void *ptr) {
    if (!ptr) return;
    PlaceArena &a = place_arena();
    std::lock_guard<std::mutex> lock(a.mu);
    for (PlaceArenaBlock &b : a.blocks[device & 63]) {
        if ((char *)ptr < (char *)b.base || (char *)ptr >= (char *)b.base + b.size) continue;
        const size_t off = (size_t)((char *)ptr - (char *)b.base);
        for (size_t i = 0; i < b.used.size(); ++i)
            if (b.used[i].first == off) { b.used.erase(b.used.begin() + (long)i); return; }
    }
}
void place_adopt(int device, void *base, size_t size) {
    PlaceArena &a = place_arena();
    std::lock_guard<std::mutex> lock(a.mu);
    PlaceArenaBlock b;
    b.base = base; b.size = size;
    a.blocks[device & 63].push_back(b);
}
size_t place_free_bytes(int device) {
    PlaceArena &a = place_arena();
    std::lock_guard<std::mutex> lock(a.mu);
    size_t n = 0;
    for (const PlaceArenaBlock &b : a.blocks[device & 63]) {
        size_t u = 0;
        for (const auto &r : b.used) u += r.second;
        n += b.size - u;
    }
    return n;
}
// placement blocks nobody holds a piece of go back to the driver (spal_cache_trim); a device left without any walks again
void place_trim() {
    PlaceArena &a = place_arena();
    std::lock_guard<std::mutex> lock(a.mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < 64; ++d) {
        auto &v = a.blocks[d];
        bool any = false;
        for (size_t i = 0; i < v.size();) {
            if (v[i].used.empty()) {
                if (!any) { (void)hipSetDevice(d); any = true; }
                (void)hipFree(v[i].base);
                v.erase(v.begin() + (long)i);
            } else {
                ++i;
            }
        }
        if (v.empty()) a.walked[d] = false;
    }
    (void)hipSetDevice(cur);
}
bool place_walked(int device) {
    PlaceArena &a = place_arena();
    std::lock_guard<std::mutex> lock(a.mu);
    return a.walked[device & 63];
}
void place_set_walked(int device) {
    PlaceArena &a = place_arena();
    std::lock_guard<std::mutex> lock(a.mu);
    a.walked[device & 63] = true;
}

hipError_t stream_acquire(hipStream_t *out) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    {
        StreamPool &p = stream_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        for (size_t i = 0; i < p.idle.size(); ++i)
            if (p.idle[i].first == device) {
                *out = p.idle[i].second;
                p.idle.erase(p.idle.begin() + i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void stream_release(hipStream_t s) {
    if (!s) return;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        (void)hipStreamDestroy(s);
        return;
    }
    StreamPool &p = stream_pool();
    std::lock_guard<std::mutex> lock(p.mu);
    if (p.idle.size() < 64) p.idle.emplace_back(device, s);
    else (void)hipStreamDestroy(s);
}

hipError_t dev_alloc(void **ptr, size_t bytes) {
    *ptr = nullptr;
    if (bytes == 0) bytes = 1;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    DevCache &c = dev_cache();
    const size_t want = size_class(bytes);
    {
        std::lock_guard<std::mutex> lock(c.mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < c.free_blocks.size(); ++i) {
            const auto &b = c.free_blocks[i];
            const bool fits = want >= kCacheLargeBytes ? (b.bytes >= want && b.bytes <= want + want / 4)
                                                       : b.bytes == want;
            if (b.device == device && fits && (best == (size_t)-1 || b.bytes < c.free_blocks[best].bytes))
                best = i;
        }
        if (best != (size_t)-1) {
            DevCache::Block b = c.free_blocks[best];
            c.free_blocks.erase(c.free_blocks.begin() + best);
            c.cached_bytes -= b.bytes;
            c.live.push_back(b);
            *ptr = b.p;
            return hipSuccess;
        }
    }
    e = hipMalloc(ptr, want);
    if (e == hipErrorOutOfMemory) {  // give the cache back and retry once
        (void)hipGetLastError();
        dev_cache_trim();
        e = hipMalloc(ptr, want);
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lock(c.mu);
        c.live.push_back({*ptr, want, device});
    }
    return e;
}

hipError_t dev_free(void *ptr) {
    if (!ptr) return hipSuccess;
    DevCache &c = dev_cache();
    DevCache::Block b{nullptr, 0, 0};
    {
        std::lock_guard<std::mutex> lock(c.mu);
        for (size_t i = 0; i < c.live.size(); ++i)
            if (c.live[i].p == ptr) { b = c.live[i]; c.live.erase(c.live.begin() + i); break; }
    }
    if (!b.p) return hipFree(ptr);  // not ours: straight back
    // what hipFree would have done: no user of the block is still running -- on the block's OWN device, whatever
    // device the caller has selected
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != b.device) (void)hipSetDevice(b.device);
    hipError_t e = hipDeviceSynchronize();
    if (cur >= 0 && cur != b.device) (void)hipSetDevice(cur);
    std::lock_guard<std::mutex> lock(c.mu);
    if (e == hipSuccess && c.cached_bytes + b.bytes <= c.limit) {
        c.free_blocks.push_back(b);
        c.cached_bytes += b.bytes;
        return hipSuccess;
    }
    if (cur != b.device) (void)hipSetDevice(b.device);
    e = hipFree(ptr);
    if (cur >= 0 && cur != b.device) (void)hipSetDevice(cur);
    return e;
}

void dev_cache_trim() {
    DevCache &c = dev_cache();
    std::vector<DevCache::Block> blocks;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        blocks.swap(c.free_blocks);
        c.cached_bytes = 0;
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (auto &b : blocks) { (void)hipSetDevice(b.device); (void)hipFree(b.p); }
    if (prev >= 0) (void)hipSetDevice(prev);
}

// ---- per-row-block column window ---------------------------------------------
// Columns are strictly increasing inside a row (src/csr.rs:152-156), so a
// row's first and last stored column bound it.  One workgroup per row block:
// out[b] = {min first column, max last column + 1}, {0xffffffff, 0} if the
// block stores nothing.
__global__ __launch_bounds__(256) void csr_block_windows(
    const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind, uint32_t nrows,
    uint32_t R, uint2 *__restrict__ out) {
    __shared__ uint32_t s_min, s_max;
    if (threadIdx.x == 0) {
        s_min = 0xffffffffu;
        s_max = 0u;
    }
    __syncthreads();
    const uint32_t row0 = blockIdx.x * R;
    const uint32_t row1 = min(row0 + R, nrows);
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (uint32_t r = row0 + threadIdx.x; r < row1; r += 256) {
        const uint32_t a0 = rowptr[r], a1 = rowptr[r + 1];
        if (a0 < a1) {
            lo = min(lo, colind[a0]);
            hi = max(hi, colind[a1 - 1] + 1u);
        }
    }
    atomicMin(&s_min, lo);
    atomicMax(&s_max, hi);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = make_uint2(s_min, s_max);
}

// ---- plan -------------------------------------------------------------------
// LDS budget for the x window of the vector kernel.  160 KiB per CU; 72 KiB per
// workgroup keeps two workgroups resident, smaller windows admit more.  (72 rather than 64: a band of 8192
// columns under 64 rows of 1500 entries is 66 KiB.)
static constexpr uint32_t kLdsBudgetBytes = 72 * 1024;
// stream kernel: 4 product strips (kStreamTileNnz each, 32 KiB f64) + a window of
// at most 48 KiB -> at most 80 KiB per workgroup, two workgroups per CU (skewed strips, 34 KiB: 44 KiB).
static constexpr uint32_t kStreamWindowBytes = 48 * 1024;
static constexpr uint32_t kStreamWindowBytesSkew = 44 * 1024;
// ... or one workgroup per CU with a window of up to 120 KiB (+ 32 / 34 KiB of strips), for matrices whose
// super-tiles touch more pages than 48 KiB hold: LDS gathers at half the occupancy still beat x through L2
// (band of 8192 columns, f64: 463 / 392 us against 722 us)
static constexpr uint32_t kStreamBigWindowBytes = 120 * 1024;

// One workgroup per super-tile: chk[b] = {skip bits, cost, entries, rows a multiple of 128 bytes long | ulen << 16},
// ulen = 1 + the length of every row of the super-tile when they are all equal (and below 4095), else 0.
//  - skip: a bit per tile that the stream kernels must leave to csr_spmv_overflow -- it holds more entries
//    than the product strip, or a row of more than row_max entries (the stream kernels sum a row per lane:
//    such a row keeps 63 lanes waiting, 25 cycles per entry);
//  - cost: what the super-tile's tiles cost at this tile height, in entries: a streamed tile as much as a
//    half-full one at least (its fixed work: 160 entries per 16-row tile ran 2.1 x slower than 400 per 64-row tile),
//    a skipped tile 1.5 per entry + 1000 (its own workgroup in the overflow kernel).  The planner takes the
//    tile height with the smallest sum (power-law rows, 10 per row on average: 64 / 32 / 16 rows per tile
//    predicted 1 : 1.37 : 2.1, measured 196 : 270 : 380 us).
constexpr uint32_t kTileFloorEntries = 512, kOverflowTileFixed = 1000;
__global__ __launch_bounds__(256) void csr_stream_check(const uint32_t *__restrict__ rowptr,
                                                        uint32_t nrows, uint32_t R, uint32_t rpt,
                                                        uint32_t row_max, uint32_t quantum,
                                                        uint4 *__restrict__ chk) {
    __shared__ uint32_t s_long[32];
    __shared__ uint32_t s_aligned, s_ragged;
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    const uint32_t row0 = b * R, row1 = min(row0 + R, nrows);
    if (t < 32) s_long[t] = 0u;
    if (t == 0) { s_aligned = 0u; s_ragged = 0u; }
    __syncthreads();
    const uint32_t len0 = rowptr[row0 + 1] - rowptr[row0];
    uint32_t aligned = 0;   // rows a non-zero multiple of `quantum` entries (128 bytes) long: see SKEW in csr_kernels.hpp
    for (uint32_t r = row0 + t; r < row1; r += 256) {
        const uint32_t len = rowptr[r + 1] - rowptr[r];
        if (len > row_max) s_long[(r - row0) / rpt] = 1u;   // (same value from every writer)
        if (len != len0) s_ragged = 1u;
        aligned += (len != 0u && len % quantum == 0u) ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) aligned += (uint32_t)__shfl_xor((int)aligned, o, 64);
    if ((t & 63u) == 0 && aligned) atomicAdd(&s_aligned, aligned);
    __syncthreads();
    if (t < 64) {   // R / rpt <= 32 tiles
        const uint32_t r0 = row0 + t * rpt;
        bool bad = false;
        uint32_t cost = 0;
        if (t < R / rpt && r0 < row1) {
            const uint32_t rl = min(r0 + rpt, row1);
            const uint32_t e0 = rowptr[r0], e1 = rowptr[rl];
            bad = stream_tile_overflows(e0, e1) || s_long[t] != 0u;
            const uint32_t n = e1 - e0;
            cost = bad ? n + n / 2 + kOverflowTileFixed : max(n, kTileFloorEntries);
        }
        const uint64_t m = __ballot(bad);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cost += (uint32_t)__shfl_xor((int)cost, o, 64);
        if (t == 0) {
            const uint32_t ulen = (s_ragged == 0u && len0 < 4095u) ? len0 + 1u : 0u;
            chk[b] = make_uint4((uint32_t)m, cost, rowptr[row1] - rowptr[row0], s_aligned | (ulen << 16));
        }
    }
}

// Plan time: the first rows of the tiles the descriptors mark (in pieces of at most 64 rows), appended in any order.
__global__ __launch_bounds__(256) void csr_overflow_tiles(const uint4 *__restrict__ desc, uint32_t nrows,
                                                          uint32_t R, uint32_t rpt, uint32_t cap,
                                                          uint32_t *__restrict__ count,
                                                          uint32_t *__restrict__ tiles) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t r0 = i * rpt;
    if (r0 >= nrows) return;
    const uint4 d = desc[r0 / R];
    const uint32_t mode = desc_mode(d);
    if (mode != kModeStream && mode != kModeStreamGlobal) return;
    if (!((desc_skip_bits(d) >> (uint32_t)((r0 % R) / rpt)) & 1u)) return;
    for (uint64_t r = r0; r < min(r0 + rpt, (uint64_t)nrows); r += 64) {   // (csr_spmv_overflow takes up to 64 rows a piece)
        const uint32_t at = atomicAdd(count, 1u);
        if (at < cap) tiles[at] = (uint32_t)r;
    }
}

// ---- the pages a super-tile's rows touch ----------------------------------------------
// One workgroup per super-tile of R rows.  info[b] = {first column, one past the last
// column, number of pages or kNotPageable, 1 if the pages are the contiguous run that
// starts at page (first column >> kPageShift)}.  When the span holds at most `run_cap` pages
// the run is taken whole (a band); otherwise the columns are marked in an LDS bitmap
// (spans up to 16.7M columns), first for a sample of 2048 entries -- scattered columns
// are recognised and dropped there -- then for all of them, and the pages are listed in
// ascending order at pages[b * cap ...].
constexpr uint32_t kNotPageable = 0xffffffffu;
constexpr uint32_t kPageBitmapWords = 2048;   // 65536 pages
__global__ __launch_bounds__(256) void csr_block_pages(
    const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind, uint32_t nrows,
    uint32_t R, uint32_t cap, uint32_t run_cap, const uint2 *__restrict__ known_win,
    uint4 *__restrict__ info, uint32_t *__restrict__ pages) {
    __shared__ uint32_t s_bits[kPageBitmapWords];
    __shared__ uint32_t s_min, s_max, s_count, s_wsum[4];
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    if (t == 0) { s_min = 0xffffffffu; s_max = 0u; s_count = 0u; }
    __syncthreads();
    const uint32_t row0 = b * R, row1 = min(row0 + R, nrows);
    if (known_win) {   // the caller already knows {first column, one past the last} of this super-tile
        if (t == 0) { s_min = known_win[b].x; s_max = known_win[b].y; }
    } else {
        uint32_t lo = 0xffffffffu, hi = 0u;
        for (uint32_t r = row0 + t; r < row1; r += 256) {
            const uint32_t a0 = rowptr[r], a1 = rowptr[r + 1];
            if (a0 < a1) {   // columns ascend inside a row: its first and last entry bound it
                lo = min(lo, colind[a0]);
                hi = max(hi, colind[a1 - 1] + 1u);
            }
        }
        atomicMin(&s_min, lo);
        atomicMax(&s_max, hi);
    }
    __syncthreads();
    const uint32_t cmin = s_min, cmax = s_max;
    if (cmax == 0) {   // nothing stored
        if (t == 0) info[b] = make_uint4(0xffffffffu, 0u, 0u, 1u);
        return;
    }
    const uint32_t pmin = cmin >> kPageShift, span = ((cmax - 1u) >> kPageShift) - pmin + 1u;
    if (span <= run_cap) {   // (run_cap <= cap: the budget that keeps two workgroups per CU)
        if (t == 0) info[b] = make_uint4(cmin, cmax, span, 1u);
        return;
    }
    if (span > kPageBitmapWords * 32u) {
        if (t == 0) info[b] = make_uint4(cmin, cmax, kNotPageable, 0u);
        return;
    }
    const uint32_t words = (span + 31u) / 32u;
    for (uint32_t i = t; i < words; i += 256) s_bits[i] = 0u;
    __syncthreads();
    const uint32_t e0 = rowptr[row0], e1 = rowptr[row1];
    const uint32_t es = min(e0 + 2048u, e1);
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t a0 = pass ? es : e0, a1 = pass ? e1 : es;
        for (uint32_t e = a0 + t; e < a1; e += 256) {
            const uint32_t pg = (colind[e] >> kPageShift) - pmin;
            atomicOr(&s_bits[pg >> 5], 1u << (pg & 31u));
        }
        __syncthreads();
        uint32_t c = 0;
        for (uint32_t i = t; i < words; i += 256) c += (uint32_t)__popc(s_bits[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
        if ((t & 63u) == 0) s_wsum[t >> 6] = c;
        __syncthreads();
        const uint32_t count = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        __syncthreads();
        if (count > cap) {   // block-uniform
            if (t == 0) info[b] = make_uint4(cmin, cmax, kNotPageable, 0u);
            return;
        }
        if (pass == 1 && t == 0) s_count = count;
    }
    // ascending page list: thread t owns the words [8t, 8t + 8)
    constexpr uint32_t kPer = kPageBitmapWords / 256;
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
        const uint32_t w = t * kPer + q;
        if (w < words) mine += (uint32_t)__popc(s_bits[w]);
    }
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64);
        if ((t & 63u) >= (uint32_t)o) inc += v;
    }
    if ((t & 63u) == 63u) s_wsum[t >> 6] = inc;
    __syncthreads();
    uint32_t rank = inc - mine;
    for (uint32_t i = 0; i < (t >> 6); ++i) rank += s_wsum[i];
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
        const uint32_t w = t * kPer + q;
        if (w < words) {
            uint32_t bits = s_bits[w];
            while (bits) {
                const uint32_t bit = (uint32_t)__ffs((int)bits) - 1u;
                bits &= bits - 1u;
                pages[(size_t)b * cap + rank++] = pmin + w * 32u + bit;
            }
        }
    }
    if (t == 0) info[b] = make_uint4(cmin, cmax, s_count, s_count == span ? 1u : 0u);   // (every page of the span: a run after all)
}

// Vector plans with long rows: col16 = column - window base for the blocks (R rows) whose x window is in LDS.
__global__ __launch_bounds__(256) void csr_encode_col16_window(const uint32_t *__restrict__ rowptr,
                                                               const uint32_t *__restrict__ colind,
                                                               const uint4 *__restrict__ desc,
                                                               uint16_t *__restrict__ col16, uint32_t nrows,
                                                               uint32_t R) {
    const uint32_t b = blockIdx.x;
    const uint4 d = desc[b];
    if (d.z != kModeVectorLds) return;
    const uint32_t row0 = b * R, row1 = min(row0 + R, nrows);
    const uint32_t e0 = rowptr[row0], e1 = rowptr[row1];
    for (uint32_t e = e0 + threadIdx.x; e < e1; e += 256) col16[e] = (uint16_t)(colind[e] - d.x);
}

// One workgroup per super-tile: col16 = slot of the column's page * kPageCols + column
// inside the page (the slot by binary search in the super-tile's ascending page list).
__global__ __launch_bounds__(256) void csr_encode_col16(const uint32_t *__restrict__ rowptr,
                                                        const uint32_t *__restrict__ colind,
                                                        const uint4 *__restrict__ desc,
                                                        const uint32_t *__restrict__ pages,
                                                        uint16_t *__restrict__ col16,
                                                        uint32_t nrows, uint32_t R, uint32_t ring) {
    __shared__ uint32_t s_pg[64];
    uint4 d = desc[blockIdx.x];   // Stream: {first page / offset, npages | ulen << 8, mode, contiguous}
    if (desc_mode(d) != kModeStream) return;
    d.y &= 0xffu;
    const uint32_t row0 = blockIdx.x * R, row1 = min(row0 + R, nrows);
    const uint32_t p0 = rowptr[row0], p1 = rowptr[row1];
    if (d.w & 1u) {   // contiguous run of pages starting at page d.x
        if (ring) {   // the window is a ring: slot = page % ring (csr_slide.hpp)
            for (uint32_t p = p0 + threadIdx.x; p < p1; p += 256) {
                const uint32_t c = colind[p];
                col16[p] = (uint16_t)((((c >> kPageShift) % ring) << kPageShift) | (c & (kPageCols - 1u)));
            }
            return;
        }
        const uint32_t base = d.x << kPageShift;
        for (uint32_t p = p0 + threadIdx.x; p < p1; p += 256) col16[p] = (uint16_t)(colind[p] - base);
        return;
    }
    if (threadIdx.x < 64) s_pg[threadIdx.x] = threadIdx.x < d.y ? pages[d.x + threadIdx.x] : 0xffffffffu;
    __syncthreads();
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += 256) {
        const uint32_t c = colind[p], pg = c >> kPageShift;
        uint32_t lo = 0, hi = d.y;   // the page is in the list: first slot with s_pg[slot] >= pg
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_pg[mid] < pg) lo = mid + 1; else hi = mid;
        }
        col16[p] = (uint16_t)(lo * kPageCols + (c & (kPageCols - 1u)));
    }
}

// Column-panel super-tiles (csr_panel.hpp): col16[p] = column - first column of the super-tile's span (at most 192
// pages = 49 152 columns: 16 bits).  One workgroup per listed super-tile.
__global__ __launch_bounds__(256) void csr_encode_col16_span(const uint32_t *__restrict__ rowptr,
                                                             const uint32_t *__restrict__ colind,
                                                             const uint32_t *__restrict__ ptiles,
                                                             const uint2 *__restrict__ pwin, uint16_t *__restrict__ col16,
                                                             uint32_t nrows, uint32_t R) {
    const uint32_t b = ptiles[blockIdx.x], c0 = pwin[blockIdx.x].x * kPageCols;
    const uint32_t e0 = rowptr[min(b * R, nrows)], e1 = rowptr[min(b * R + R, nrows)];
    for (uint32_t p = e0 + threadIdx.x; p < e1; p += 256) col16[p] = (uint16_t)(colind[p] - c0);
}

// ---- plan of the sliding-window kernel (csr_slide.hpp) -----------------------------------------------------
// One workgroup per STEP of SR = 4 * rpt rows (SR <= 256: a row per thread): out[i] = {first column, one past the
// last column (0: the step stores nothing), 1 + the length of every row if they are all equal else 0, the most
// 128-entry steps one of its four tiles needs | a bit per tile << 8 that can go through the strip in two HALVES}.
// A tile above the strip's 1024 entries whose halves (rpt / 2 rows each) both fit and that holds no row longer
// than row_max is such a tile: the sliding kernel takes it in two passes instead of leaving it to
// csr_spmv_overflow (rows of 1 ... 27 entries: 2 % of the 64-row tiles).  Its halves count for the step number,
// a tile that is left to the overflow kernel does not.
__global__ __launch_bounds__(256) void csr_slide_scan(const uint32_t *__restrict__ rowptr,
                                                      const uint32_t *__restrict__ colind, uint32_t nrows,
                                                      uint32_t rpt, uint32_t row_max, uint4 *__restrict__ out) {
    __shared__ uint32_t s_min, s_max, s_ragged, s_steps, s_long, s_split;
    const uint32_t t = threadIdx.x, SR = 4u * rpt;
    if (t == 0) { s_min = 0xffffffffu; s_max = 0u; s_ragged = 0u; s_steps = 0u; s_long = 0u; s_split = 0u; }
    __syncthreads();
    const uint32_t row0 = blockIdx.x * SR, row1 = min(row0 + SR, nrows);
    const uint32_t len0 = rowptr[row0 + 1] - rowptr[row0];
    if (row0 + t < row1 && t < SR) {
        const uint32_t a0 = rowptr[row0 + t], a1 = rowptr[row0 + t + 1];
        if (a0 < a1) {
            atomicMin(&s_min, colind[a0]);
            atomicMax(&s_max, colind[a1 - 1] + 1u);
        }
        if (a1 - a0 != len0) s_ragged = 1u;
        if (a1 - a0 > row_max) atomicOr(&s_long, 1u << (t / rpt));
    }
    __syncthreads();
    if (t < 4u && row0 + t * rpt < row1) {
        const uint32_t rb = row0 + t * rpt, re = min(rb + rpt, row1), rm = min(rb + rpt / 2u, re);
        const uint32_t b = rowptr[rb], m = rowptr[rm], e = rowptr[re];
        uint32_t steps = (e - (b & ~1u) + 127u) >> 7;
        if (stream_tile_overflows(b, e)) {
            const bool halves = rpt >= 2u && !((s_long >> t) & 1u) && !stream_tile_overflows(b, m) && !stream_tile_overflows(m, e);
            steps = halves ? max((m - (b & ~1u) + 127u) >> 7, (e - (m & ~1u) + 127u) >> 7) : 0u;
            if (halves) atomicOr(&s_split, 1u << t);
        }
        atomicMax(&s_steps, steps);
    }
    __syncthreads();
    if (t == 0)
        out[blockIdx.x] = make_uint4(s_min, s_max, (s_ragged == 0u && len0 < 4095u) ? len0 + 1u : 0u, s_steps | (s_split << 8));
}

// Decides whether the sliding kernel can run this stream plan and, if so, builds its step descriptors.
// desc / skip: the chosen stream plan's super-tiles (R rows each, 16 tiles of rpt rows); super_pages: the most
// pages one of them stages (the one-super-tile-per-workgroup kernels read the same ring).
static int slide_plan(spal_csr *a, uint32_t R, uint32_t rpt, const std::vector<uint4> &desc,
                      const std::vector<uint32_t> &skip, uint32_t super_pages) {
    CsrPlan &p = a->plan;
    p.slide = 0;
    p.ring_pages = 0;
    if (a->d_sdesc) { SPAL_HIP_TRY(dev_free(a->d_sdesc)); a->d_sdesc = nullptr; }
    if (a->d_ovtiles_slide) { SPAL_HIP_TRY(dev_free(a->d_ovtiles_slide)); a->d_ovtiles_slide = nullptr; }
    a->n_ovtiles_slide = 0;
    a->n_split_tiles = 0;
    const uint32_t V = 16u / (uint32_t)a->elem_size;
    if (p.slide_user == 0 || p.tiles_per_wave != 4 || rpt > 64u || p.skew || a->ncols < kPageCols || a->nnz == 0) return SPAL_OK;
    for (const uint4 &d : desc)
        if (d.z != kModeStream || !(d.w & 1u)) return SPAL_OK;   // a page list, x through L2, vector rows: not a band
    const uint32_t SR = 4u * rpt;
    const uint32_t nsteps = (uint32_t)((a->nrows + SR - 1) / SR);
    uint4 *d_scan = nullptr;
    SPAL_HIP_TRY(dev_alloc((void **)&d_scan, (size_t)nsteps * sizeof(uint4)));
    hipLaunchKernelGGL(csr_slide_scan, dim3(nsteps), dim3(256), 0, a->stream, a->d_rowptr, a->d_colind,
                       (uint32_t)a->nrows, rpt, (uint32_t)p.stream_row_max, d_scan);
    std::vector<uint4> scan(nsteps);
    hipError_t e = hipMemcpyAsync(scan.data(), d_scan, (size_t)nsteps * sizeof(uint4), hipMemcpyDeviceToHost, a->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(a->stream);
    (void)dev_free(d_scan);
    SPAL_HIP_TRY(e);
    // windows: a step that stores nothing keeps its predecessor's window (nothing enters)
    std::vector<uint2> sd(nsteps);
    std::vector<uint32_t> left_over;     // first rows of the tiles csr_spmv_overflow computes when the sliding kernel runs
    uint32_t n_split = 0;
    uint32_t S = 1, ulen = scan[0].z;
    bool uni = true;
    uint32_t pf = 0, pn = 1;
    for (uint32_t i = 0; i < nsteps; ++i) {
        if (scan[i].y) {
            pf = scan[i].x >> kPageShift;
            pn = ((scan[i].y - 1u) >> kPageShift) - pf + 1u;
        }
        sd[i] = make_uint2(pf, pn);
        uni = uni && scan[i].z != 0u && scan[i].z == ulen;
        // tiles the stream kernels skip: those that fit the strip in two halves stay with the sliding kernel (split),
        // the others go to csr_spmv_overflow and do not bound S
        const uint32_t tile0 = i * 4u;
        uint32_t sk = 0;
        for (uint32_t w = 0; w < 4u; ++w) {
            const uint32_t tl = tile0 + w, b = tl / 16u;
            if (b < skip.size() && ((skip[b] >> (tl % 16u)) & 1u)) sk |= 1u << w;
        }
        const uint32_t sp = p.split_tiles_on ? (sk & ((scan[i].w >> 8) & 0xfu)) : 0u;
        sd[i].y |= (sk << 8) | (sp << 20);
        S = std::max(S, scan[i].w & 0xffu);
        for (uint32_t w = 0; w < 4u; ++w) {
            if ((sp >> w) & 1u) ++n_split;
            else if ((sk >> w) & 1u) left_over.push_back((tile0 + w) * rpt);
        }
    }
    if (S > (uint32_t)kStreamSteps) S = (uint32_t)kStreamSteps;   // (cannot be: the scan counts fitting tiles and halves only)
    // ring size: what the largest super-tile stages, and room for the pages that enter with the next step
    const uint32_t page_bytes = kPageCols * (uint32_t)a->elem_size;
    const uint32_t strips = (uint32_t)kStreamWaves * (uint32_t)stream_strip<false>() * (uint32_t)a->elem_size;
    const uint32_t cap2 = (80u * 1024u - strips) / page_bytes;            // two workgroups per CU
    const uint32_t cap1 = std::min<uint32_t>(255u, (160u * 1024u - strips) / page_bytes);   // one
    uint32_t want = super_pages;
    for (uint32_t i = 0; i + 1 < nsteps; ++i) {
        const uint32_t lo = std::min(sd[i].x, sd[i + 1].x);
        const uint32_t hi = std::max(sd[i].x + (sd[i].y & 0xffu), sd[i + 1].x + (sd[i + 1].y & 0xffu));
        want = std::max(want, hi - lo);
    }
    const uint32_t NP = want <= cap2 ? want : std::min(want, std::max(cap1, super_pages));
    if (NP < super_pages || NP > 255u) return SPAL_OK;   // (cannot be: super_pages fits the budget it was planned for)
    // which steps' entering pages are prefetched
    const uint32_t safe_cols = (uint32_t)(a->ncols / V) * V;   // below this column, x is made of whole 16-byte vectors
    const uint32_t VP = kPageCols / V;
    for (uint32_t i = 1; i < nsteps; ++i) {
        const uint32_t f0 = sd[i - 1].x, e0 = f0 + (sd[i - 1].y & 0xffu), f1 = sd[i].x, e1 = f1 + (sd[i].y & 0xffu);
        const uint32_t lo = std::min(f0, f1), hi = std::max(e0, e1);
        uint32_t entering = 0;
        if (f0 >= e1 || e0 <= f1) entering = e1 - f1;
        else entering = (f1 < f0 ? f0 - f1 : 0u) + (e1 > e0 ? e1 - e0 : 0u);
        const bool whole = (uint64_t)e1 * kPageCols <= safe_cols;
        if (hi - lo <= NP && entering * VP <= kSlideAsyncVecs * (uint32_t)kStreamBlock && whole) sd[i].y |= kSlideAsync;
    }
    SPAL_HIP_TRY(dev_alloc((void **)&a->d_sdesc, (size_t)nsteps * sizeof(uint2)));
    SPAL_HIP_TRY(hipMemcpyAsync(a->d_sdesc, sd.data(), (size_t)nsteps * sizeof(uint2), hipMemcpyHostToDevice, a->stream));
    a->n_ovtiles_slide = (uint32_t)left_over.size();
    a->n_split_tiles = n_split;
    if (!left_over.empty()) {
        SPAL_HIP_TRY(dev_alloc((void **)&a->d_ovtiles_slide, left_over.size() * 4));
        SPAL_HIP_TRY(hipMemcpyAsync(a->d_ovtiles_slide, left_over.data(), left_over.size() * 4, hipMemcpyHostToDevice, a->stream));
    }
    SPAL_HIP_TRY(hipStreamSynchronize(a->stream));   // `sd`, `left_over` go out of scope
    p.slide = 1;
    p.ring_pages = (int)NP;
    p.slide_steps = nsteps;
    p.slide_S = (int)S;
    p.slide_uniform = (uni && ulen != 0u) ? (int)ulen : 0;
    // every tile of the sliding kernel issues S loads per array (counted waits): where the tiles are on average less than
    // 60 % of the largest one -- ragged short rows, the short part of a row split: 4 of 8 steps -- half its loads are
    // re-reads, and the one-super-tile-per-workgroup kernel, which issues what a tile holds, is faster (power-law short
    // part: 104 -> 68 us); the autotune still times both
    {
        const double tiles = (double)nsteps * kStreamWaves;
        const double avg_steps = tiles > 0 ? (double)a->nnz / tiles / 128.0 : 0.0;
        p.slide_fill_ok = (p.slide_fill_user >= 0) ? p.slide_fill_user : (avg_steps >= 0.6 * (double)std::max(4u, S) ? 1 : 0);
    }
    // every row of the matrix that long: all steps stream (no tile left to the overflow kernel, none split) and the entries
    // add up
    p.all_rows_uniform = (p.slide_uniform && left_over.empty() && n_split == 0 &&
                          (uint64_t)a->nrows * (uint64_t)(ulen - 1u) == a->nnz) ? 1 : 0;
    (void)R;
    return SPAL_OK;
}

static int pick_lanes(double mean_row) {
    int L = 2;
    while (L < 64 && (double)L < mean_row) L <<= 1;
    return L;
}

template <typename K>
static hipError_t raise_lds_cap(K kern, int device, size_t lds, std::atomic<uint64_t> &configured) {
    if (lds <= 48 * 1024) return hipSuccess;
    const uint64_t bit = 1ull << (device & 63);
    if (configured.load(std::memory_order_relaxed) & bit) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e == hipSuccess) configured.fetch_or(bit, std::memory_order_relaxed);
    return e;
}

template <typename T, int L, int U, bool LDSX, int BLOCK, int LB = 1>
static hipError_t launch_vec(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    const CsrPlan &p = a->plan;
    const uint32_t per_xcd = (p.nblocks + 7) / 8;
    const size_t lds = LDSX ? (size_t)p.lds_entries * sizeof(T) : 0;
    auto kern = csr_spmv_vector<T, L, U, LDSX, true, BLOCK, LB>;
    static std::atomic<uint64_t> configured{0};
    hipError_t e = raise_lds_cap(kern, a->device, lds, configured);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(BLOCK), lds, st, a->d_rowptr, a->d_colind,
                       (const T *)a->d_values, (const T *)x, (T *)y, a->d_desc, (uint32_t)a->nrows,
                       (uint32_t)a->nnz, (uint32_t)p.rows_per_block, p.nblocks, per_xcd);
    return hipGetLastError();
}

// long rows, x windows in LDS: the kernel that reads 16-bit window-relative columns (plan: vec_col16)
template <typename T, int L, int BLOCK>
static hipError_t launch_vec_col16(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    const CsrPlan &p = a->plan;
    const uint32_t per_xcd = (p.nblocks + 7) / 8;
    const size_t lds = (size_t)p.lds_entries * sizeof(T);
    auto kern = csr_spmv_vector_col16<T, L, BLOCK, 1>;   // (the batched rest-of-row loop, LB = 4, measured 10 ... 20 % slower here)
    static std::atomic<uint64_t> configured{0};
    hipError_t e = raise_lds_cap(kern, a->device, lds, configured);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(BLOCK), lds, st, a->d_rowptr, a->d_colind, a->d_col16,
                       (const T *)a->d_values, (const T *)x, (T *)y, a->d_desc, (uint32_t)a->nrows,
                       (uint32_t)a->nnz, (uint32_t)p.rows_per_block, p.nblocks, per_xcd);
    return hipGetLastError();
}

template <typename T, int L, int U, bool LDSX>
static hipError_t launch_vec_block(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    if constexpr ((L == 64 || L == 32) && U == 1 && LDSX) {
        if (a->plan.vec_col16)
            return a->plan.threads == 1024 ? launch_vec_col16<T, L, 1024>(a, x, y, st)
                                           : launch_vec_col16<T, L, 512>(a, x, y, st);
    }
    if constexpr (L == 16) {  // the long-row form exists for 16 lanes per row only (the planner's choice)
        if (a->plan.long_rows)
            return a->plan.threads == 1024 ? launch_vec<T, L, U, LDSX, 1024, 4>(a, x, y, st)
                                           : launch_vec<T, L, U, LDSX, 512, 4>(a, x, y, st);
    }
    return a->plan.threads == 1024 ? launch_vec<T, L, U, LDSX, 1024>(a, x, y, st)
                                   : launch_vec<T, L, U, LDSX, 512>(a, x, y, st);
}

template <typename T, int L, int U>
static hipError_t launch_vec_lds(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    return a->plan.lds_x ? launch_vec_block<T, L, U, true>(a, x, y, st)
                         : launch_vec_block<T, L, U, false>(a, x, y, st);
}

template <typename T, int L>
static hipError_t launch_vec_unroll(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    switch (a->plan.unroll) {
        case 1: return launch_vec_lds<T, L, 1>(a, x, y, st);
        case 2: return launch_vec_lds<T, L, 2>(a, x, y, st);
        case 4: return launch_vec_lds<T, L, 4>(a, x, y, st);
        default: return hipErrorInvalidValue;
    }
}

// the panel kernel takes the flagged super-tiles (its page loads are 16-byte vectors of x); otherwise the
// stream kernels gather x for them from global memory, as for any other super-tile wider than LDS
static bool panel_runs(const spal_csr *a, const void *x) {
    return a->plan.panel_on && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
}

// stream kernel; its vector fallback for non-streamable super-tiles uses U = 2
template <typename T, int TPW, int RPT, bool SKEW = false, int PF = 1>
static hipError_t launch_stream_tpw(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    // the in-kernel vector fallback takes super-tiles with a tile of more than 1024 entries, i.e. with heavy
    // rows: a wave per row, four (colind, value) pairs per lane in flight (any geometry is correct)
    constexpr int L = 64;
    const CsrPlan &p = a->plan;
    // the 8 XCDs: one contiguous run of super-tiles each, or (option "xcd_chunk", default 32) interleaved in chunks
    const uint32_t C = (uint32_t)p.xcd_chunk;
    const uint32_t per_xcd = C ? (kXcdChunked | C) : (p.nblocks + 7) / 8;
    const uint32_t grid = C ? ((p.nblocks + 8 * C - 1) / (8 * C)) * 8 * C : ((p.nblocks + 7) / 8) * 8;
    const size_t lds = ((size_t)kStreamWaves * stream_strip<SKEW>() + p.lds_entries) * sizeof(T);
    auto kern = csr_spmv_stream<T, L, 1, true, TPW, RPT, SKEW, PF>;
    static std::atomic<uint64_t> configured{0};
    hipError_t e = raise_lds_cap(kern, a->device, lds, configured);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kStreamBlock), lds, st, a->d_rowptr, a->d_colind,
                       a->d_col16, (const T *)a->d_values, (const T *)x, (T *)y, a->d_desc, a->d_pages,
                       (uint32_t)a->nrows, (uint32_t)a->ncols, (uint32_t)a->nnz, p.nblocks, per_xcd,
                       (uint32_t)(p.nt_store == 1 ? 1 : 0) | ((a->n_ptiles && panel_runs(a, x)) ? 2u : 0u) | (uint32_t)p.diag,
                       (uint32_t)p.ring_pages);
    return hipGetLastError();
}

// persistent form: 2 workgroups per CU, contiguous chunks of each XCD's run
template <typename T, int TPW, int RPT, bool SKEW = false>
static hipError_t launch_stream_persistent(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    constexpr int L = 64;
    const CsrPlan &p = a->plan;
    const uint32_t per_xcd = (p.nblocks + 7) / 8;
    const size_t lds = ((size_t)kStreamWaves * stream_strip<SKEW>() + p.lds_entries) * sizeof(T);
    // grid: as many workgroups as the device holds at once -- 160 KiB of LDS per CU decide
    // (f64 band: 2 per CU = 512; f32, whose strips and window are half the size: 4 per CU)
    int grid = p.persistent_blocks;
    if (grid <= 0) {
        const int per_cu = (int)std::min<size_t>(8, std::max<size_t>(1, (160 * 1024) / (lds + 1024)));
        grid = 256 * per_cu;
    }
    const uint32_t slots = (uint32_t)std::max(1, grid / 8);                   // workgroups per XCD
    const uint32_t chunk = (per_xcd + slots - 1) / slots;
    const uint32_t used = (per_xcd + chunk - 1) / chunk;                      // non-empty slots
    auto kern = csr_spmv_stream_persistent<T, L, 1, true, TPW, RPT, SKEW>;
    static std::atomic<uint64_t> configured{0};
    hipError_t e = raise_lds_cap(kern, a->device, lds, configured);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(used * 8), dim3(kStreamBlock), lds, st, a->d_rowptr, a->d_colind,
                       a->d_col16, (const T *)a->d_values, (const T *)x, (T *)y, a->d_desc, a->d_pages,
                       (uint32_t)a->nrows, (uint32_t)a->ncols, (uint32_t)a->nnz, p.nblocks, per_xcd, chunk,
                       (uint32_t)(p.nt_store == 1 ? 1 : 0) | ((a->n_ptiles && panel_runs(a, x)) ? 2u : 0u), (uint32_t)p.ring_pages);
    return hipGetLastError();
}

// rows of the tiles the stream kernels skipped (more than 1024 entries in one tile, or a very long row)
template <typename T>
static hipError_t launch_overflow(const spal_csr *a, const void *x, void *y, hipStream_t st, const uint32_t *tiles,
                                  uint32_t ntiles) {
    hipLaunchKernelGGL(csr_spmv_overflow<T>, dim3(ntiles), dim3(kStreamBlock), 0, st, a->d_rowptr,
                       a->d_colind, (const T *)a->d_values, (const T *)x, (T *)y, tiles,
                       ntiles, (uint32_t)std::min(a->plan.rows_per_tile, 64), (uint32_t)a->nrows);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_stream_main(const spal_csr *a, const void *x, void *y, hipStream_t st);

static bool slide_runs(const spal_csr *a, const void *x) {   // (its page loads are 16-byte vectors of x)
    return a->plan.slide && a->plan.slide_on && a->plan.slide_fill_ok && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
}

template <typename T>
static hipError_t launch_stream(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    hipError_t e = launch_stream_main<T>(a, x, y, st);
    if (e == hipSuccess && a->n_ptiles && panel_runs(a, x)) e = launch_panel(a, x, y, st);
    // the sliding kernel keeps the tiles that fit the strip in halves: a shorter list is left over
    if (e == hipSuccess && slide_runs(a, x)) {
        if (a->n_ovtiles_slide) e = launch_overflow<T>(a, x, y, st, a->d_ovtiles_slide, a->n_ovtiles_slide);
    } else if (e == hipSuccess && a->n_ovtiles) {
        e = launch_overflow<T>(a, x, y, st, a->d_ovtiles + 1, a->n_ovtiles);
    }
    return e;
}

template <typename T>
static hipError_t launch_stream_main(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    const CsrPlan &p = a->plan;
    if (slide_runs(a, x)) return launch_slide(a, x, y, st);   // the sliding-window kernel
    if (p.tiles_per_wave == 8) return launch_stream_tpw<T, 8, 64>(a, x, y, st);  // (64-row tiles only)
    // two tiles of loads ahead: instantiated for the plain form of the 64- / 32- / 16-row tiles without skew
    if (p.prefetch == 2 && !p.persistent && !p.skew) {
        if (p.rows_per_tile == 64) return launch_stream_tpw<T, 4, 64, false, 2>(a, x, y, st);
        if (p.rows_per_tile == 32) return launch_stream_tpw<T, 4, 32, false, 2>(a, x, y, st);
        if (p.rows_per_tile == 16) return launch_stream_tpw<T, 4, 16, false, 2>(a, x, y, st);
    }
#define SPAL_STREAM_CASE(RPT, SKEW) \
    case RPT: return p.persistent ? launch_stream_persistent<T, (RPT > 128 ? 1 : RPT > 64 ? 2 : 4), RPT, SKEW>(a, x, y, st) \
                                  : launch_stream_tpw<T, (RPT > 128 ? 1 : RPT > 64 ? 2 : 4), RPT, SKEW>(a, x, y, st);
    if (p.skew) {
        switch (p.rows_per_tile) {
            SPAL_STREAM_CASE(256, true) SPAL_STREAM_CASE(128, true) SPAL_STREAM_CASE(64, true) SPAL_STREAM_CASE(32, true) SPAL_STREAM_CASE(24, true)
            SPAL_STREAM_CASE(16, true) SPAL_STREAM_CASE(12, true) SPAL_STREAM_CASE(8, true)
            default: return hipErrorInvalidValue;
        }
    }
    switch (p.rows_per_tile) {
        SPAL_STREAM_CASE(256, false) SPAL_STREAM_CASE(128, false) SPAL_STREAM_CASE(64, false) SPAL_STREAM_CASE(32, false) SPAL_STREAM_CASE(24, false)
        SPAL_STREAM_CASE(16, false) SPAL_STREAM_CASE(12, false) SPAL_STREAM_CASE(8, false)
        default: return hipErrorInvalidValue;
    }
#undef SPAL_STREAM_CASE
}

template <typename T>
static hipError_t launch_lanes(const spal_csr *a, const void *x, void *y, hipStream_t st) {
    if (__atomic_load_n(&a->plan.cblock, __ATOMIC_ACQUIRE) && a->plan.cblock_on) return launch_cblock(a, x, y, st);   // columns anywhere: csr_cblock.hpp
    if (a->plan.kernel == 2) return launch_stream<T>(a, x, y, st);
    switch (a->plan.lanes_per_row) {
#define SPAL_LANES_CASE(LL) \
    case LL: return launch_vec_unroll<T, LL>(a, x, y, st);
        SPAL_LANES_CASE(2) SPAL_LANES_CASE(4) SPAL_LANES_CASE(8) SPAL_LANES_CASE(16)
        SPAL_LANES_CASE(32) SPAL_LANES_CASE(64)
#undef SPAL_LANES_CASE
        default: return hipErrorInvalidValue;
    }
}

// A handle assembled on the device (spal_coo_assemble_csr) is a complete CSR matrix the moment the assembly returns --
// shape, download, conversions work on its arrays -- but the plan of the PRODUCT kernels (tile heights, x windows, 16-bit
// columns, ...: csr_plan_build, a dozen small kernels and host round trips, ~0.3 ms at config 5) is not part of
// `CsrMatrix::from(&coo)` and is built when something first needs it: the first product, spal_csr_plan, set_option,
// autotune, alloc_vectors or describe.  Under mu_cb, pending until finished, like the column-blocked copy below.
int csr_ensure_plan(spal_csr *a, hipStream_t launch_stream, bool from_launch) {
    if (!__atomic_load_n(&a->plan_pending, __ATOMIC_ACQUIRE)) return SPAL_OK;
    std::lock_guard<std::mutex> lock(a->mu_cb);
    if (!a->plan_pending) return SPAL_OK;
    if (from_launch) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(launch_stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
            return fail(SPAL_ERR_INVALID_ARGUMENT, "the first product of a device-assembled handle plans its kernels and cannot be "
                                                   "captured into a graph: call spal_csr_plan (or run one product) before the capture");
    }
    SPAL_TRY(csr_plan_build(a));
    __atomic_store_n(&a->plan_pending, 0, __ATOMIC_RELEASE);
    return SPAL_OK;
}

int csr_launch(spal_csr *a, const void *x_dev, void *y_dev, hipStream_t stream) {
    SPAL_TRY(csr_ensure_plan(a, stream, true));
    if (!a->parts.empty()) {   // row blocks: each writes its own rows of y
        for (size_t b = 0; b < a->parts.size(); ++b)
            SPAL_TRY(csr_launch(a->parts[b], x_dev, (char *)y_dev + a->part_row0[b] * (uint64_t)a->elem_size, stream));
        return SPAL_OK;
    }
    if (a->bw_on) {   // skewed rows, columns near the rows: the block-window kernel (spal_csr_blockwin.hip)
        SPAL_HIP_TRY(blockwin_launch(a, x_dev, y_dev, stream));
        return SPAL_OK;
    }
    if (a->split_short) {   // row split: the short rows' handle writes every row of y, the long rows are then overwritten
        SPAL_TRY(csr_launch(a->split_short, x_dev, y_dev, stream));
        const uint32_t grid = a->split_nheavy + (a->split_nlong - a->split_nheavy + kStreamWaves - 1) / kStreamWaves;
        if (a->elem_size == 8)
            hipLaunchKernelGGL(csr_spmv_row_list<double>, dim3(grid), dim3(kStreamBlock), 0, stream, a->d_rowptr, a->d_colind,
                               (const double *)a->d_values, (const double *)x_dev, (double *)y_dev, a->d_split_rows, a->split_nlong,
                               a->split_nheavy);
        else
            hipLaunchKernelGGL(csr_spmv_row_list<float>, dim3(grid), dim3(kStreamBlock), 0, stream, a->d_rowptr, a->d_colind,
                               (const float *)a->d_values, (const float *)x_dev, (float *)y_dev, a->d_split_rows, a->split_nlong,
                               a->split_nheavy);
        SPAL_HIP_TRY(hipGetLastError());
        return SPAL_OK;
    }
    if (a->nnz == 0) {
        const uint64_t n = a->nrows;
        if (a->elem_size == 8)
            hipLaunchKernelGGL(fill_zero<double>, dim3((n + 255) / 256), dim3(256), 0, stream,
                               (double *)y_dev, n);
        else
            hipLaunchKernelGGL(fill_zero<float>, dim3((n + 255) / 256), dim3(256), 0, stream,
                               (float *)y_dev, n);
        SPAL_HIP_TRY(hipGetLastError());
        return SPAL_OK;
    }
    if (__atomic_load_n(&a->plan.cblock_pending, __ATOMIC_ACQUIRE)) {
        // a handle assembled on the device whose columns are anywhere: the tiled copy, built once by the first product.
        // `cblock_pending` stays set until the build has FINISHED, so every concurrent caller takes the lock and waits
        // for it (spal.h: products on one handle may run concurrently); the builder publishes plan.cblock last.
        std::lock_guard<std::mutex> lock(a->mu_cb);
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (a->plan.cblock_pending && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
            (void)cblock_plan(a, false);    // a failure means "does not qualify": the stream kernels run
            __atomic_store_n(&a->plan.cblock_pending, 0, __ATOMIC_RELEASE);
        }
    }
    hipError_t e = a->elem_size == 8 ? launch_lanes<double>(a, x_dev, y_dev, stream)
                                     : launch_lanes<float>(a, x_dev, y_dev, stream);
    if (e != hipSuccess)
        return fail(SPAL_ERR_HIP, "csr spmv launch failed: %s", hipGetErrorString(e));
    return SPAL_OK;
}

// per-row-block column windows for block size R -> host vector {cmin, cmax + 1}
// ({0xffffffff, 0} for a block without entries).  The device pass runs once per
// matrix at 256-row granularity; every candidate R that is a multiple of 256 is
// derived from it on the host.
static constexpr uint32_t kWinBase = 256;

static int block_windows_device(spal_csr *a, uint32_t R, std::vector<uint2> &win) {
    const uint32_t nb = (uint32_t)((a->nrows + R - 1) / R);
    uint2 *d_win = nullptr;
    SPAL_HIP_TRY(dev_alloc((void **)&d_win, (size_t)nb * sizeof(uint2)));
    hipLaunchKernelGGL(csr_block_windows, dim3(nb), dim3(256), 0, a->stream, a->d_rowptr,
                       a->d_colind, (uint32_t)a->nrows, R, d_win);
    win.resize(nb);
    hipError_t e = hipMemcpyAsync(win.data(), d_win, (size_t)nb * sizeof(uint2),
                                  hipMemcpyDeviceToHost, a->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(a->stream);
    (void)dev_free(d_win);
    SPAL_HIP_TRY(e);
    return SPAL_OK;
}

static int block_windows(spal_csr *a, uint32_t R, std::vector<uint2> &win) {
    if (R % kWinBase) return block_windows_device(a, R, win);
    if (a->win_base.empty()) SPAL_TRY(block_windows_device(a, kWinBase, a->win_base));
    const uint32_t k = R / kWinBase;
    const uint32_t nb = (uint32_t)((a->nrows + R - 1) / R);
    win.resize(nb);
    for (uint32_t b = 0; b < nb; ++b) {
        uint2 w = make_uint2(0xffffffffu, 0u);
        const size_t j1 = std::min<size_t>((size_t)(b + 1) * k, a->win_base.size());
        for (size_t j = (size_t)b * k; j < j1; ++j) {
            w.x = std::min(w.x, a->win_base[j].x);
            w.y = std::max(w.y, a->win_base[j].y);
        }
        win[b] = w;
    }
    return SPAL_OK;
}

// Stream plan: super-tiles of R rows; returns the fraction of rows
// that can be streamed and fills `desc`.
static int stream_plan(spal_csr *a, uint32_t R, uint32_t rpt, std::vector<uint4> &desc, uint32_t &cap,
                       double &frac, uint32_t **out_pages, uint32_t &n_over, std::vector<uint32_t> &skip,
                       double &cost, bool decide_skew, std::vector<uint2> &panel_win) {
    *out_pages = nullptr;
    n_over = 0;
    const uint32_t nb = (uint32_t)((a->nrows + R - 1) / R);
    // pages of 256 columns that fit the LDS budget: 24 (f64) / 48 (f32)
    // page budgets: `small` keeps two workgroups per CU, `page_cap` (<= 64: page ids travel in a wave's lanes) one
    const uint32_t page_bytes = kPageCols * (uint32_t)a->elem_size;
    const uint32_t page_cap = std::min<uint32_t>(64u, kStreamBigWindowBytes / page_bytes);
    auto small_pages = [&]() {
        return std::min<uint32_t>(page_cap, a->plan.skew ? (a->elem_size == 4 ? 62u : kStreamWindowBytesSkew / page_bytes)
                                                         : (a->elem_size == 4 ? 64u : kStreamWindowBytes / page_bytes));
    };
    uint32_t small_cap = small_pages();
    // temporaries of the plan: returned to the allocator on every path out of this function
    DevBuf b_pages, b_info, b_ok, b_win;
    SPAL_HIP_TRY(b_ok.alloc((size_t)nb * sizeof(uint4)));
    SPAL_HIP_TRY(b_info.alloc((size_t)nb * sizeof(uint4)));
    SPAL_HIP_TRY(b_pages.alloc((size_t)nb * page_cap * 4));
    uint32_t *d_pages = b_pages.as<uint32_t>();
    uint4 *d_info = b_info.as<uint4>(), *d_ok = b_ok.as<uint4>();
    hipLaunchKernelGGL(csr_stream_check, dim3(nb), dim3(256), 0, a->stream, a->d_rowptr, (uint32_t)a->nrows, R,
                       rpt, (uint32_t)a->plan.stream_row_max, 128u / (uint32_t)a->elem_size, d_ok);
    // column windows already known per 256 rows (e.g. handed over by the assembly): fold and pass them
    uint2 *d_win = nullptr;
    if (!a->win_base.empty() && R % kWinBase == 0) {
        std::vector<uint2> win;
        SPAL_TRY(block_windows(a, R, win));
        SPAL_HIP_TRY(b_win.alloc((size_t)nb * sizeof(uint2)));
        d_win = b_win.as<uint2>();
        SPAL_HIP_TRY(hipMemcpyAsync(d_win, win.data(), (size_t)nb * sizeof(uint2), hipMemcpyHostToDevice, a->stream));
        SPAL_HIP_TRY(hipStreamSynchronize(a->stream));   // `win` goes out of scope
    }
    hipLaunchKernelGGL(csr_block_pages, dim3(nb), dim3(256), 0, a->stream, a->d_rowptr, a->d_colind,
                       (uint32_t)a->nrows, R, page_cap,
                       a->plan.window_pages > 0 ? std::min<uint32_t>(page_cap, (uint32_t)a->plan.window_pages) : small_cap,
                       d_win, d_info, d_pages);
    std::vector<uint4> chk(nb);   // {a bit per tile that does not stream, cost of the tiles, entries, -}
    std::vector<uint4> info(nb);
    SPAL_HIP_TRY(hipMemcpyAsync(chk.data(), d_ok, (size_t)nb * sizeof(uint4), hipMemcpyDeviceToHost, a->stream));
    SPAL_HIP_TRY(hipMemcpyAsync(info.data(), d_info, (size_t)nb * sizeof(uint4), hipMemcpyDeviceToHost, a->stream));
    SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
    (void)dev_free(b_ok.release());
    (void)dev_free(b_info.release());
    (void)dev_free(b_win.release());
    if (decide_skew) {   // skewed product strips when most rows are a multiple of 128 bytes long (16 f64 / 32 f32 entries)
        uint64_t aligned = 0;
        for (uint32_t b = 0; b < nb; ++b) aligned += chk[b].w & 0xffffu;
        a->plan.skew = 2 * aligned > a->nrows ? 1 : 0;
        small_cap = small_pages();   // (the page kernel above ran with the budget of the previous setting: at worst
                                     //  a super-tile of 23 or 24 pages gathers x through L2)
    }
    const uint32_t budget = kStreamWindowBytes / (uint32_t)a->elem_size;
    const uint32_t valign = 16u / (uint32_t)a->elem_size;
    // which page budget?  Rows weighted by what their mode costs per entry, from measurements on bands
    // (f64): LDS window at two workgroups per CU 1.0, at one workgroup per CU 1.35, x through L2 2.3
    uint32_t use_cap = small_cap;
    if (a->plan.window_pages > 0) {
        use_cap = std::min<uint32_t>(page_cap, (uint32_t)a->plan.window_pages);
    } else if (page_cap > small_cap) {
        // per super-tile, in bytes: its entries' stream (10 B each) plus the window's pages (staged through L2,
        // weighted 0.7); the 1.35 and 2.3 are measured on bands.  (A window as large as the entries it serves
        // does not pay: 256-row super-tiles of 10-entry rows spread over 10 000 columns ran 380 us with the
        // large window, 227 us with x through L2.)
        double cost_small = 0, cost_big = 0;
        for (uint32_t b = 0; b < nb; ++b) {
            if (info[b].y == 0) continue;
            const double stream = 10.0 * (double)chk[b].z, gather = 2.3 * stream;
            const double window = 0.7 * (double)page_bytes * (double)info[b].z;
            const bool pageable = info[b].z != kNotPageable;
            cost_small += pageable && info[b].z <= small_cap ? stream + window : gather;
            cost_big += pageable ? 1.35 * (stream + window) : gather;
        }
        if (cost_big < 0.97 * cost_small) use_cap = page_cap;
    }
    desc.assign(nb, make_uint4(0, 0, kModeVectorGlobal, 0));
    panel_win.assign(nb, make_uint2(0u, 0u));
    uint64_t rows_stream = 0;
    cap = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        const uint64_t rows = std::min<uint64_t>(R, a->nrows - (uint64_t)b * R);
        const uint4 w = info[b];   // {first column, one past the last, pages or kNotPageable, contiguous}
        if (w.y == 0) {  // nothing stored: stream mode with one (arbitrary) page writes the zeros
            desc[b] = make_uint4(0, 1, kModeStream, 1);
            cap = std::max(cap, kPageCols);
            rows_stream += rows;
            continue;
        }
        if (w.z != kNotPageable && w.z <= use_cap) {   // the pages its rows touch fit the LDS budget
            desc[b] = make_uint4(w.w ? (w.x >> kPageShift) : b * page_cap, w.z, kModeStream, w.w);
            cap = std::max(cap, w.z * kPageCols);
            rows_stream += rows;
            continue;
        }
        if (a->plan.stream_global) {   // columns too scattered for LDS: x through L2 ...
            desc[b] = make_uint4(0, 0, kModeStreamGlobal, 0);
            // ... unless the column SPAN is a few LDS windows wide (a wide band): then the super-tile is taken in
            // column panels by csr_spmv_panel (csr_panel.hpp), desc.w bit 1
            const uint32_t p_first = w.x >> kPageShift, p_span = ((w.y - 1u) >> kPageShift) - p_first + 1u;
            if (a->plan.panel_pages > 0 && rpt <= 64u && a->plan.tiles_per_wave == 4 && !a->plan.skew &&
                p_span <= std::min<uint32_t>((uint32_t)a->plan.panel_pages, 255u)) {   // (span-relative columns are 16-bit: < 65 536)
                desc[b].w |= 2u;
                panel_win[b] = make_uint2(p_first, p_span);
            }
            rows_stream += rows;
            continue;
        }
        // (stream_global switched off) vector rows, x window in LDS when the span fits
        const uint32_t cb = w.x & ~(valign - 1);
        const uint32_t len = w.y - cb;
        if (len <= budget) {
            desc[b] = make_uint4(cb, len, kModeVectorLds, 0);
            cap = std::max(cap, len);
        }
    }
    // super-tiles whose rows all have the same length: the kernels derive the row bounds and do not read rowptr
    if (a->plan.uniform_rows)
        for (uint32_t b = 0; b < nb; ++b)
            if (desc[b].z == kModeStream || desc[b].z == kModeStreamGlobal) desc[b].y |= (chk[b].w >> 16) << 8;
    // what the caller ranks tile heights by: the share of rows whose TILE streams (the marked tiles of a
    // super-tile in a stream mode are left to csr_spmv_overflow)
    {
        uint64_t rows_tiles = 0;
        skip.assign(nb, 0u);
        cost = 0.0;
        for (uint32_t b = 0; b < nb; ++b) {
            if (desc[b].z != kModeStream && desc[b].z != kModeStreamGlobal) {
                cost += 2.0 * (double)chk[b].z;   // (vector rows inside the stream kernel)
                continue;
            }
            skip[b] = chk[b].x;
            cost += (double)chk[b].y;
            const uint64_t rows = std::min<uint64_t>(R, a->nrows - (uint64_t)b * R);
            const uint32_t over = (uint32_t)__builtin_popcount(skip[b]);
            rows_tiles += rows - std::min<uint64_t>(rows, (uint64_t)over * rpt);
            n_over += over;
        }
        frac = a->nrows ? (double)rows_tiles / (double)a->nrows : 0.0;
    }
    (void)rows_stream;
    *out_pages = (uint32_t *)b_pages.release();
    return SPAL_OK;
}

// Chooses the kernel and its parameters and builds the per-block tables.
// the assembly's per-group column spans -> win_base (per 256 rows), once
static int csr_fetch_group_windows(spal_csr *a) {
    if (!a->d_win_groups) return SPAL_OK;
    std::vector<uint2> g(a->win_groups);
    hipError_t e = hipMemcpy(g.data(), a->d_win_groups, (size_t)a->win_groups * sizeof(uint2), hipMemcpyDeviceToHost);
    (void)dev_free(a->d_win_groups);
    a->d_win_groups = nullptr;
    SPAL_HIP_TRY(e);
    const uint32_t per = 256u >> a->win_group_bits;
    a->win_base.assign(((size_t)a->nrows + 255) / 256, make_uint2(0xffffffffu, 0u));
    for (uint32_t i = 0; i < a->win_groups; ++i) {
        uint2 &w = a->win_base[i / per];
        w.x = std::min(w.x, g[i].x);
        w.y = std::max(w.y, g[i].y);
    }
    return SPAL_OK;
}

static void csr_free(spal_csr *a);

// setup: out = {rows longer than T, 64-row tiles that hold one, their entries (low, high word)} -- what decides whether the
// row split is worth building, before anything is copied to the host
__global__ __launch_bounds__(256) void csr_long_rows_scan(const uint32_t *__restrict__ rowptr, uint32_t nrows, uint32_t T,
                                                          unsigned long long *__restrict__ out) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t len = r < nrows ? rowptr[r + 1] - rowptr[r] : 0u;
    const bool lng = len > T;
    const uint64_t m = __ballot(lng);          // a wave = a 64-row tile
    unsigned long long entries = lng ? len : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) entries += __shfl_xor(entries, o);
    if ((threadIdx.x & 63) == 0 && m) {
        atomicAdd(&out[0], (unsigned long long)__popcll(m));
        atomicAdd(&out[1], 1ull);
        atomicAdd(&out[2], entries);
    }
}

// ROW SPLIT (round 4).  Power-law row lengths: 0.7 % of the rows are longer than the 128 entries a lane
// may sum, but a 64-row tile holds such a row with probability 36 % -- a third of the ROWS went to the overflow kernel, tile
// by tile, for the sake of those few (2M rows, 20M entries, columns within +-5000: 196 us = 0.18).  When long rows keep a
// tenth of the tiles and more from streaming, the handle multiplies as A = A_short + A_long instead: A_short is a compacted
// copy WITHOUT the long rows' entries (they are empty rows in it) with a complete plan of its own -- stream kernels, sliding
// window, column blocks, whatever its structure asks for --, the long rows are listed and taken a wave each out of the
// original arrays (csr_spmv_row_list).  Short rows stay bit-identical to the reference's order; long rows are tree sums, as
// they were in the overflow kernel (1e-10).  Built on the host side from a copy of rowptr (setup time).
static int csr_try_row_split(spal_csr *a, bool *did) {
    CsrPlan &p = a->plan;
    *did = false;
    if (a->split_short) { csr_free(a->split_short); a->split_short = nullptr; }
    (void)dev_free(a->d_split_rows); a->d_split_rows = nullptr;
    a->split_nlong = 0; a->split_long_entries = 0;
    if (p.row_split == 0 || a->split_child || !a->parts.empty() || a->nnz == 0 || a->nrows < 2) return SPAL_OK;
    const uint32_t T = (uint32_t)std::max(1, p.split_threshold);
    {   // on the device first (one pass over rowptr, 24 bytes back): most matrices have no such rows, or too few
        unsigned long long *d_cnt = nullptr, cnt[3] = {0, 0, 0};
        SPAL_HIP_TRY(dev_alloc((void **)&d_cnt, 24));
        hipError_t e = hipMemsetAsync(d_cnt, 0, 24, a->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(csr_long_rows_scan, dim3((uint32_t)((a->nrows + 255) / 256)), dim3(256), 0, a->stream, a->d_rowptr,
                               (uint32_t)a->nrows, T, d_cnt);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(cnt, d_cnt, 24, hipMemcpyDeviceToHost, a->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(a->stream);
        (void)dev_free(d_cnt);
        SPAL_HIP_TRY(e);
        const uint64_t ntiles0 = (a->nrows + 63) / 64, nnz_s0 = a->nnz - cnt[2];
        const bool wanted0 = p.row_split == 1 ? cnt[0] != 0
                                              : (cnt[1] * 10 >= ntiles0 && nnz_s0 >= a->nnz / 4 && (double)nnz_s0 / (double)a->nrows <= 64.0);
        if (!wanted0 || cnt[0] == 0 || nnz_s0 == 0) return SPAL_OK;
    }
    std::vector<uint32_t> rp((size_t)a->nrows + 1);
    SPAL_HIP_TRY(hipMemcpy(rp.data(), a->d_rowptr, rp.size() * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> rows_long;
    uint64_t tiles_hit = 0, long_entries = 0, ntiles = (a->nrows + 63) / 64;
    for (uint64_t t = 0; t < ntiles; ++t) {
        bool hit = false;
        const uint64_t r1 = std::min<uint64_t>(a->nrows, (t + 1) * 64);
        for (uint64_t r = t * 64; r < r1; ++r) {
            const uint32_t len = rp[r + 1] - rp[r];
            if (len > T) { hit = true; rows_long.push_back((uint32_t)r); long_entries += len; }
        }
        tiles_hit += hit ? 1 : 0;
    }
    // longest first: the rows of more than 1024 entries get a workgroup each and start first (a row of 5000 entries by one
    // wave is 20 dependent trips, 80 us -- the launch's tail)
    std::stable_sort(rows_long.begin(), rows_long.end(), [&](uint32_t x, uint32_t y) { return rp[x + 1] - rp[x] > rp[y + 1] - rp[y]; });
    uint32_t n_heavy = 0;
    while (n_heavy < rows_long.size() && rp[rows_long[n_heavy] + 1] - rp[rows_long[n_heavy]] > 1024u) ++n_heavy;
    const uint64_t nnz_s = a->nnz - long_entries;
    const bool wanted = p.row_split == 1 ? !rows_long.empty()
                                         : (tiles_hit * 10 >= ntiles && nnz_s >= a->nnz / 4 &&       // a tenth of the tiles poisoned; the short part is worth a plan
                                            (double)nnz_s / (double)a->nrows <= 64.0);              // ... and streams
    if (!wanted || rows_long.empty() || nnz_s == 0) return SPAL_OK;
    // the short part's arrays
    std::vector<uint32_t> rps((size_t)a->nrows + 1);
    uint32_t run = 0;
    for (uint64_t r = 0; r < a->nrows; ++r) {
        rps[r] = run;
        const uint32_t len = rp[r + 1] - rp[r];
        if (len <= T) run += len;
    }
    rps[a->nrows] = run;
    const uint64_t cap = (uint64_t)nnz_s + kStreamPad;
    uint32_t *d_rps = nullptr, *d_cis = nullptr;
    void *d_vas = nullptr;
    hipError_t e = dev_alloc((void **)&d_rps, rps.size() * 4);
    if (e == hipSuccess) e = dev_alloc((void **)&d_cis, cap * 4);
    if (e == hipSuccess) e = dev_alloc(&d_vas, cap * (size_t)a->elem_size);
    std::vector<uint32_t> list(rows_long.size() * 3);   // {row, first entry, one past the last}
    for (size_t i = 0; i < rows_long.size(); ++i) {
        list[3 * i] = rows_long[i]; list[3 * i + 1] = rp[rows_long[i]]; list[3 * i + 2] = rp[rows_long[i] + 1];
    }
    if (e == hipSuccess) e = dev_alloc((void **)&a->d_split_rows, list.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(d_rps, rps.data(), rps.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(a->d_split_rows, list.data(), list.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset((char *)d_cis + (size_t)nnz_s * 4, 0, kStreamPad * 4);
    if (e == hipSuccess) e = hipMemset((char *)d_vas + (size_t)nnz_s * a->elem_size, 0, kStreamPad * (size_t)a->elem_size);
    if (e == hipSuccess) {
        const uint32_t grid = (uint32_t)((a->nrows + 255) / 256);
        if (a->elem_size == 8)
            hipLaunchKernelGGL(csr_split_copy<double>, dim3(grid), dim3(256), 0, a->stream, a->d_rowptr, d_rps, a->d_colind,
                               (const double *)a->d_values, d_cis, (double *)d_vas, (uint32_t)a->nrows);
        else
            hipLaunchKernelGGL(csr_split_copy<float>, dim3(grid), dim3(256), 0, a->stream, a->d_rowptr, d_rps, a->d_colind,
                               (const float *)a->d_values, d_cis, (float *)d_vas, (uint32_t)a->nrows);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(a->stream);
    spal_csr *child = nullptr;
    int st = e == hipSuccess ? SPAL_OK : SPAL_ERR_HIP;
    if (st == SPAL_OK) {
        // (a handle of its own: its plan is built here, eagerly, with this handle's user options that concern the stream kernels)
        st = csr_adopt_device(a->device, a->elem_size, a->nrows, a->ncols, nnz_s, cap, d_rps, d_cis, d_vas, &child, nullptr, true, true);
        if (st == SPAL_OK) {
            child->split_child = 1;
            child->plan.row_split = 0;
            st = csr_ensure_plan(child, nullptr, false);
            if (st != SPAL_OK) { csr_free(child); child = nullptr; d_rps = nullptr; d_cis = nullptr; d_vas = nullptr; }
        }
    }
    if (st != SPAL_OK) {   // an optional form: without it the handle runs the kernels it always ran
        (void)dev_free(d_rps); (void)dev_free(d_cis); (void)dev_free(d_vas);
        (void)dev_free(a->d_split_rows); a->d_split_rows = nullptr;
        (void)hipGetLastError();
        return SPAL_OK;
    }
    // The short part of a skewed matrix is ragged short rows; where its super-tiles are too wide for an LDS window (they went
    // to the column panels) the column-blocked kernels may be the faster family (power-law rows, columns within +-5000: 93 us
    // in panels, 73 - 82 us column-blocked) -- or not (uniform rows in a band of 16 384: panels).  Setup time: both are timed
    // on scratch vectors, the faster stays.
    if (child->plan.kernel == 2 && child->n_ptiles * 2u > child->plan.nblocks && !child->plan.cblock && child->plan.cblock_user < 0) {
        (void)cblock_plan(child, true);
        if (child->plan.cblock) {
            void *sx = nullptr, *sy = nullptr;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            hipError_t te = dev_alloc(&sx, std::max<uint64_t>(a->ncols, 1) * (size_t)a->elem_size);
            if (te == hipSuccess) te = dev_alloc(&sy, std::max<uint64_t>(a->nrows, 1) * (size_t)a->elem_size);
            if (te == hipSuccess) te = hipMemsetAsync(sx, 0, a->ncols * (size_t)a->elem_size, child->stream);
            if (te == hipSuccess) te = hipEventCreate(&e0);
            if (te == hipSuccess) te = hipEventCreate(&e1);
            float ms[2] = {0.f, 0.f};
            int trc = SPAL_OK;
            for (int on = 0; on < 2 && te == hipSuccess && trc == SPAL_OK; ++on) {
                child->plan.cblock_on = on;
                for (int i = 0; i < 2 && trc == SPAL_OK; ++i) trc = csr_launch(child, sx, sy, child->stream);
                te = hipEventRecord(e0, child->stream);
                for (int i = 0; i < 5 && trc == SPAL_OK; ++i) trc = csr_launch(child, sx, sy, child->stream);
                if (te == hipSuccess) te = hipEventRecord(e1, child->stream);
                if (te == hipSuccess) te = hipEventSynchronize(e1);
                if (te == hipSuccess) te = hipEventElapsedTime(&ms[on], e0, e1);
            }
            const bool keep = te == hipSuccess && trc == SPAL_OK && ms[1] < ms[0];
            child->plan.cblock_on = 1;
            if (!keep) cblock_free(child);
            child->cblock_us[0] = ms[0] * 200.f; child->cblock_us[1] = ms[1] * 200.f;   // (us per launch: 5 launches)
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            (void)dev_free(sx); (void)dev_free(sy);
            (void)hipGetLastError();
        }
    }
    a->split_short = child;
    a->split_nheavy = n_heavy;
    a->split_nlong = (uint32_t)rows_long.size();
    a->split_long_entries = long_entries;
    *did = true;
    return SPAL_OK;
}

// Skewed matrices whose columns stay near their rows: the block-window kernel against the row split just built, both timed on
// scratch vectors (setup time); the faster form stays, the other is freed.
static int csr_blockwin_or_split(spal_csr *a) {
    if (blockwin_plan(a) != SPAL_OK || !a->bw_rows) { blockwin_free(a); (void)hipGetLastError(); return SPAL_OK; }   // (optional form)
    void *sx = nullptr, *sy = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t te = dev_alloc(&sx, std::max<uint64_t>(a->ncols, 2) * (size_t)a->elem_size);
    if (te == hipSuccess) te = dev_alloc(&sy, std::max<uint64_t>(a->nrows, 1) * (size_t)a->elem_size);
    if (te == hipSuccess) te = hipMemsetAsync(sx, 0, a->ncols * (size_t)a->elem_size, a->stream);
    if (te == hipSuccess) te = hipEventCreate(&e0);
    if (te == hipSuccess) te = hipEventCreate(&e1);
    float ms[2] = {0.f, 0.f};
    int trc = SPAL_OK;
    for (int on = 0; on < 2 && te == hipSuccess && trc == SPAL_OK; ++on) {
        a->bw_on = on;
        for (int i = 0; i < 2 && trc == SPAL_OK; ++i) trc = csr_launch(a, sx, sy, a->stream);
        te = hipEventRecord(e0, a->stream);
        for (int i = 0; i < 5 && trc == SPAL_OK; ++i) trc = csr_launch(a, sx, sy, a->stream);
        if (te == hipSuccess) te = hipEventRecord(e1, a->stream);
        if (te == hipSuccess) te = hipEventSynchronize(e1);
        if (te == hipSuccess) te = hipEventElapsedTime(&ms[on], e0, e1);
    }
    const bool keep = te == hipSuccess && trc == SPAL_OK && ms[1] < ms[0];
    const float us0 = ms[0] * 200.f, us1 = ms[1] * 200.f;   // (us per product: 5 launches)
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)dev_free(sx); (void)dev_free(sy);
    (void)hipGetLastError();
    if (keep) {
        a->bw_on = 1;
        if (a->split_short) { csr_free(a->split_short); a->split_short = nullptr; }
        (void)dev_free(a->d_split_rows); a->d_split_rows = nullptr;
        a->split_nlong = 0; a->split_nheavy = 0; a->split_long_entries = 0;
        a->plan.kernel = 4;
    } else {
        blockwin_free(a);
    }
    a->bw_us[0] = us0; a->bw_us[1] = us1;
    return SPAL_OK;
}

int csr_plan_build(spal_csr *a) {
    SPAL_TRY(csr_fetch_group_windows(a));
    blockwin_free(a);
    a->bw_us[0] = a->bw_us[1] = 0.f;
    if (a->plan.blockwin == 1 && !a->split_child) {   // asked for by name (tests): whenever the windows fit
        if (a->split_short) { csr_free(a->split_short); a->split_short = nullptr; }
        (void)dev_free(a->d_split_rows); a->d_split_rows = nullptr;
        a->split_nlong = 0;
        if (blockwin_plan(a) != SPAL_OK) { blockwin_free(a); (void)hipGetLastError(); }   // (an optional form: without it the plan below)
        if (a->bw_rows) { a->bw_on = 1; a->plan.kernel = 4; return SPAL_OK; }
    }
    {
        bool did = false;
        SPAL_TRY(csr_try_row_split(a, &did));
        if (did) {   // ("split": the products run through the short part's handle -- unless the block-window kernel beats it)
            a->plan.kernel = 3;
            if (a->plan.blockwin != 0) SPAL_TRY(csr_blockwin_or_split(a));
            return SPAL_OK;
        }
    }
    CsrPlan &p = a->plan;
    const double mean = a->nrows ? (double)a->nnz / (double)a->nrows : 0.0;
    // vector kernel geometry, from measurements (tools/lab.py ab): one lane per entry
    // up to 64 entries per row; longer rows loop in batches of 4 L entries per
    // lane group, which 16 lanes per row keep busiest (128/row: 65 %, L = 64: 38 %)
    // rows longer than a wave (tools/lab.py longrows, 70 ... 1500 entries per row): a whole wave per row,
    // one row group in flight, and few rows per workgroup (below) beat 16 lanes per row everywhere
    // (100/row 56 % against 37 %, 400/row 67 % against 17 %, 1500/row 54 % against 23 %)
    if (!p.user_lanes) p.lanes_per_row = mean > 85.0 ? 64 : mean > 64.0 ? 32 : pick_lanes(mean);
    p.long_rows = mean > 64.0 ? 1 : 0;   // (only the 16-lane instantiation has the batched rest-of-row loop)
    if (!p.user_unroll) p.unroll = mean > 64.0 ? 1 : 4;
    if (!p.user_threads) p.threads = 1024;
    if (a->d_desc) {
        SPAL_HIP_TRY(dev_free(a->d_desc));
        a->d_desc = nullptr;
    }
    p.stream_row_fraction = 0.0;
    p.vec_col16 = 0;
    p.slide = 0;
    p.ring_pages = 0;
    if (a->nnz == 0) {
        cblock_free(a);
        p.kernel = 1;
        p.rows_per_block = 1024;
        p.nblocks = (uint32_t)((a->nrows + 1023) / 1024);
        p.lds_x = 0;
        return SPAL_OK;
    }
    const uint32_t valign = 16u / (uint32_t)a->elem_size;

    // ---- stream kernel: rows short enough that 64 / 32 / 24 / 16 / 12 / 8 of them fit a tile (auto: at least half
    // the rows in tiles that stream; fuller strips pay: 33/row 24 rows per tile 100 us vs 16 rows 109 us, 70/row
    // 12 rows 143 us vs 8 rows 159 us, 81/row 155 vs 187 us).  Measured against the vector kernel on bands (tools/lab.py rpt8): 54/row 82 % vs
    // 51 %, 63/row 84 % vs 47 %, 64/row 80 % (skewed strips) vs 50 %, 81/row 67 % vs 46 %, 100/row 71 % vs 54 %,
    // 120/row 68 % vs 56 %; 4-row tiles for 150 ... 250/row were level with or behind the vector kernel.
    if ((p.user_kernel == 0 && mean <= 120.0) || p.user_kernel == 2) {
        if (p.tiles_per_wave != 4 && p.tiles_per_wave != 8) p.tiles_per_wave = 4;
        const int rpt_all[] = {256, 128, 64, 32, 24, 16, 12, 8};   // (48 rows per tile measured behind 32: 20/row 124 vs 111 us)
        std::vector<int> rpts;
        if (p.user_rows_per_tile) rpts.push_back(p.rows_per_tile);
        else if (p.tiles_per_wave == 8) rpts.push_back(64);
        else rpts.assign(rpt_all + (mean <= 4.0 ? 0 : mean <= 8.0 ? 1 : 2), rpt_all + 8);   // (256 / 128 rows of more than 4 / 8 entries do not fit a tile)
        std::vector<uint4> desc, best_desc;
        uint32_t cap = 0, best_cap = 0;
        double frac = 0.0, best_frac = -1.0, best_cost = -1.0;
        int best_rpt = rpts[0];
        uint32_t *best_pages = nullptr;
        uint32_t n_over = 0, best_over = 0;
        std::vector<uint32_t> skip, best_skip;
        std::vector<uint2> pwin, best_pwin;
        if (a->d_pages) { SPAL_HIP_TRY(dev_free(a->d_pages)); a->d_pages = nullptr; }
        if (a->d_ovtiles) { SPAL_HIP_TRY(dev_free(a->d_ovtiles)); a->d_ovtiles = nullptr; }
        a->n_ovtiles = 0;
        for (int rpt : rpts) {
            const uint32_t R = (uint32_t)stream_rows(rpt > 128 ? 1 : rpt > 64 ? 2 : p.tiles_per_wave, rpt);   // (128 / 256-row tiles: two / one per wave, the same 1024 rows)
            uint32_t *pg = nullptr;
            double cost = 0.0;
            int st = stream_plan(a, R, (uint32_t)rpt, desc, cap, frac, &pg, n_over, skip, cost, !p.user_skew && rpt == rpts[0], pwin);
            if (st != SPAL_OK) { (void)dev_free(best_pages); return st; }
            if (best_cost < 0.0 || cost < 0.95 * best_cost) {  // a narrower tile must be estimated cheaper (see csr_stream_check)
                best_cost = cost;
                best_frac = frac; best_rpt = rpt; best_cap = cap; best_over = n_over; best_desc.swap(desc); best_skip.swap(skip);
                best_pwin.swap(pwin);
                (void)dev_free(best_pages);
                best_pages = pg;
            } else {
                (void)dev_free(pg);
            }
            if (best_cost <= 1.1 * (double)a->nnz) break;   // no tile height costs less than one per entry
        }
        if (!(p.user_kernel == 2 || best_frac >= 0.5)) (void)dev_free(best_pages);
        if (p.user_kernel == 2 || best_frac >= 0.5) {
            a->d_pages = best_pages;
            const uint32_t R = (uint32_t)stream_rows(best_rpt > 128 ? 1 : best_rpt > 64 ? 2 : p.tiles_per_wave, best_rpt);
            p.kernel = 2;
            p.rows_per_tile = best_rpt;
            p.rows_per_block = (int)R;
            p.threads = kStreamBlock;
            p.nblocks = (uint32_t)best_desc.size();
            p.lds_x = best_cap > 0;
            p.lds_entries = (std::max(best_cap, valign) + valign - 1) & ~(valign - 1);
            // one workgroup per CU (the large page budget): nothing else on the CU hides a workgroup's
            // cold start, the persistent form does (band of 8192 columns: 400 vs 460 us)
            if (!p.user_persistent)
                p.persistent = ((size_t)kStreamWaves * (p.skew ? stream_strip<true>() : stream_strip<false>()) + p.lds_entries) * a->elem_size > 80u * 1024u ? 1 : 0;
            p.stream_row_fraction = best_frac;
            uint64_t lds_rows = 0;
            for (uint32_t b = 0; b < p.nblocks; ++b)
                if (best_desc[b].z == kModeVectorLds || best_desc[b].z == kModeStream)
                    lds_rows += std::min<uint64_t>(R, a->nrows - (uint64_t)b * R);
            p.lds_row_fraction = (double)lds_rows / (double)a->nrows;
            uint64_t uni_rows = 0;
            for (uint32_t b = 0; b < p.nblocks; ++b)
                if ((best_desc[b].z == kModeStream || best_desc[b].z == kModeStreamGlobal) && (best_desc[b].y >> 8))
                    uni_rows += std::min<uint64_t>(R, a->nrows - (uint64_t)b * R);
            p.uniform_row_fraction = (double)uni_rows / (double)a->nrows;
            SPAL_HIP_TRY(dev_alloc((void **)&a->d_desc, (size_t)p.nblocks * sizeof(uint4)));
            std::vector<uint4> packed(best_desc);   // + the tiles to skip (see desc_skip_bits)
            for (uint32_t b = 0; b < p.nblocks; ++b) {
                packed[b].w |= (best_skip[b] & 0xffffu) << 16;
                packed[b].z |= best_skip[b] & 0xffff0000u;
            }
            SPAL_HIP_TRY(hipMemcpyAsync(a->d_desc, packed.data(), (size_t)p.nblocks * sizeof(uint4),
                                        hipMemcpyHostToDevice, a->stream));
            SPAL_HIP_TRY(hipStreamSynchronize(a->stream));   // `packed` goes out of scope
            // wide bands: the super-tiles csr_spmv_panel takes in column panels
            {
                if (a->d_ptiles) { SPAL_HIP_TRY(dev_free(a->d_ptiles)); a->d_ptiles = nullptr; }
                if (a->d_pwin) { SPAL_HIP_TRY(dev_free(a->d_pwin)); a->d_pwin = nullptr; }
                std::vector<uint32_t> ids;
                std::vector<uint2> wins;
                for (uint32_t b = 0; b < p.nblocks; ++b)
                    if (best_desc[b].z == kModeStreamGlobal && (best_desc[b].w & 2u)) { ids.push_back(b); wins.push_back(best_pwin[b]); }
                a->n_ptiles = (uint32_t)ids.size();
                if (a->n_ptiles) {
                    SPAL_HIP_TRY(dev_alloc((void **)&a->d_ptiles, ids.size() * sizeof(uint32_t)));
                    SPAL_HIP_TRY(dev_alloc((void **)&a->d_pwin, wins.size() * sizeof(uint2)));
                    SPAL_HIP_TRY(hipMemcpyAsync(a->d_ptiles, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice, a->stream));
                    SPAL_HIP_TRY(hipMemcpyAsync(a->d_pwin, wins.data(), wins.size() * sizeof(uint2), hipMemcpyHostToDevice, a->stream));
                    SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
                    // LDS: the panel window beside the strips (the one-super-tile kernels of this plan need none for these)
                    // a panel: 80 KB of LDS (two workgroups per CU), shared with the product strips; option
                    // "panel_window" sets it in pages (up to 156 KB: one workgroup per CU, fewer passes over the entries)
                    const uint32_t page_b = kPageCols * (uint32_t)a->elem_size;
                    uint32_t pp = (80u * 1024u) / page_b;
                    if (p.panel_window_user > 0) pp = std::min<uint32_t>((uint32_t)p.panel_window_user, (156u * 1024u) / page_b);
                    p.panel_window_pages = (int)pp;
                }
            }
            // bands and the like: the sliding-window kernel (csr_slide.hpp) and its ring-addressed window
            SPAL_TRY(slide_plan(a, R, (uint32_t)best_rpt, best_desc, best_skip, best_cap / kPageCols));
            if (p.ring_pages) p.lds_entries = (uint32_t)p.ring_pages * kPageCols;
            // 16-bit columns only where some super-tile reads them (a matrix whose columns are scattered
            // everywhere streams with the 32-bit ones: no 2 B/entry array to allocate and clear)
            bool any_stream = false;
            for (uint32_t b = 0; b < p.nblocks && !any_stream; ++b) any_stream = best_desc[b].z == kModeStream;
            if (any_stream || a->n_ptiles) {
                if (!a->d_col16) {
                    SPAL_HIP_TRY(dev_alloc((void **)&a->d_col16, (size_t)a->cap_entries * sizeof(uint16_t)));
                    SPAL_HIP_TRY(hipMemsetAsync(a->d_col16, 0, (size_t)a->cap_entries * sizeof(uint16_t), a->stream));
                }
                if (any_stream)
                    hipLaunchKernelGGL(csr_encode_col16, dim3(p.nblocks), dim3(256), 0, a->stream, a->d_rowptr,
                                       a->d_colind, a->d_desc, a->d_pages, a->d_col16, (uint32_t)a->nrows, R,
                                       (uint32_t)p.ring_pages);
                if (a->n_ptiles)   // the column-panel kernel's super-tiles: columns relative to the span's first column
                    hipLaunchKernelGGL(csr_encode_col16_span, dim3(a->n_ptiles), dim3(256), 0, a->stream, a->d_rowptr,
                                       a->d_colind, a->d_ptiles, a->d_pwin, a->d_col16, (uint32_t)a->nrows, R);
                SPAL_HIP_TRY(hipGetLastError());
            }
            if (best_over) {   // the tiles the stream kernels skip: listed for csr_spmv_overflow
                uint32_t *d_list = nullptr;   // [count][first rows]
                const uint32_t pieces = (uint32_t)std::max(1, best_rpt / 64) * best_over;   // (at most)
                SPAL_HIP_TRY(dev_alloc((void **)&d_list, ((size_t)pieces + 1) * 4));
                a->d_ovtiles = d_list;
                SPAL_HIP_TRY(hipMemsetAsync(d_list, 0, 4, a->stream));
                const uint64_t ntile = (a->nrows + (uint64_t)best_rpt - 1) / (uint64_t)best_rpt;
                hipLaunchKernelGGL(csr_overflow_tiles, dim3((uint32_t)((ntile + 255) / 256)), dim3(256), 0, a->stream,
                                   a->d_desc, (uint32_t)a->nrows, R, (uint32_t)best_rpt, pieces, d_list,
                                   d_list + 1);
                SPAL_HIP_TRY(hipGetLastError());
                uint32_t listed = 0;
                SPAL_HIP_TRY(hipMemcpyAsync(&listed, d_list, 4, hipMemcpyDeviceToHost, a->stream));
                SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
                if (listed > pieces || listed < best_over)   // (cannot happen: both count the same tiles)
                    return fail(SPAL_ERR_HIP, "csr plan: %u tiles listed for the overflow kernel, %u counted", listed, best_over);
                a->n_ovtiles = listed;
            }
            SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
            // columns anywhere (most rows sit in super-tiles that gather x from global memory over a span no panel
            // holds): the column-blocked kernel and its tiled copy of the matrix (csr_cblock.hpp)
            {
                uint64_t far_rows = 0;
                for (uint32_t b = 0; b < p.nblocks; ++b)
                    if (best_desc[b].z == kModeStreamGlobal && !(best_desc[b].w & 2u))
                        far_rows += std::min<uint64_t>(R, a->nrows - (uint64_t)b * R);
                p.nonlocal_row_fraction = (double)far_rows / (double)a->nrows;
                p.cblock_pending = 0;
                if (getenv("SPAL_CBLOCK_DEBUG"))
                    fprintf(stderr, "[spal cblock] plan: nonlocal rows %.3f, user %d, lazy %d\n", p.nonlocal_row_fraction, p.cblock_user, a->cblock_lazy);
                if (p.cblock_user == 1 || (p.cblock_user < 0 && p.nonlocal_row_fraction >= 0.5)) {
                    if (a->cblock_lazy && p.cblock_user < 0) { cblock_free(a); p.cblock_pending = 1; }   // built by the first product
                    else (void)cblock_plan(a, p.cblock_user == 1);   // (a failure: the stream kernels run, `cblock_failed`)
                } else {
                    cblock_free(a);
                }
            }
            return SPAL_OK;
        }
    }
    cblock_free(a);

    // ---- vector kernel
    p.kernel = 1;
    if (!p.user_threads && p.threads != 512 && p.threads != 1024) p.threads = 1024;
    if (p.threads != 512 && p.threads != 1024) p.threads = 1024;
    const uint32_t budget = kLdsBudgetBytes / (uint32_t)a->elem_size;  // elements
    const uint32_t cand_all[] = {4096, 2048, 1024, 512};
    std::vector<uint32_t> cands;
    if (p.user_rows_per_block) {
        cands.push_back((uint32_t)p.rows_per_block);
    } else if (mean > 64.0) {
        // long rows: about 100 000 entries per workgroup (1024 rows at 100/row ... 64 rows at 1500/row), so
        // that there are workgroups enough for 256 CUs -- 4096 rows of 400 entries were 61 workgroups
        uint32_t r0 = 1024;
        while (r0 > 64 && (double)r0 * mean > 131072.0) r0 >>= 1;
        // ... and a number of workgroups that fills whole rounds of the 512 the device holds at once (two of 1024
        // threads per CU): these launches are two or three rounds long, and 1250 workgroups (2.44 rounds) ran at
        // 60 ... 67 % where 980 (1.9 rounds) ran at 73 ... 78 % (tools/lab.py longrows threads).  R need not be
        // a power of two.
        if (p.threads == 1024) {
            const double want = std::max(1.0, (double)a->nnz / 100000.0);              // workgroups of ~100 000 entries
            const uint64_t rounds = std::max<uint64_t>(1, (uint64_t)(want / 512.0 + 0.5));
            // (98 % of the slots, rounded down by taking R up to a multiple of 16: a launch planned to the last slot
            // spills into one more round -- 200 entries per row: 947 ... 977 workgroups 213 ... 218 us, 1009 of them
            // 232 ... 240 us -- and an R that gives some waves one row more than others costs as much: 1500 per
            // row, R = 64: 162 ... 168 us, R = 68 or 72: 174 us)
            const uint64_t nb = rounds * 502;
            uint64_t R = (a->nrows + nb - 1) / nb;
            R = std::min<uint64_t>(4096, std::max<uint64_t>(16, (R + 15) / 16 * 16));   // 16 waves, the same number of rows each
            cands.push_back((uint32_t)R);
        }
        cands.push_back(r0);
        if (r0 > 64) cands.push_back(r0 >> 1);
    } else {
        cands.assign(cand_all, cand_all + 4);
    }

    std::vector<uint4> best_desc;
    uint32_t best_R = 0, best_cap = 0;
    double best_frac = -1.0;
    const bool want_lds = p.user_lds ? p.lds_x != 0 : true;
    for (uint32_t R : cands) {
        std::vector<uint2> win;
        SPAL_TRY(block_windows(a, R, win));
        const uint32_t nb = (uint32_t)win.size();
        std::vector<uint4> desc(nb, make_uint4(0, 0, kModeVectorGlobal, 0));
        uint64_t fit_rows = 0;
        uint32_t cap = 0;
        for (uint32_t b = 0; b < nb; ++b) {
            const uint2 w = win[b];
            const uint64_t rows = std::min<uint64_t>(R, a->nrows - (uint64_t)b * R);
            if (w.y == 0) {  // block stores nothing: no window needed
                fit_rows += rows;
                continue;
            }
            const uint32_t cb = w.x & ~(valign - 1);
            const uint32_t len = w.y - cb;
            if (want_lds && len <= budget) {
                desc[b] = make_uint4(cb, len, kModeVectorLds, 0);
                cap = std::max(cap, len);
                fit_rows += rows;
            }
        }
        const double frac = (double)fit_rows / (double)a->nrows;
        // prefer the largest R whose blocks (nearly) all fit; otherwise the best coverage
        const bool good = frac >= 0.9;
        if (best_R == 0 || (good && best_frac < 0.9) || (!good && best_frac < 0.9 && frac > best_frac)) {
            best_R = R; best_frac = frac; best_cap = cap; best_desc.swap(desc);
        }
        if (good) break;
    }
    p.rows_per_block = (int)best_R;
    p.nblocks = (uint32_t)((a->nrows + best_R - 1) / best_R);
    p.lds_row_fraction = best_frac;
    // LDS only pays when most rows can use it; otherwise run without the
    // allocation so more workgroups fit per CU.
    const bool use_lds = want_lds && best_cap > 0 && (p.user_lds || best_frac >= 0.5);
    p.lds_x = use_lds ? 1 : 0;
    p.lds_entries = use_lds ? ((best_cap + valign - 1) & ~(valign - 1)) : 0;
    if (!use_lds)
        for (auto &d : best_desc) d = make_uint4(0, 0, kModeVectorGlobal, 0);
    SPAL_HIP_TRY(dev_alloc((void **)&a->d_desc, (size_t)p.nblocks * sizeof(uint4)));
    SPAL_HIP_TRY(hipMemcpy(a->d_desc, best_desc.data(), (size_t)p.nblocks * sizeof(uint4),
                           hipMemcpyHostToDevice));
    // long rows with LDS windows (at most 64 KiB: 16 bits address them): 2-byte columns for those blocks
    p.vec_col16 = (p.vec_col16_allowed && use_lds && p.long_rows && p.unroll == 1 &&
                   (p.lanes_per_row == 64 || p.lanes_per_row == 32) && best_cap <= 65536u) ? 1 : 0;
    if (p.vec_col16) {
        if (!a->d_col16) {
            SPAL_HIP_TRY(dev_alloc((void **)&a->d_col16, (size_t)a->cap_entries * sizeof(uint16_t)));
            SPAL_HIP_TRY(hipMemsetAsync(a->d_col16, 0, (size_t)a->cap_entries * sizeof(uint16_t), a->stream));
        }
        hipLaunchKernelGGL(csr_encode_col16_window, dim3(p.nblocks), dim3(256), 0, a->stream, a->d_rowptr,
                           a->d_colind, a->d_desc, a->d_col16, (uint32_t)a->nrows, best_R);
        SPAL_HIP_TRY(hipGetLastError());
        SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
    }
    return SPAL_OK;
}

static void csr_free(spal_csr *a) {
    if (!a) return;
    for (spal_csr *part : a->parts) csr_free(part);
    (void)dev_free(a->d_rowptr);
    (void)dev_free(a->d_colind);
    (void)dev_free(a->d_values);
    (void)dev_free(a->d_desc);
    if (a->col16_placed) place_free(a->device, a->d_col16);
    else (void)dev_free(a->d_col16);
    (void)dev_free(a->d_pages);
    (void)dev_free(a->d_ovtiles);
    (void)dev_free(a->d_sdesc);
    (void)dev_free(a->d_ovtiles_slide);
    (void)dev_free(a->d_ptiles);
    (void)dev_free(a->d_pwin);
    if (a->split_short) csr_free(a->split_short);
    (void)dev_free(a->d_split_rows);
    blockwin_free(a);
    if (a->d_vec_block && a->vec_block_owned) (void)hipFree(a->d_vec_block);
    else place_free(a->device, a->d_vec_block);
    (void)dev_free(a->d_win_groups);
    cblock_free(a);
    (void)dev_free(a->d_x);
    (void)dev_free(a->d_y);
    stream_release(a->stream);
    delete a;
}

int csr_adopt_device(int device, int elem_size, uint64_t nrows, uint64_t ncols, uint64_t nnz,
                     uint64_t cap_entries, uint32_t *d_rowptr, uint32_t *d_colind, void *d_values,
                     spal_csr **out, const std::vector<uint2> *win256, bool eager_copies, bool lazy_plan,
                     uint2 *d_win_groups, uint32_t win_groups, uint32_t win_group_bits) {
    spal_csr *a = new spal_csr;
    if (win256 && win256->size() == (nrows + kWinBase - 1) / kWinBase) a->win_base = *win256;
    if (d_win_groups && win_group_bits <= 8 && win_groups == (uint32_t)((nrows + (1ull << win_group_bits) - 1) >> win_group_bits)) {
        a->d_win_groups = d_win_groups; a->win_groups = win_groups; a->win_group_bits = win_group_bits;
    } else if (d_win_groups) {
        (void)dev_free(d_win_groups);
    }
    a->device = device;
    a->elem_size = elem_size;
    a->nrows = nrows; a->ncols = ncols; a->nnz = nnz;
    a->d_rowptr = d_rowptr; a->d_colind = d_colind; a->d_values = d_values;
    a->cap_entries = cap_entries;
    a->cblock_lazy = eager_copies ? 0 : 1;   // (a handle created from host arrays pays for its second copy at create time)
    auto bail = [&](int st) {
        // the caller keeps ownership of the arrays it passed in on failure
        if (a->d_rowptr == d_rowptr) a->d_rowptr = nullptr;
        if (a->d_colind == d_colind) a->d_colind = nullptr;
        if (a->d_values == d_values) a->d_values = nullptr;
        csr_free(a);
        return st;
    };
    hipError_t e = stream_acquire(&a->stream);
    if (e != hipSuccess) return bail(fail(SPAL_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)));
    // the stream kernel reads whole 128-entry steps: keep kStreamPad spare entries
    const uint64_t need = nnz + kStreamPad;
    if (cap_entries < need) {
        uint32_t *ci = nullptr;
        void *va = nullptr;
        e = dev_alloc((void **)&ci, need * sizeof(uint32_t));
        if (e == hipSuccess) e = dev_alloc((void **)&va, need * (size_t)elem_size);
        if (e == hipSuccess) e = hipMemsetAsync(ci, 0, need * sizeof(uint32_t), a->stream);
        if (e == hipSuccess) e = hipMemsetAsync(va, 0, need * (size_t)elem_size, a->stream);
        if (e == hipSuccess && nnz) e = hipMemcpyAsync(ci, d_colind, nnz * sizeof(uint32_t), hipMemcpyDeviceToDevice, a->stream);
        if (e == hipSuccess && nnz) e = hipMemcpyAsync(va, d_values, nnz * (size_t)elem_size, hipMemcpyDeviceToDevice, a->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(a->stream);
        if (e != hipSuccess) {
            (void)dev_free(ci); (void)dev_free(va);
            return bail(fail(e == hipErrorOutOfMemory ? SPAL_ERR_OUT_OF_MEMORY : SPAL_ERR_HIP,
                             "csr_adopt_device: %s", hipGetErrorString(e)));
        }
        a->d_colind = ci; a->d_values = va; a->cap_entries = need;
    }
    int st = SPAL_OK;
    if (lazy_plan && nnz != 0) a->plan_pending = 1;   // (csr_ensure_plan)
    else st = csr_plan_build(a);
    if (st != SPAL_OK) {
        const bool swapped = a->d_colind != d_colind;
        if (swapped) { (void)dev_free(a->d_colind); (void)dev_free(a->d_values); a->d_colind = d_colind; a->d_values = d_values; }
        return bail(st);
    }
    if (a->d_colind != d_colind) {  // the padded copies replaced the caller's arrays
        (void)dev_free(d_colind);
        (void)dev_free(d_values);
    }
    *out = a;
    return SPAL_OK;
}

// rows [r0, r1) of validated host arrays -> a handle whose offsets are relative to rowptr[r0]
template <typename T>
static int csr_create_rows(int device, uint64_t r0, uint64_t r1, uint64_t ncols, const uint64_t *rowptr,
                           const uint64_t *colind, const T *values, spal_csr **out) {
    const uint64_t nrows = r1 - r0, e0 = rowptr[r0], nnz = rowptr[r1] - e0;
    // narrow usize -> u32 on the host (threads), then one upload per array
    std::vector<uint32_t> rp32(nrows + 1), ci32(nnz);
    parallel_for(nrows + 1, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b; i < e; ++i) rp32[i] = (uint32_t)(rowptr[r0 + i] - e0);
    });
    parallel_for(nnz, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b; i < e; ++i) ci32[i] = (uint32_t)colind[e0 + i];
    });
    uint32_t *d_rp = nullptr, *d_ci = nullptr;
    void *d_v = nullptr;
    auto cleanup = [&] { (void)dev_free(d_rp); (void)dev_free(d_ci); (void)dev_free(d_v); };
    const uint64_t cap = nnz + kStreamPad;  // spare entries for the stream kernel's whole-step reads
    hipError_t e = dev_alloc((void **)&d_rp, (nrows + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = dev_alloc((void **)&d_ci, cap * sizeof(uint32_t));
    if (e == hipSuccess) e = dev_alloc((void **)&d_v, cap * sizeof(T));
    if (e == hipSuccess) e = hipMemset((char *)d_ci + nnz * sizeof(uint32_t), 0, kStreamPad * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset((char *)d_v + nnz * sizeof(T), 0, kStreamPad * sizeof(T));
    if (e == hipSuccess) e = hipMemcpy(d_rp, rp32.data(), (nrows + 1) * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) e = hipMemcpy(d_ci, ci32.data(), nnz * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) e = hipMemcpy(d_v, values + e0, nnz * sizeof(T), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        cleanup();
        return fail(e == hipErrorOutOfMemory ? SPAL_ERR_OUT_OF_MEMORY : SPAL_ERR_HIP,
                    "spal_csr_create: upload failed: %s", hipGetErrorString(e));
    }
    spal_csr *a = nullptr;
    int st = csr_adopt_device(device, (int)sizeof(T), nrows, ncols, nnz, cap, d_rp, d_ci, d_v, &a, nullptr, true);
    if (st != SPAL_OK) { cleanup(); return st; }
    *out = a;
    return SPAL_OK;
}

// entries one set of 32-bit device offsets addresses; SPAL_CSR_PART_ENTRIES lowers it (tests of the row-block path)
static uint64_t csr_part_entries() {
    const char *s = getenv("SPAL_CSR_PART_ENTRIES");
    const uint64_t u = s ? strtoull(s, nullptr, 10) : 0;
    return u ? std::min<uint64_t>(u, kMaxEntries) : kMaxEntries;
}

template <typename T>
static int csr_create(int device, uint64_t nrows, uint64_t ncols, const uint64_t *rowptr,
                      uint64_t rowptr_len, const uint64_t *colind, uint64_t colind_len,
                      const T *values, uint64_t values_len, spal_csr_t *out) {
    if (!out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_create: out is NULL");
    *out = nullptr;
    if (!rowptr || (!colind && colind_len) || (!values && values_len))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_create: null array");
    int reason = 0;
    SPAL_TRY(spal_csr_validate(nrows, ncols, rowptr, rowptr_len, colind, colind_len, values_len, &reason));
    const uint64_t nnz = rowptr[nrows];
    if (nrows >= 0xffffffffull || ncols > 0xffffffffull)
        return fail(SPAL_ERR_UNSUPPORTED, "shape %llu x %llu does not fit 32-bit device indices",
                    (unsigned long long)nrows, (unsigned long long)ncols);
    DeviceGuard guard(device);
    if (guard.status != SPAL_OK) return guard.status;
    const uint64_t limit = csr_part_entries();
    if (nnz <= limit) return csr_create_rows<T>(device, 0, nrows, ncols, rowptr, colind, values, out);

    // more entries than 32-bit offsets address: row blocks of at most `limit` entries, cut at multiples of 1024
    // rows (super-tiles stay whole) where that leaves a block non-empty
    spal_csr *a = new spal_csr;
    a->device = device;
    a->elem_size = (int)sizeof(T);
    a->nrows = nrows; a->ncols = ncols; a->nnz = nnz;
    hipError_t e = stream_acquire(&a->stream);
    if (e != hipSuccess) { csr_free(a); return fail(SPAL_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    for (uint64_t r0 = 0; r0 < nrows;) {
        // largest r1 with rowptr[r1] - rowptr[r0] <= limit
        uint64_t r1 = (uint64_t)(std::upper_bound(rowptr + r0, rowptr + nrows + 1, rowptr[r0] + limit) - rowptr) - 1;
        if (r1 <= r0) {   // (one row above the limit: impossible while columns are < 2^32 and strictly increasing)
            csr_free(a);
            return fail(SPAL_ERR_UNSUPPORTED, "row %llu alone holds more than %llu entries", (unsigned long long)r0,
                        (unsigned long long)limit);
        }
        if (r1 < nrows && (r1 & ~1023ull) > r0) r1 &= ~1023ull;
        spal_csr *part = nullptr;
        const int st = csr_create_rows<T>(device, r0, r1, ncols, rowptr, colind, values, &part);
        if (st != SPAL_OK) { csr_free(a); return st; }
        a->parts.push_back(part);
        a->part_row0.push_back(r0);
        a->part_entry0.push_back(rowptr[r0]);
        r0 = r1;
    }
    a->part_row0.push_back(nrows);
    a->part_entry0.push_back(nnz);
    *out = a;
    return SPAL_OK;
}

template <typename T>
static int csr_spmv_host(spal_csr_t a, const T *x, uint64_t x_len, T *y, uint64_t y_len) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_spmv: handle is NULL");
    if (a->elem_size != (int)sizeof(T))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_spmv: handle holds %s values",
                    a->elem_size == 8 ? "f64" : "f32");
    if (x_len != a->ncols)
        return fail(SPAL_ERR_INVALID_ARGUMENT,
                    "dimension mismatch: x.len() = %llu but ncols = %llu (assert_eq!, csr/ops/mul.rs:9)",
                    (unsigned long long)x_len, (unsigned long long)a->ncols);
    if (y_len != a->nrows)
        return fail(SPAL_ERR_INVALID_ARGUMENT, "y.len() = %llu but nrows = %llu",
                    (unsigned long long)y_len, (unsigned long long)a->nrows);
    if (!x || !y) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_spmv: null vector");
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    std::lock_guard<std::mutex> lock(a->mu);
    if (!a->d_x) SPAL_HIP_TRY(dev_alloc((void **)&a->d_x, a->ncols * sizeof(T)));
    if (!a->d_y) SPAL_HIP_TRY(dev_alloc((void **)&a->d_y, a->nrows * sizeof(T)));
    SPAL_HIP_TRY(hipMemcpyAsync(a->d_x, x, a->ncols * sizeof(T), hipMemcpyHostToDevice, a->stream));
    SPAL_TRY(csr_launch(a, a->d_x, a->d_y, a->stream));
    SPAL_HIP_TRY(hipMemcpyAsync(y, a->d_y, a->nrows * sizeof(T), hipMemcpyDeviceToHost, a->stream));
    SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
    return SPAL_OK;
}

template <typename T>
static int csr_spmv_dev(spal_csr_t a, const T *x_dev, T *y_dev, void *stream) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_spmv_dev: handle is NULL");
    if (a->elem_size != (int)sizeof(T))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_spmv_dev: handle holds %s values",
                    a->elem_size == 8 ? "f64" : "f32");
    if (!x_dev || !y_dev) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_spmv_dev: null vector");
    int cur = -1;
    SPAL_HIP_TRY(hipGetDevice(&cur));
    if (cur != a->device) {
        DeviceGuard guard(a->device);
        if (guard.status != SPAL_OK) return guard.status;
        return csr_launch(a, x_dev, y_dev, (hipStream_t)stream);
    }
    return csr_launch(a, x_dev, y_dev, (hipStream_t)stream);
}

template <typename T>
static int csr_download(spal_csr_t a, uint64_t *rowptr, uint64_t *colind, T *values) {
    if (!a || !rowptr) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_download: null argument");
    if (a->elem_size != (int)sizeof(T))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_download: handle holds %s values",
                    a->elem_size == 8 ? "f64" : "f32");
    if (a->nnz && (!colind || !values))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_download: null array");
    if (!a->parts.empty()) {   // row blocks: each fills its slices; offsets become absolute again
        for (size_t b = 0; b < a->parts.size(); ++b) {
            const uint64_t r0 = a->part_row0[b], e0 = a->part_entry0[b];
            SPAL_TRY(csr_download<T>(a->parts[b], rowptr + r0, colind ? colind + e0 : nullptr, values ? values + e0 : nullptr));
            for (uint64_t i = r0; i <= a->part_row0[b + 1]; ++i) rowptr[i] += e0;
        }
        return SPAL_OK;
    }
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    std::vector<uint32_t> rp(a->nrows + 1), ci(a->nnz);
    SPAL_HIP_TRY(hipMemcpy(rp.data(), a->d_rowptr, rp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (a->nnz) {
        SPAL_HIP_TRY(hipMemcpy(ci.data(), a->d_colind, ci.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        SPAL_HIP_TRY(hipMemcpy(values, a->d_values, a->nnz * sizeof(T), hipMemcpyDeviceToHost));
    }
    for (uint64_t i = 0; i <= a->nrows; ++i) rowptr[i] = rp[i];
    for (uint64_t i = 0; i < a->nnz; ++i) colind[i] = ci[i];
    return SPAL_OK;
}

// Times the applicable variants of the planned kernel on the caller's vectors
// and keeps the fastest (all variants compute identical results).  Setup-time
// work: it synchronises `stream`.
//  1. form: one super-tile per workgroup, or the walking form -- the sliding-window kernel when the plan has
//     it (bands), else the persistent form -- each with plain or non-temporal y stores;
//  2. placement of the 16-bit columns relative to the values (two streams out of one class of region of the device's
//     memory disturb each other): the columns are tried in up to `place_tries` blocks of 1 GiB taken one
//     after the other from the device's memory; the fastest place is kept.
template <typename T>
static int csr_autotune(spal_csr_t a, const T *x_dev, T *y_dev, void *stream, int iters) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_autotune: handle is NULL");
    if (a->elem_size != (int)sizeof(T))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_autotune: handle holds %s values",
                    a->elem_size == 8 ? "f64" : "f32");
    if (!x_dev || !y_dev) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_autotune: null vector");
    if (!a->parts.empty()) {   // row blocks: each tunes its own plan on its rows of y
        for (size_t b = 0; b < a->parts.size(); ++b)
            SPAL_TRY(csr_autotune<T>(a->parts[b], x_dev, y_dev + a->part_row0[b], stream, iters));
        return SPAL_OK;
    }
    if (iters < 1) iters = 1;
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    SPAL_TRY(csr_ensure_plan(a, nullptr, false));
    if (a->bw_on) return SPAL_OK;   // (nothing to tune: one kernel, its geometry fixed by the windows)
    if (a->split_short) return csr_autotune<T>(a->split_short, x_dev, y_dev, stream, iters);   // (the short part's kernels; y is scratch here)
    std::lock_guard<std::mutex> lock(a->mu);
    CsrPlan &p = a->plan;
    for (float &t : a->tuned_us) t = 0.f;
    a->place_us[0] = a->place_us[1] = 0.f;
    a->place_tried = 0;
    if (a->nnz == 0 || p.kernel != 2 || p.tiles_per_wave != 4) return SPAL_OK;  // nothing to choose from
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    SPAL_HIP_TRY(hipEventCreate(&e0));
    {
        const hipError_t ee = hipEventCreate(&e1);
        if (ee != hipSuccess) {
            (void)hipEventDestroy(e0);
            return fail(SPAL_ERR_HIP, "spal_csr_autotune: hipEventCreate: %s", hipGetErrorString(ee));
        }
    }
    int rc = SPAL_OK;
    auto timed = [&](int n, float *ms_per_launch) {   // n launches of the current configuration
        for (int i = 0; i < 3 && rc == SPAL_OK; ++i) rc = csr_launch(a, x_dev, y_dev, st);
        if (rc != SPAL_OK) return;
        hipError_t e = hipEventRecord(e0, st);
        for (int i = 0; i < n && rc == SPAL_OK; ++i) rc = csr_launch(a, x_dev, y_dev, st);
        if (e == hipSuccess) e = hipEventRecord(e1, st);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) rc = fail(SPAL_ERR_HIP, "spal_csr_autotune: %s", hipGetErrorString(e));
        *ms_per_launch = ms / (float)n;
    };
    // ---- 0. columns anywhere: the column-blocked kernel against the stream kernels (results are bit-identical)
    a->cblock_us[0] = a->cblock_us[1] = 0.f;
    {   // (under the lock a first product on another thread takes for the same build, csr_launch)
        std::lock_guard<std::mutex> lock(a->mu_cb);
        if (p.cblock_pending) {
            (void)cblock_plan(a, false);
            __atomic_store_n(&p.cblock_pending, 0, __ATOMIC_RELEASE);
        }
    }
    if (p.cblock) {
        float ms[2] = {0.f, 0.f};
        for (int round = 0; round < 2 && rc == SPAL_OK; ++round)
            for (int on = 0; on < 2 && rc == SPAL_OK; ++on) { p.cblock_on = on; timed(std::max(3, iters / 3), &ms[on]); }
        if (rc == SPAL_OK) {
            a->cblock_us[0] = ms[0] * 1e3f; a->cblock_us[1] = ms[1] * 1e3f;
            p.cblock_on = ms[1] <= ms[0] ? 1 : 0;
        }
        if (p.cblock_on) {   // nothing of the stream kernels' forms to choose
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            return rc;
        }
    }
    // ---- 1. the form.  candidate c: bit 0 = walking form (sliding kernel / persistent), bit 1 = non-temporal y stores
    const bool walking_is_slide = p.slide != 0;
    const int planned = ((walking_is_slide ? p.slide_on : p.persistent) ? 1 : 0) | (p.nt_store ? 2 : 0);   // what the plan chose by structure
    int best = planned;
    float best_ms = 1e30f, planned_ms = 1e30f;
    for (int round = 0; round < 2 && rc == SPAL_OK; ++round) {      // round 0 also settles the clocks
        for (int cand = 0; cand < 4 && rc == SPAL_OK; ++cand) {
            if (walking_is_slide) { p.slide_on = cand & 1; p.slide_fill_ok = 1; p.persistent = 0; }
            else p.persistent = cand & 1;
            p.nt_store = (cand >> 1) & 1;
            float ms = 0.f;
            timed(iters, &ms);
            if (rc == SPAL_OK && round == 1) {
                a->tuned_us[cand] = ms * 1e3f;
                if (ms < best_ms) { best_ms = ms; best = cand; }
                if (cand == planned) planned_ms = ms;
            }
        }
    }
    // (a form has to beat the planned one by 1 %: at config 3 the two forms measure within 0.1 us of each other on some boxes,
    //  and the one-super-tile form picked on such a margin then ran 4 % slower over the timed launches than the sliding form does)
    if (planned_ms <= 1.01f * best_ms) best = planned;
    if (walking_is_slide) { p.slide_on = best & 1; p.slide_fill_ok = 1; p.persistent = 0; }
    else p.persistent = best & 1;
    p.user_persistent = true;   // measured: a later re-plan keeps it
    p.nt_store = (best >> 1) & 1;
    // ---- 2. where the 16-bit columns lie relative to the values (two streams out of one class of region
    // disturb each other, +12 us at config 3; out of two classes they do not): the columns are copied into blocks of
    // 1 GiB taken one after the other from the device's memory, the kernel is timed on each, the fastest place is kept
    // (round 2 re-allocated the 1.1 GB values array up to 12 times and, the candidates lying side by side in one
    // region, often found nothing).  `place_tries` blocks (default 8: ~25 ms), up to three times as many while nothing better turns up.
    // Round 4: the candidates are the process's placement blocks (spal_csr_alloc_vectors' walk found and kept them: at most
    // two, no hipMalloc here), the columns become a PIECE of the one that wins by 1 % and more.
    int tries = p.place_tries;
    if (const char *e = getenv("SPAL_PLACE_TRIES")) tries = atoi(e);
    const size_t cbytes = a->d_col16 ? (size_t)a->cap_entries * sizeof(uint16_t) : 0;
    if (rc == SPAL_OK && tries > 0 && cbytes >= ((size_t)64 << 20) && !a->col16_placed) {
        const int n = std::max(4, iters / 4);
        float cur_ms = 0.f;
        timed(n, &cur_ms);
        a->place_us[0] = cur_ms * 1e3f;
        uint16_t *const original = a->d_col16;
        std::vector<void *> cand;
        std::vector<float> ms_of;
        const int kept = place_block_count(a->device);
        for (int k = 0; k < kept && k < tries && rc == SPAL_OK; ++k) {
            void *b = place_alloc(a->device, k, cbytes);
            if (!b) continue;
            hipError_t e = hipMemcpyAsync(b, original, cbytes, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) { place_free(a->device, b); rc = fail(SPAL_ERR_HIP, "spal_csr_autotune: %s", hipGetErrorString(e)); break; }
            a->d_col16 = (uint16_t *)b;
            float ms = 0.f;
            timed(n, &ms);
            cand.push_back(b); ms_of.push_back(ms);
            ++a->place_tried;
        }
        int best = -1;
        for (size_t k = 0; k < ms_of.size(); ++k)
            if (ms_of[k] < 0.99f * cur_ms && (best < 0 || ms_of[k] < ms_of[(size_t)best])) best = (int)k;
        (void)hipStreamSynchronize(st);
        a->d_col16 = best >= 0 ? (uint16_t *)cand[(size_t)best] : original;
        for (size_t k = 0; k < cand.size(); ++k)
            if ((int)k != best) place_free(a->device, cand[k]);
        if (best >= 0) { (void)dev_free(original); a->col16_placed = 1; }
        a->place_us[1] = (best >= 0 ? ms_of[(size_t)best] : cur_ms) * 1e3f;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

}  // namespace spal

using namespace spal;

extern "C" {

int spal_device_count(int *count) {
    if (!count) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return SPAL_OK;
}

int spal_csr_create_f64(int device, uint64_t nrows, uint64_t ncols, const uint64_t *rowptr,
                        uint64_t rowptr_len, const uint64_t *colind, uint64_t colind_len,
                        const double *values, uint64_t values_len, spal_csr_t *out) {
    return csr_create<double>(device, nrows, ncols, rowptr, rowptr_len, colind, colind_len, values,
                              values_len, out);
}
int spal_csr_create_f32(int device, uint64_t nrows, uint64_t ncols, const uint64_t *rowptr,
                        uint64_t rowptr_len, const uint64_t *colind, uint64_t colind_len,
                        const float *values, uint64_t values_len, spal_csr_t *out) {
    return csr_create<float>(device, nrows, ncols, rowptr, rowptr_len, colind, colind_len, values,
                             values_len, out);
}

int spal_csr_destroy(spal_csr_t a) {
    if (!a) return SPAL_OK;
    DeviceGuard guard(a->device);
    csr_free(a);
    return SPAL_OK;
}

int spal_csr_shape(spal_csr_t a, uint64_t *nrows, uint64_t *ncols, uint64_t *nnz, int *elem_size) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_shape: handle is NULL");
    if (nrows) *nrows = a->nrows;
    if (ncols) *ncols = a->ncols;
    if (nnz) *nnz = a->nnz;
    if (elem_size) *elem_size = a->elem_size;
    return SPAL_OK;
}

int spal_csr_spmv_f64(spal_csr_t a, const double *x, uint64_t x_len, double *y, uint64_t y_len) {
    return csr_spmv_host<double>(a, x, x_len, y, y_len);
}
int spal_csr_spmv_f32(spal_csr_t a, const float *x, uint64_t x_len, float *y, uint64_t y_len) {
    return csr_spmv_host<float>(a, x, x_len, y, y_len);
}
int spal_csr_spmv_dev_f64(spal_csr_t a, const double *x_dev, double *y_dev, void *stream) {
    return csr_spmv_dev<double>(a, x_dev, y_dev, stream);
}
int spal_csr_spmv_dev_f32(spal_csr_t a, const float *x_dev, float *y_dev, void *stream) {
    return csr_spmv_dev<float>(a, x_dev, y_dev, stream);
}
int spal_csr_download_f64(spal_csr_t a, uint64_t *rowptr, uint64_t *colind, double *values) {
    return csr_download<double>(a, rowptr, colind, values);
}
int spal_csr_download_f32(spal_csr_t a, uint64_t *rowptr, uint64_t *colind, float *values) {
    return csr_download<float>(a, rowptr, colind, values);
}

int spal_csr_set_option(spal_csr_t a, const char *key, int64_t value) {
    if (!a || !key) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_set_option: null argument");
    if (!a->parts.empty()) {   // row blocks: every block takes the option (each plans for its own rows)
        for (spal_csr *part : a->parts) SPAL_TRY(spal_csr_set_option(part, key, value));
        return SPAL_OK;
    }
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    SPAL_TRY(csr_ensure_plan(a, nullptr, false));
    if (!strcmp(key, "blockwin")) {
        // the block-window kernel (spal_csr_blockwin.hip): -1 = timed against the row split where that is built, 0 = never,
        // 1 = whenever the matrix's windows fit LDS
        if (value < -1 || value > 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "blockwin must be -1 (auto), 0 or 1");
        std::lock_guard<std::mutex> lock(a->mu);
        a->plan.blockwin = (int)value;
        return a->split_child ? SPAL_OK : csr_plan_build(a);
    }
    if (!strcmp(key, "row_split") || !strcmp(key, "row_split_threshold")) {
        // skewed row lengths: rows above the threshold multiplied apart from the rest (csr_try_row_split): -1 = when they keep
        // a tenth of the 64-row tiles from streaming, 0 = never, 1 = whenever there is such a row
        std::lock_guard<std::mutex> lock(a->mu);
        if (!strcmp(key, "row_split")) {
            if (value < -1 || value > 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "row_split must be -1 (auto), 0 or 1");
            a->plan.row_split = (int)value;
        } else {
            if (value < 1 || value > 4096) return fail(SPAL_ERR_INVALID_ARGUMENT, "row_split_threshold must be in [1, 4096]");
            a->plan.split_threshold = (int)value;
        }
        return a->split_child ? SPAL_OK : csr_plan_build(a);
    }
    if (a->split_short) return spal_csr_set_option(a->split_short, key, value);   // every other option concerns the short part's kernels
    if (a->bw_on) return SPAL_OK;   // (the block-window kernel has no options of its own; the stream kernels' do not concern it)
    std::lock_guard<std::mutex> lock(a->mu);
    CsrPlan saved = a->plan;
    CsrPlan &p = a->plan;
    if (!strcmp(key, "kernel")) {
        if (value < 0 || value > 2)
            return fail(SPAL_ERR_INVALID_ARGUMENT, "kernel must be 0 (auto), 1 (vector) or 2 (stream)");
        p.user_kernel = (int)value;
        if (value == 0) {
            p.user_rows_per_block = p.user_lanes = p.user_lds = p.user_unroll = p.user_threads = false;
            p.user_rows_per_tile = false;
        }
    } else if (!strcmp(key, "rows_per_block")) {
        if (value == 0) p.user_rows_per_block = false;
        else if (value < 16 || value > 65536 || (value % 16))
            return fail(SPAL_ERR_INVALID_ARGUMENT, "rows_per_block must be a multiple of 16 in [16, 65536]");
        else { p.rows_per_block = (int)value; p.user_rows_per_block = true; }
    } else if (!strcmp(key, "lanes_per_row")) {
        if (value == 0) p.user_lanes = false;
        else if (value < 2 || value > 64 || (value & (value - 1)))
            return fail(SPAL_ERR_INVALID_ARGUMENT, "lanes_per_row must be one of 2,4,8,16,32,64");
        else { p.lanes_per_row = (int)value; p.user_lanes = true; }
    } else if (!strcmp(key, "lds_x")) {
        if (value < -1 || value > 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "lds_x must be -1 (auto), 0 or 1");
        if (value < 0) p.user_lds = false;
        else { p.lds_x = (int)value; p.user_lds = true; }
    } else if (!strcmp(key, "unroll")) {
        if (value == 0) p.user_unroll = false;
        else if (value != 1 && value != 2 && value != 4)
            return fail(SPAL_ERR_INVALID_ARGUMENT, "unroll must be 1, 2 or 4");
        else { p.unroll = (int)value; p.user_unroll = true; }
    } else if (!strcmp(key, "stream_global")) {
        if (value != 0 && value != 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "stream_global must be 0 or 1");
        p.stream_global = (int)value;
    } else if (!strcmp(key, "window_pages")) {
        // stream kernel: LDS budget of a super-tile in 256-column pages; 0 = automatic (24 f64 pages
        // at two workgroups per CU, or up to 60 at one when that is estimated to pay)
        if (value < 0 || value > 64) return fail(SPAL_ERR_INVALID_ARGUMENT, "window_pages must be in [0, 64]");
        p.window_pages = (int)value;
    } else if (!strcmp(key, "col16")) {
        // vector kernel, long rows: 16-bit window-relative columns for blocks whose x window is in LDS
        if (value != 0 && value != 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "col16 must be 0 or 1");
        p.vec_col16_allowed = (int)value;
    } else if (!strcmp(key, "skew")) {
        // stream kernel: skewed product strips (-1 = automatic: when most rows are a multiple of 128 bytes long)
        if (value < -1 || value > 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "skew must be -1 (auto), 0 or 1");
        p.user_skew = value >= 0;
        if (value >= 0) p.skew = (int)value;
    } else if (!strcmp(key, "stream_row_max")) {
        // stream kernel: a tile with a row longer than this is left to the overflow kernel
        if (value < 1 || value > 1024) return fail(SPAL_ERR_INVALID_ARGUMENT, "stream_row_max must be in [1, 1024]");
        p.stream_row_max = (int)value;
    } else if (!strcmp(key, "nt_store")) {
        if (value != 0 && value != 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "nt_store must be 0 or 1");
        p.nt_store = (int)value;
    } else if (!strcmp(key, "persistent")) {
        if (value != 0 && value != 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "persistent must be 0 or 1");
        p.persistent = (int)value;
        p.user_persistent = true;
    } else if (!strcmp(key, "persistent_blocks")) {
        if (value != 0 && (value < 8 || value > 4096 || (value % 8)))
            return fail(SPAL_ERR_INVALID_ARGUMENT, "persistent_blocks must be 0 (auto) or a multiple of 8 in [8, 4096]");
        p.persistent_blocks = (int)value;
    } else if (!strcmp(key, "rows_per_tile")) {
        if (value == 0) p.user_rows_per_tile = false;
        else if (value != 256 && value != 128 && value != 64 && value != 32 && value != 24 && value != 16 && value != 12 && value != 8)
            return fail(SPAL_ERR_INVALID_ARGUMENT, "rows_per_tile must be 0 (auto), 256, 128, 64, 32, 24, 16, 12 or 8");
        else if (p.tiles_per_wave == 8 && value != 64)
            return fail(SPAL_ERR_INVALID_ARGUMENT, "rows_per_tile must be 64 (or 0 = auto) while tiles_per_wave = 8");
        else { p.rows_per_tile = (int)value; p.user_rows_per_tile = true; }
    } else if (!strcmp(key, "tiles_per_wave")) {
        if (value != 4 && value != 8) return fail(SPAL_ERR_INVALID_ARGUMENT, "tiles_per_wave must be 4 or 8");
        // the 8-tile form exists for 64-row tiles only (launch_stream_main): a plan built for another tile height
        // and launched with it would read descriptors of other rows
        if (value == 8 && p.user_rows_per_tile && p.rows_per_tile != 64)
            return fail(SPAL_ERR_INVALID_ARGUMENT, "tiles_per_wave = 8 needs rows_per_tile = 64 (or 0 = auto)");
        p.tiles_per_wave = (int)value;
    } else if (!strcmp(key, "slide")) {
        // the sliding-window kernel for band-like plans: -1 = use it where the plan allows (default), 0 = never
        // (the plan then keeps the window-relative col16 of the one-super-tile-per-workgroup kernels), 1 = as -1;
        // "slide_on" 0 keeps the ring plan but launches the one-super-tile-per-workgroup kernels on it (A/B)
        if (value < -1 || value > 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "slide must be -1 (auto), 0 or 1");
        p.slide_user = (int)value;
    } else if (!strcmp(key, "panel_pages")) {
        // super-tiles whose column span is at most this many 256-column pages (and wider than the LDS window) are
        // taken in column panels by csr_spmv_panel; 0 = never (x through L2)
        if (value < 0 || value > 255) return fail(SPAL_ERR_INVALID_ARGUMENT, "panel_pages must be in [0, 255]");
        p.panel_pages = (int)value;
    } else if (!strcmp(key, "panel_window")) {
        if (value < 0 || value > 624) return fail(SPAL_ERR_INVALID_ARGUMENT, "panel_window must be in [0, 624] pages");
        p.panel_window_user = (int)value;
    } else if (!strcmp(key, "panel_on")) {
        if (value != 0 && value != 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "panel_on must be 0 or 1");
        p.panel_on = (int)value;
    } else if (!strcmp(key, "slide_run")) {
        // sliding kernel: steps (of 4 tiles) per run; runs are dealt round-robin to an XCD's workgroups (0 = one run each)
        if (value < 0 || value > 65535) return fail(SPAL_ERR_INVALID_ARGUMENT, "slide_run must be in [0, 65535]");
        p.slide_run = (int)value;
    } else if (!strcmp(key, "slide_on")) {
        if (value != 0 && value != 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "slide_on must be 0 or 1");
        p.slide_on = (int)value;
        p.slide_fill_user = value ? 1 : -1;   // (asked for by name: also where the tiles are ragged)
        p.slide_fill_ok = value ? 1 : p.slide_fill_ok;
    } else if (!strcmp(key, "slide_even")) {
        // sliding kernel, one run per workgroup: steps split evenly over all workgroups of an XCD (default 1) or runs of ceil(steps / workgroups)
        if (value != 0 && value != 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "slide_even must be 0 or 1");
        p.slide_even = (int)value;
    } else if (!strcmp(key, "arith_bounds")) {
        // sliding kernel, matrices whose rows ALL have one length: tile bounds computed (r * length) instead of loaded (default 1)
        if (value != 0 && value != 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "arith_bounds must be 0 or 1");
        p.arith_bounds = (int)value;
    } else if (!strcmp(key, "split_tiles")) {
        // sliding kernel: tiles above 1024 entries whose two halves fit the strip are computed in two passes (1,
        // default) or left to the overflow kernel like every other skipped tile (0)
        if (value != 0 && value != 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "split_tiles must be 0 or 1");
        p.split_tiles_on = (int)value;
    } else if (!strcmp(key, "place_tries")) {
        // autotune: blocks of 1 GiB the 16-bit columns are tried in (0 = leave them where they are)
        if (value < 0 || value > 16) return fail(SPAL_ERR_INVALID_ARGUMENT, "place_tries must be in [0, 16]");
        p.place_tries = (int)value;
    } else if (!strcmp(key, "uniform_rows")) {
        // stream kernel: super-tiles whose rows all have one length do not read rowptr (default 1)
        if (value != 0 && value != 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "uniform_rows must be 0 or 1");
        p.uniform_rows = (int)value;
    } else if (!strcmp(key, "prefetch")) {
        // stream kernel: tiles of loads ahead of the one being summed
        if (value != 1 && value != 2) return fail(SPAL_ERR_INVALID_ARGUMENT, "prefetch must be 1 or 2");
        p.prefetch = (int)value;
    } else if (!strcmp(key, "diag")) {
#ifdef SPAL_DIAG
        if (value < 0 || value > 0xffff || (value & 0xff)) return fail(SPAL_ERR_INVALID_ARGUMENT, "diag: bits 8 ... 15 only");
        p.diag = (int)value;
#else
        return fail(SPAL_ERR_INVALID_ARGUMENT, "diag: this library is not an ablation build (-DSPAL_DIAG)");
#endif
    } else if (!strcmp(key, "walk_blocks")) {
        // spal_csr_alloc_vectors: blocks of 1 GiB the placement walk probes at most (1 = the first block, no search)
        if (value < 1 || value > 128) return fail(SPAL_ERR_INVALID_ARGUMENT, "walk_blocks must be in [1, 128]");
        a->walk_max = (int)value;
        return SPAL_OK;
    } else if (!strcmp(key, "xcd_chunk")) {
        // one-super-tile stream kernel: super-tiles dealt to the 8 XCDs in chunks of this many (0 = one contiguous run per XCD)
        if (value < 0 || value > 4096) return fail(SPAL_ERR_INVALID_ARGUMENT, "xcd_chunk must be in [0, 4096]");
        p.xcd_chunk = (int)value;
    } else if (!strcmp(key, "cblock")) {
        // the column-blocked kernel (csr_cblock.hpp): -1 = when most rows gather x from beyond L2, 0 = never, 1 = always
        if (value < -1 || value > 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "cblock must be -1 (auto), 0 or 1");
        p.cblock_user = (int)value;
        p.cblock_on = 1;
    } else if (!strcmp(key, "cblock_rows")) {
        if (value < 0 || value > 8192)
            return fail(SPAL_ERR_INVALID_ARGUMENT, "cblock_rows (rows of a row block of the column-blocked kernel) must be 0 (auto) or in [1, 8192]");
        p.cblock_rows_user = (int)value;
    } else if (!strcmp(key, "cblock_shift")) {
        if (value != 0 && (value < 8 || value > 24))
            return fail(SPAL_ERR_INVALID_ARGUMENT, "cblock_shift (log2 of the columns of a column block) must be 0 (auto: 2 MB of x) or in [8, 24]");
        p.cblock_shift_user = (int)value;
    } else if (!strcmp(key, "cblock_form")) {
        // -1 = by the entries per run, 0 = entry-parallel kernel, 1 = rows form (rows of a row block: 256 ... 4096, a power of two)
        if (value < -1 || value > 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "cblock_form must be -1 (auto), 0 (entry-parallel) or 1 (rows form)");
        p.cblock_form_user = (int)value;
    } else if (!strcmp(key, "threads")) {
        if (value == 0) p.user_threads = false;
        else if (value != 512 && value != 1024)
            return fail(SPAL_ERR_INVALID_ARGUMENT, "threads must be 512 or 1024");
        else { p.threads = (int)value; p.user_threads = true; }
    } else {
        return fail(SPAL_ERR_INVALID_ARGUMENT, "unknown option '%s'", key);
    }
    int st = csr_plan_build(a);
    if (st != SPAL_OK) { a->plan = saved; (void)csr_plan_build(a); }
    return st;
}

int spal_csr_alloc_vectors(spal_csr_t a, void **x_dev, void **y_dev, void *stream) {
    if (!a || !x_dev || !y_dev) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_alloc_vectors: null argument");
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    SPAL_TRY(csr_ensure_plan(a, nullptr, false));
    std::lock_guard<std::mutex> lock(a->mu);
    const size_t es = (size_t)a->elem_size;
    auto up = [](size_t v) { return (v + 4095) & ~(size_t)4095; };
    const size_t xb = up(std::max<uint64_t>(a->ncols, 1) * es), yb = up(std::max<uint64_t>(a->nrows, 1) * es);
    if (a->d_vec_block) {
        *x_dev = (char *)a->d_vec_block + a->vec_x_off;
        *y_dev = (char *)a->d_vec_block + a->vec_y_off;
        return SPAL_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    // small products, empty matrices, row-block handles: nothing to place -- a block of their own, exactly as large as needed
    size_t walk_min = (size_t)256 << 20;    // matrices the caches do not hold
    if (const char *e = getenv("SPAL_WALK_MIN_BYTES")) walk_min = (size_t)strtoull(e, nullptr, 10);
    const bool walk = a->parts.empty() && a->nnz != 0 && (size_t)a->nnz * (es + 2) >= walk_min && a->walk_max > 1 &&
                      xb + yb <= ((size_t)1 << 30);
    if (!walk) {
        void *b = nullptr;
        hipError_t e = hipMalloc(&b, up(xb + yb));
        if (e == hipSuccess) e = hipMemsetAsync(b, 0, xb + yb, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            if (b) (void)hipFree(b);
            return fail(e == hipErrorOutOfMemory ? SPAL_ERR_OUT_OF_MEMORY : SPAL_ERR_HIP, "spal_csr_alloc_vectors: %s", hipGetErrorString(e));
        }
        a->d_vec_block = b; a->vec_block_owned = 1; a->vec_x_off = 0; a->vec_y_off = xb;
        a->walk_blocks = 1; a->walk_probes = 0;
        *x_dev = (char *)b; *y_dev = (char *)b + xb;
        return SPAL_OK;
    }
    // Where x and y lie relative to the matrix stream decides +-5 - 12 % of a product.  The process keeps a few
    // PLACEMENT BLOCKS of 1 GiB per device (place_*): the FIRST handle that asks walks the device's memory -- blocks taken
    // one after the other, its kernel timed into a candidate y in each, at most `walk_blocks` (8 GiB) held at once -- and
    // keeps the block where it ran fastest and, when a second class of region showed (3 % apart), the one where it ran
    // slowest; the others go back.  Every LATER handle times itself in the kept blocks only (no hipMalloc, two probes) and
    // takes its vectors -- and, in the autotune, its 16-bit columns -- as PIECES of them: no handle keeps a GiB for 160 MB.
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SPAL_OK;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    auto probe = [&](void *xc, float *us_out) {           // the handle's kernel, x and y at xc
        void *yc = (char *)xc + xb;
        if (e == hipSuccess) e = hipMemsetAsync(xc, 0, xb + yb, st);   // x = 0: the time of a product does not depend on the values
        for (int i = 0; i < 3 && rc == SPAL_OK; ++i) rc = csr_launch(a, xc, yc, st);
        if (e == hipSuccess) e = hipEventRecord(e0, st);
        const int n = 8;
        for (int i = 0; i < n && rc == SPAL_OK; ++i) rc = csr_launch(a, xc, yc, st);
        if (e == hipSuccess) e = hipEventRecord(e1, st);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        *us_out = ms * 1e3f / (float)n;
    };
    std::vector<float> us;          // per candidate
    std::vector<void *> piece;      // its x (a piece of a placement block)
    a->walk_blocks = 0;
    a->walk_probes = 0;
    // (i) the process's blocks
    const int kept = place_block_count(a->device);
    for (int k = 0; k < kept && e == hipSuccess && rc == SPAL_OK; ++k) {
        void *pc = place_alloc(a->device, k, xb + yb);
        if (!pc) continue;
        float t = 0.f;
        probe(pc, &t);
        piece.push_back(pc); us.push_back(t);
        ++a->walk_probes;
    }
    // (ii) the walk, once per process and device (or when the kept blocks are full)
    if ((!place_walked(a->device) || piece.empty()) && e == hipSuccess && rc == SPAL_OK) {
        const size_t block = (size_t)1 << 30;
        std::vector<void *> fresh;
        std::vector<float> fresh_us;
        for (int k = 0; k < a->walk_max && e == hipSuccess && rc == SPAL_OK; ++k) {
            void *b = nullptr;
            if (hipMalloc(&b, block) != hipSuccess) { (void)hipGetLastError(); break; }   // the device is full: what we have
            float t = 0.f;
            probe(b, &t);
            fresh.push_back(b); fresh_us.push_back(t);
            ++a->walk_blocks; ++a->walk_probes;
        }
        if (e == hipSuccess && rc == SPAL_OK && !fresh.empty()) {
            size_t lo = 0, hi = 0;
            for (size_t k = 1; k < fresh.size(); ++k) {
                if (fresh_us[k] < fresh_us[lo]) lo = k;
                if (fresh_us[k] > fresh_us[hi]) hi = k;
            }
            const bool two = fresh_us[hi] > 1.03f * fresh_us[lo];
            for (size_t k = 0; k < fresh.size(); ++k) {
                if (k == lo || (two && k == hi)) {
                    place_adopt(a->device, fresh[k], block);
                    void *pc = place_alloc(a->device, place_block_count(a->device) - 1, xb + yb);   // (its start: where it was timed)
                    piece.push_back(pc); us.push_back(fresh_us[k]);
                } else {
                    (void)hipFree(fresh[k]);
                }
            }
            place_set_walked(a->device);
            if (getenv("SPAL_WALK_DEBUG")) {
                fprintf(stderr, "[spal walk] us per product by new block:");
                for (float t : fresh_us) fprintf(stderr, " %.1f", t);
                fprintf(stderr, "  -> kept %zu%s\n", lo, two ? " and the slowest" : "");
            }
        } else {
            for (void *b : fresh) (void)hipFree(b);
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    size_t best = 0;
    for (size_t k = 1; k < us.size(); ++k) if (us[k] < us[best]) best = k;
    const bool ok = e == hipSuccess && rc == SPAL_OK && !piece.empty() && piece[best] != nullptr;
    for (size_t k = 0; k < piece.size(); ++k)
        if (!ok || k != best) place_free(a->device, piece[k]);
    if (rc != SPAL_OK) return rc;
    if (!ok) return fail(e == hipSuccess ? SPAL_ERR_OUT_OF_MEMORY : SPAL_ERR_HIP, "spal_csr_alloc_vectors: %s",
                         e == hipSuccess ? "no device memory for the vectors" : hipGetErrorString(e));
    a->d_vec_block = piece[best];
    a->vec_block_owned = 0;
    a->vec_x_off = 0;
    a->vec_y_off = xb;
    a->walk_us[0] = us[best];
    a->walk_us[1] = *std::max_element(us.begin(), us.end());
    if (getenv("SPAL_WALK_DEBUG")) {
        fprintf(stderr, "[spal walk] us per product by candidate:");
        for (float t : us) fprintf(stderr, " %.1f", t);
        fprintf(stderr, "  -> %zu (%d new blocks, %d probes)\n", best, a->walk_blocks, a->walk_probes);
    }
    *x_dev = (char *)a->d_vec_block + a->vec_x_off;
    *y_dev = (char *)a->d_vec_block + a->vec_y_off;
    return SPAL_OK;
}

int spal_csr_autotune_f64(spal_csr_t a, const double *x_dev, double *y_dev, void *stream, int iters) {
    return csr_autotune<double>(a, x_dev, y_dev, stream, iters);
}
int spal_csr_autotune_f32(spal_csr_t a, const float *x_dev, float *y_dev, void *stream, int iters) {
    return csr_autotune<float>(a, x_dev, y_dev, stream, iters);
}

int spal_csr_plan(spal_csr_t a) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_plan: handle is NULL");
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    return csr_ensure_plan(a, nullptr, false);
}
int spal_csr_describe(spal_csr_t a, char *buf, size_t buf_len) {
    if (!a || !buf || !buf_len) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csr_describe: null argument");
    if (!a->parts.empty()) {   // row blocks: the shape of the whole, the cuts, and the first block's plan
        std::string rows = "[";
        for (size_t b = 0; b < a->part_row0.size(); ++b) rows += (b ? ", " : "") + std::to_string(a->part_row0[b]);
        rows += "]";
        std::vector<char> first(buf_len);
        SPAL_TRY(spal_csr_describe(a->parts[0], first.data(), first.size()));
        snprintf(buf, buf_len,
                 "{\"format\": \"csr\", \"dtype\": \"%s\", \"nrows\": %llu, \"ncols\": %llu, \"nnz\": %llu, "
                 "\"kernel\": \"row_blocks\", \"parts\": %zu, \"part_rows\": %s, \"part0\": %s}",
                 a->elem_size == 8 ? "f64" : "f32", (unsigned long long)a->nrows, (unsigned long long)a->ncols,
                 (unsigned long long)a->nnz, a->parts.size(), rows.c_str(), first.data());
        return SPAL_OK;
    }
    {   // (describing the plan of a device-assembled handle builds it: what is printed is what the products will run)
        DeviceGuard guard(a->device);
        if (guard.status != SPAL_OK) return guard.status;
        SPAL_TRY(csr_ensure_plan(a, nullptr, false));
    }
    if (a->bw_on) {
        snprintf(buf, buf_len,
                 "{\"format\": \"csr\", \"dtype\": \"%s\", \"nrows\": %llu, \"ncols\": %llu, \"nnz\": %llu, \"kernel\": \"blockwin\", "
                 "\"index_bits\": 32, \"block_rows\": %u, \"blocks\": %u, \"window_columns\": %u, \"threads_per_block\": 1024, "
                 "\"pass_entries\": 4096, \"thread_rows_up_to\": 32, \"setup_us\": [%.1f, %.1f]}",
                 a->elem_size == 8 ? "f64" : "f32", (unsigned long long)a->nrows, (unsigned long long)a->ncols,
                 (unsigned long long)a->nnz, a->bw_rows, a->bw_blocks, a->bw_cols, a->bw_us[0], a->bw_us[1]);
        return SPAL_OK;
    }
    if (a->split_short) {   // row split: the whole, what was split off, and the short part's plan
        std::vector<char> part(buf_len);
        SPAL_TRY(spal_csr_describe(a->split_short, part.data(), part.size()));
        snprintf(buf, buf_len,
                 "{\"format\": \"csr\", \"dtype\": \"%s\", \"nrows\": %llu, \"ncols\": %llu, \"nnz\": %llu, \"kernel\": \"split\", "
                 "\"split_threshold\": %d, \"split_long_rows\": %u, \"split_long_entries\": %llu, \"short_part\": %s}",
                 a->elem_size == 8 ? "f64" : "f32", (unsigned long long)a->nrows, (unsigned long long)a->ncols,
                 (unsigned long long)a->nnz, a->plan.split_threshold, a->split_nlong, (unsigned long long)a->split_long_entries, part.data());
        return SPAL_OK;
    }
    const CsrPlan &p = a->plan;
    snprintf(buf, buf_len,
             "{\"format\": \"csr\", \"dtype\": \"%s\", \"nrows\": %llu, \"ncols\": %llu, \"nnz\": %llu, "
             "\"index_bits\": %d, \"kernel\": \"%s\", \"lanes_per_row\": %d, \"unroll\": %d, "
             "\"rows_per_block\": %d, \"rows_per_tile\": %d, \"blocks\": %u, \"threads_per_block\": %d, \"lds_x\": %d, "
             "\"lds_window_bytes\": %llu, \"lds_row_fraction\": %.4f, \"stream_row_fraction\": %.4f, "
             "\"overflow_tiles\": %u, \"skew\": %d, \"persistent\": %d, \"nt_store\": %d, \"uniform_row_fraction\": %.4f, "
             "\"prefetch\": %d, \"slide\": %d, \"ring_pages\": %d, \"tile_steps\": %d, \"split_tiles\": %u, \"panel_tiles\": %u, \"autotune_us\": [%.1f, %.1f, %.1f, %.1f], \"placement_us\": [%.1f, %.1f], \"placement_tries\": %d, \"addr\": [\"%llx\", \"%llx\", \"%llx\"], "
             "\"vectors_walk_us\": [%.1f, %.1f], \"vectors_walk_blocks\": %d, \"vectors_probes\": %d, \"placement_blocks\": %d, \"placement_free_bytes\": %llu, "
             "\"nonlocal_row_fraction\": %.4f, \"cblock\": %d, \"cblock_pending\": %d, \"cblock_rows\": %d, \"cblock_form\": \"%s\", \"cblock_run\": %.2f, \"cblock_cols\": %llu, \"cblock_col_blocks\": %d, \"cblock_row_blocks\": %u, \"cblock_us\": [%.1f, %.1f], \"cblock_failed\": %d}",
             a->elem_size == 8 ? "f64" : "f32", (unsigned long long)a->nrows,
             (unsigned long long)a->ncols, (unsigned long long)a->nnz, (p.kernel == 2 || p.vec_col16) ? 16 : 32,
             (p.cblock && p.cblock_on) ? "cblock" : p.kernel == 2 ? "stream" : "vector", p.lanes_per_row, p.kernel == 2 ? 2 : p.unroll,
             p.rows_per_block, p.kernel == 2 ? p.rows_per_tile : 0, p.nblocks, p.threads, p.lds_x,
             (unsigned long long)p.lds_entries * (unsigned long long)a->elem_size, p.lds_row_fraction,
             p.stream_row_fraction,
             p.kernel == 2 ? ((p.slide && p.slide_on) ? a->n_ovtiles_slide : a->n_ovtiles) : 0u,   // (tiles the overflow kernel runs)
             (p.kernel == 2 && p.skew) ? 1 : 0,
             (p.kernel == 2 && p.persistent && p.tiles_per_wave == 4) ? 1 : 0,
             p.kernel == 2 ? p.nt_store : 0, p.kernel == 2 ? p.uniform_row_fraction : 0.0,
             p.kernel == 2 ? p.prefetch : 0, (p.kernel == 2 && p.slide && p.slide_on && p.slide_fill_ok) ? 1 : 0,
             p.kernel == 2 ? p.ring_pages : 0, (p.kernel == 2 && p.slide) ? p.slide_S : 0,
             (p.kernel == 2 && p.slide && p.slide_on) ? a->n_split_tiles : 0u,
             (p.kernel == 2 && p.panel_on) ? a->n_ptiles : 0u,
             (double)a->tuned_us[0], (double)a->tuned_us[1],
             (double)a->tuned_us[2], (double)a->tuned_us[3], (double)a->place_us[0], (double)a->place_us[1],
             a->place_tried, (unsigned long long)(uintptr_t)a->d_values,
             (unsigned long long)(uintptr_t)a->d_col16, (unsigned long long)(uintptr_t)a->d_rowptr,
             (double)a->walk_us[0], (double)a->walk_us[1], a->walk_blocks, a->walk_probes, place_block_count(a->device),
             (unsigned long long)place_free_bytes(a->device),
             p.kernel == 2 ? p.nonlocal_row_fraction : 0.0, (p.cblock && p.cblock_on) ? 1 : 0, p.cblock_pending, p.cblock ? p.cblock_rows : 0, !p.cblock ? "" : p.cblock_form ? "rows" : "entry", p.cblock ? (double)p.cblock_run : 0.0,
             p.cblock ? (1ull << p.cblock_shift) : 0ull, p.cblock ? p.cblock_nbc : 0, p.cblock ? p.cblock_nrb : 0u,
             (double)a->cblock_us[0], (double)a->cblock_us[1], a->cblock_failed);
    return SPAL_OK;
}

// ---- device memory helpers ---------------------------------------------------
int spal_dev_malloc(int device, size_t bytes, void **ptr) {
    if (!ptr) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_dev_malloc: ptr is NULL");
    *ptr = nullptr;
    DeviceGuard guard(device);
    if (guard.status != SPAL_OK) return guard.status;
    SPAL_HIP_TRY(dev_alloc((void **)ptr, bytes ? bytes : 1));
    return SPAL_OK;
}
int spal_dev_free(int device, void *ptr) {
    if (!ptr) return SPAL_OK;
    DeviceGuard guard(device);
    if (guard.status != SPAL_OK) return guard.status;
    SPAL_HIP_TRY(dev_free(ptr));
    return SPAL_OK;
}
int spal_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes) {
    if (bytes && (!dst_dev || !src_host)) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_memcpy_h2d: null pointer");
    DeviceGuard guard(device);
    if (guard.status != SPAL_OK) return guard.status;
    if (bytes) SPAL_HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return SPAL_OK;
}
int spal_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes) {
    if (bytes && (!dst_host || !src_dev)) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_memcpy_d2h: null pointer");
    DeviceGuard guard(device);
    if (guard.status != SPAL_OK) return guard.status;
    if (bytes) SPAL_HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return SPAL_OK;
}
int spal_cache_trim(void) {
    dev_cache_trim();
    place_trim();
    return SPAL_OK;
}
int spal_device_synchronize(int device) {
    DeviceGuard guard(device);
    if (guard.status != SPAL_OK) return guard.status;
    SPAL_HIP_TRY(hipDeviceSynchronize());
    return SPAL_OK;
}

}  // extern "C"
