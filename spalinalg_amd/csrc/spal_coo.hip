// spal_coo.hip -- COO upload and device-side COO -> CSR assembly.
//
// Contract (reference src/csr/conv/coo.rs:4-115):
//   order entries by (row, col), STABLY w.r.t. insertion order;
//   sum every run of equal (row, col) left to right (separately rounded adds);
//   drop sums that compare equal to zero (-0.0 dropped, NaN kept);
//   emit CSR (columns strictly increasing inside a row).
// rowptr / colind / values are bit-identical to the reference's result: the
// sort is a stable LSD radix sort and each run is summed by ONE thread in
// insertion order.
//
// Pipeline (all on the device, one host sync for the data-dependent size):
//   key = row << cbits | col  (u64), payload = insertion index (u32)
//   ceil((rbits + cbits) / 8) radix passes: histogram -> scan -> stable scatter
//   run heads + sequential run sums -> keep flags -> scan -> compaction
//   rowptr[r] = lower_bound(row of kept entries, r)
#include "spal_internal.hpp"

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

__global__ __launch_bounds__(kScanThreads) void scan_apply(const uint32_t *__restrict__ in,
                                                           uint32_t *__restrict__ out, uint64_t n,
                                                           const uint32_t *__restrict__ sums) {
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
    }
}

// out[i] = sum in[0..i) ; *d_total (device, may be NULL) = sum of all.  `sums`
// must hold ceil(n / kScanTile) u32.  in == out allowed.
static hipError_t exclusive_scan_u32(const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *sums,
                                     uint32_t *d_total, hipStream_t st) {
    if (n == 0) {
        if (d_total) return hipMemsetAsync(d_total, 0, sizeof(uint32_t), st);
        return hipSuccess;
    }
    const uint32_t tiles = (uint32_t)((n + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL(scan_tile_sums, dim3(tiles), dim3(kScanThreads), 0, st, in, n, sums);
    hipLaunchKernelGGL(scan_sums_inplace, dim3(1), dim3(kScanThreads), 0, st, sums, tiles, d_total);
    hipLaunchKernelGGL(scan_apply, dim3(tiles), dim3(kScanThreads), 0, st, in, out, n, sums);
    return hipGetLastError();
}

// --------------------------------------------------------------------------
// stable LSD radix sort of (u64 key, u32 payload), 8 bits per pass
// --------------------------------------------------------------------------
constexpr int kSortThreads = 256;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortItems = 16;                            // per thread
constexpr int kSortTile = kSortThreads * kSortItems;      // 4096 keys per workgroup
constexpr int kWaveChunk = 64 * kSortItems;               // 1024 consecutive keys per wave

__global__ __launch_bounds__(256) void coo_make_keys(const uint32_t *__restrict__ rows,
                                                     const uint32_t *__restrict__ cols,
                                                     uint64_t *__restrict__ keys,
                                                     uint32_t *__restrict__ idx, uint64_t len,
                                                     uint32_t cbits) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < len) {
        keys[i] = ((uint64_t)rows[i] << cbits) | (uint64_t)cols[i];
        idx[i] = (uint32_t)i;
    }
}

// counts[d * nblk + blk] = number of keys of tile blk with digit d
__global__ __launch_bounds__(kSortThreads) void radix_hist(const uint64_t *__restrict__ keys,
                                                           uint64_t len, uint32_t shift,
                                                           uint32_t *__restrict__ counts,
                                                           uint32_t nblk) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t t0 = (uint64_t)blockIdx.x * kSortTile;
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = t0 + (uint64_t)j * kSortThreads + threadIdx.x;
        if (i < len) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 0xffu], 1u);
    }
    __syncthreads();
    counts[(uint64_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter.  Wave w of a workgroup owns the tile's keys [w*1024, (w+1)*1024)
// and walks them 64 at a time, so tile order = (wave, round, lane).  The rank
// of a key among equal digits is
//   offs[d][blk] (scanned counts) + keys of earlier waves + keys of earlier
//   rounds of this wave + earlier lanes of this round,
// all computed without atomics, hence deterministic and stable.
__global__ __launch_bounds__(kSortThreads) void radix_scatter(
    const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint64_t *__restrict__ kout,
    uint32_t *__restrict__ vout, uint64_t len, uint32_t shift, const uint32_t *__restrict__ offs,
    uint32_t nblk) {
    // volatile: lanes of a wave hand counts to each other through this array
    // between two rounds; the compiler must re-read it every round
    __shared__ uint32_t cnt_store[kSortWaves][256];
    volatile uint32_t (*cnt)[256] = cnt_store;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < kSortWaves * 256; i += kSortThreads) (&cnt_store[0][0])[i] = 0;
    __syncthreads();

    const uint64_t w0 = (uint64_t)blockIdx.x * kSortTile + (uint64_t)w * kWaveChunk;
    const uint64_t lt = (1ull << lane) - 1ull;
    uint64_t key[kSortItems];
    uint32_t val[kSortItems], rank[kSortItems];
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = w0 + (uint64_t)j * 64 + lane;
        const bool ok = i < len;
        key[j] = ok ? kin[i] : 0ull;
        val[j] = ok ? vin[i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = w0 + (uint64_t)j * 64 + lane;
        const bool ok = i < len;
        const uint32_t d = (uint32_t)(key[j] >> shift) & 0xffu;
        // lanes of this round with the same digit (inactive tail lanes excluded)
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = ok ? cnt[w][d] : 0u;
        rank[j] = before + (uint32_t)__popcll(peers & lt);
        // the lowest peer lane publishes the new count (one writer per digit)
        if (ok && (peers & lt) == 0) cnt[w][d] = before + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    // exclusive prefix over the waves, per digit, plus the tile's global offset
    {
        const uint32_t d = threadIdx.x;  // 256 threads = 256 digits
        uint32_t run = offs[(uint64_t)d * nblk + blockIdx.x];
#pragma unroll
        for (int ww = 0; ww < kSortWaves; ++ww) {
            const uint32_t c = cnt[ww][d];
            cnt[ww][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = w0 + (uint64_t)j * 64 + lane;
        if (i < len) {
            const uint32_t d = (uint32_t)(key[j] >> shift) & 0xffu;
            const uint32_t pos = cnt[w][d] + rank[j];
            kout[pos] = key[j];
            vout[pos] = val[j];
        }
    }
}

// --------------------------------------------------------------------------
// runs of equal keys: sequential sums in insertion order, zero drop
// --------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void coo_run_sums(const uint64_t *__restrict__ keys,
                                                    const uint32_t *__restrict__ idx,
                                                    const T *__restrict__ vals, uint64_t len,
                                                    T *__restrict__ runsum,
                                                    uint32_t *__restrict__ keep) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    const uint64_t k = keys[i];
    uint32_t flag = 0;
    if (i == 0 || keys[i - 1] != k) {
        // coo.rs:42-46: colval[prev] += val, one entry after the other
        T acc = vals[idx[i]];
        for (uint64_t j = i + 1; j < len && keys[j] == k; ++j) acc = acc + vals[idx[j]];
        runsum[i] = acc;
        flag = (acc != T(0)) ? 1u : 0u;  // coo.rs:64  `colval[ptr] != T::zero()`
    }
    keep[i] = flag;
}

template <typename T>
__global__ __launch_bounds__(256) void coo_compact(const uint64_t *__restrict__ keys,
                                                   const T *__restrict__ runsum,
                                                   const uint32_t *__restrict__ keep,
                                                   const uint32_t *__restrict__ pos, uint64_t len,
                                                   uint32_t cbits, uint32_t *__restrict__ out_row,
                                                   uint32_t *__restrict__ out_col,
                                                   T *__restrict__ out_val) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len || !keep[i]) return;
    const uint32_t q = pos[i];
    const uint64_t k = keys[i];
    out_row[q] = (uint32_t)(k >> cbits);
    out_col[q] = (uint32_t)(k & ((1ull << cbits) - 1ull));
    out_val[q] = runsum[i];
}

// rowptr[r] = first kept entry whose row is >= r   (r in [0, nrows])
__global__ __launch_bounds__(256) void coo_rowptr(const uint32_t *__restrict__ out_row,
                                                  uint32_t nnz, uint32_t nrows,
                                                  uint32_t *__restrict__ rowptr) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r > nrows) return;
    uint32_t lo = 0, hi = nnz;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)out_row[mid] < r) lo = mid + 1; else hi = mid;
    }
    rowptr[r] = lo;
}

static uint32_t bits_for(uint64_t n) {  // bits needed for values in [0, n)
    uint32_t b = 0;
    while (b < 64 && (1ull << b) < n) ++b;
    return b ? b : 1;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename U> U *as() { return reinterpret_cast<U *>(p); }
    void *release() { void *q = p; p = nullptr; return q; }
};

template <typename T>
static int coo_assemble_t(spal_coo *c, hipStream_t st, spal_csr **out) {
    const uint64_t len = c->len;
    const uint32_t cbits = bits_for(c->ncols), rbits = bits_for(c->nrows);
    const uint32_t nblk = (uint32_t)((len + kSortTile - 1) / kSortTile);
    const uint64_t ncounts = 256ull * std::max<uint32_t>(nblk, 1);

    DevBuf k0, k1, v0, v1, counts, sums, runsum, keep, total;
    uint32_t nnz = 0;
    if (len) {
        SPAL_HIP_TRY(k0.alloc(len * 8));
        SPAL_HIP_TRY(k1.alloc(len * 8));
        SPAL_HIP_TRY(v0.alloc(len * 4));
        SPAL_HIP_TRY(v1.alloc(len * 4));
        SPAL_HIP_TRY(counts.alloc(ncounts * 4));
        const uint64_t scan_n = std::max<uint64_t>(ncounts, len);
        SPAL_HIP_TRY(sums.alloc(((scan_n + kScanTile - 1) / kScanTile) * 4));
        SPAL_HIP_TRY(total.alloc(4));
        const uint32_t g256 = (uint32_t)((len + 255) / 256);
        hipLaunchKernelGGL(coo_make_keys, dim3(g256), dim3(256), 0, st, c->d_rows, c->d_cols,
                           k0.as<uint64_t>(), v0.as<uint32_t>(), len, cbits);
        uint64_t *ka = k0.as<uint64_t>(), *kb = k1.as<uint64_t>();
        uint32_t *va = v0.as<uint32_t>(), *vb = v1.as<uint32_t>();
        for (uint32_t shift = 0; shift < rbits + cbits; shift += 8) {
            hipLaunchKernelGGL(radix_hist, dim3(nblk), dim3(kSortThreads), 0, st, ka, len, shift,
                               counts.as<uint32_t>(), nblk);
            SPAL_HIP_TRY(exclusive_scan_u32(counts.as<uint32_t>(), counts.as<uint32_t>(), ncounts,
                                            sums.as<uint32_t>(), nullptr, st));
            hipLaunchKernelGGL(radix_scatter, dim3(nblk), dim3(kSortThreads), 0, st, ka, va, kb, vb,
                               len, shift, counts.as<uint32_t>(), nblk);
            std::swap(ka, kb);
            std::swap(va, vb);
        }
        SPAL_HIP_TRY(hipGetLastError());
        // runs -> sums -> keep flags (the scratch key / payload buffers are free again)
        SPAL_HIP_TRY(runsum.alloc(len * sizeof(T)));
        uint32_t *d_keep = vb;                 // reuse: payload scratch
        uint32_t *d_pos = reinterpret_cast<uint32_t *>(kb);  // reuse: key scratch (len*8 >= len*4)
        hipLaunchKernelGGL(coo_run_sums<T>, dim3(g256), dim3(256), 0, st, ka, va,
                           (const T *)c->d_vals, len, runsum.as<T>(), d_keep);
        SPAL_HIP_TRY(exclusive_scan_u32(d_keep, d_pos, len, sums.as<uint32_t>(),
                                        total.as<uint32_t>(), st));
        SPAL_HIP_TRY(hipMemcpyAsync(&nnz, total.p, 4, hipMemcpyDeviceToHost, st));
        SPAL_HIP_TRY(hipStreamSynchronize(st));  // the one data-dependent size

        DevBuf orow, ocol, oval, rowptr;
        SPAL_HIP_TRY(orow.alloc((size_t)nnz * 4));
        SPAL_HIP_TRY(ocol.alloc((size_t)nnz * 4));
        SPAL_HIP_TRY(oval.alloc((size_t)nnz * sizeof(T)));
        SPAL_HIP_TRY(rowptr.alloc((c->nrows + 1) * 4));
        hipLaunchKernelGGL(coo_compact<T>, dim3(g256), dim3(256), 0, st, ka, runsum.as<T>(), d_keep,
                           d_pos, len, cbits, orow.as<uint32_t>(), ocol.as<uint32_t>(), oval.as<T>());
        hipLaunchKernelGGL(coo_rowptr, dim3((uint32_t)((c->nrows + 1 + 255) / 256)), dim3(256), 0, st,
                           orow.as<uint32_t>(), nnz, (uint32_t)c->nrows, rowptr.as<uint32_t>());
        SPAL_HIP_TRY(hipGetLastError());
        SPAL_HIP_TRY(hipStreamSynchronize(st));
        spal_csr *a = nullptr;
        SPAL_TRY(csr_adopt_device(c->device, (int)sizeof(T), c->nrows, c->ncols, nnz, nnz,
                                  rowptr.as<uint32_t>(), ocol.as<uint32_t>(), oval.p, &a));
        rowptr.release(); ocol.release(); oval.release();
        *out = a;
        return SPAL_OK;
    }
    // no entries at all: an empty CSR matrix
    DevBuf rowptr, ocol, oval;
    SPAL_HIP_TRY(rowptr.alloc((c->nrows + 1) * 4));
    SPAL_HIP_TRY(ocol.alloc(4));
    SPAL_HIP_TRY(oval.alloc(sizeof(T)));
    SPAL_HIP_TRY(hipMemsetAsync(rowptr.p, 0, (c->nrows + 1) * 4, st));
    SPAL_HIP_TRY(hipStreamSynchronize(st));
    spal_csr *a = nullptr;
    SPAL_TRY(csr_adopt_device(c->device, (int)sizeof(T), c->nrows, c->ncols, 0, 0,
                              rowptr.as<uint32_t>(), ocol.as<uint32_t>(), oval.p, &a));
    rowptr.release(); ocol.release(); oval.release();
    *out = a;
    return SPAL_OK;
}

static void coo_free(spal_coo *c) {
    if (!c) return;
    (void)hipFree(c->d_rows);
    (void)hipFree(c->d_cols);
    (void)hipFree(c->d_vals);
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
    if (nrows > 0xffffffffull || ncols > 0xffffffffull || len >= 0xffffffffull)
        return fail(SPAL_ERR_UNSUPPORTED, "COO shape does not fit 32-bit device indices");
    // every entry inside the matrix (push asserts, src/coo.rs:432-433)
    std::vector<uint32_t> r32(len), c32(len);
    std::vector<int> bad(host_threads(), 0);
    parallel_for(len, [&](uint64_t b, uint64_t e, unsigned t) {
        for (uint64_t i = b; i < e; ++i) {
            if (rows[i] >= nrows || cols[i] >= ncols) { bad[t] = 1; return; }
            r32[i] = (uint32_t)rows[i];
            c32[i] = (uint32_t)cols[i];
        }
    });
    for (int f : bad)
        if (f) return fail(SPAL_ERR_INDEX_OUT_OF_BOUNDS,
                           "CooMatrix::push would panic: assertion failed: row < nrows && col < ncols");
    DeviceGuard guard(device);
    if (guard.status != SPAL_OK) return guard.status;
    spal_coo *c = new spal_coo;
    c->device = device; c->elem_size = (int)sizeof(T);
    c->nrows = nrows; c->ncols = ncols; c->len = len;
    hipError_t e = hipMalloc(&c->d_rows, std::max<uint64_t>(len, 1) * 4);
    if (e == hipSuccess) e = hipMalloc(&c->d_cols, std::max<uint64_t>(len, 1) * 4);
    if (e == hipSuccess) e = hipMalloc(&c->d_vals, std::max<uint64_t>(len, 1) * sizeof(T));
    if (e == hipSuccess && len) e = hipMemcpy(c->d_rows, r32.data(), len * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && len) e = hipMemcpy(c->d_cols, c32.data(), len * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && len) e = hipMemcpy(c->d_vals, vals, len * sizeof(T), hipMemcpyHostToDevice);
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
int spal_coo_assemble_csr(spal_coo_t c, void *stream, spal_csr_t *out) {
    if (!c || !out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_coo_assemble_csr: null argument");
    *out = nullptr;
    DeviceGuard guard(c->device);
    if (guard.status != SPAL_OK) return guard.status;
    return c->elem_size == 8 ? coo_assemble_t<double>(c, (hipStream_t)stream, out)
                             : coo_assemble_t<float>(c, (hipStream_t)stream, out);
}
int spal_coo_to_csr_f64(int device, uint64_t nrows, uint64_t ncols, uint64_t len, const uint64_t *rows,
                        const uint64_t *cols, const double *vals, spal_csr_t *out) {
    return coo_to_csr<double>(device, nrows, ncols, len, rows, cols, vals, out);
}
int spal_coo_to_csr_f32(int device, uint64_t nrows, uint64_t ncols, uint64_t len, const uint64_t *rows,
                        const uint64_t *cols, const float *vals, spal_csr_t *out) {
    return coo_to_csr<float>(device, nrows, ncols, len, rows, cols, vals, out);
}

}  // extern "C"
