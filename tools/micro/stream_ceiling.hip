// Micro-benchmark (development tool): what the HBM delivers for the FOOTPRINT of the CSR stream kernel
// at config 3 (1.12 GB of f64 values + 0.28 GB of 16-bit columns + 40 MB of row pointers + 80 MB of x read,
// 80 MB of y written), with no gather and no arithmetic -- the ceiling next to which the product kernel's
// time is read -- as a function of (a) the workgroups a CU holds (forced through the LDS allocation) and
// (b) how many tiles of loads a wave keeps in flight.  Also a plain 16-byte copy for the chip's copy rate.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/stream_ceiling.hip -o tools/micro/stream_ceiling
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) double f64x2;

__global__ __launch_bounds__(256) void copy16(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n; i += 4 * stride) {
        u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
              c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
        __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void read16(const u32x4 *__restrict__ src, uint32_t *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + 3 * stride < n; i += 4 * stride) {
        u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
              c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n; i += stride) acc ^= src[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;   // (never: keeps the loads alive)
}

// The stream kernel's loads for one tile of 64 rows x 14 entries: 7 x (16 B of values + 4 B of columns) per lane,
// 2 row pointers per lane.
struct Tile {
    f64x2 v[7];
    uint32_t c[7], r0, r1;
};
__device__ __forceinline__ void tile_load(Tile &t, const double *__restrict__ vals, const uint16_t *__restrict__ col16,
                                          const uint32_t *__restrict__ rowptr, uint32_t row, uint32_t lane) {
    const size_t e0 = (size_t)row * 14 + lane * 2;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        t.v[j] = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(vals + e0 + j * 128));
        t.c[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(col16 + e0 + j * 128));
    }
    t.r0 = rowptr[row + lane];
    t.r1 = rowptr[row + lane + 1];
}
__device__ __forceinline__ double tile_use(const Tile &t) {
    double s = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) s += t.v[j].x + t.v[j].y + (double)t.c[j];
    return s + (double)(t.r1 - t.r0);
}

// One workgroup (4 waves) per super-tile of 1024 rows; a wave owns 4 tiles of 64 rows; PF tiles of loads ahead.
// XWIN: stage 40 KB of x into LDS per super-tile (through L2, like the product kernel), else x is not read.
template <int PF, bool XWIN>
__global__ __launch_bounds__(256, 2) void footprint(const double *__restrict__ vals, const uint16_t *__restrict__ col16,
                                                    const uint32_t *__restrict__ rowptr, const double *__restrict__ x,
                                                    double *__restrict__ y, uint32_t nrows, uint32_t nblocks, uint32_t per_xcd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // per_xcd == 0: blocks in launch order (one front through memory); else every XCD a contiguous run of per_xcd blocks
    const uint32_t b = per_xcd ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if (b >= nblocks || (per_xcd && (blockIdx.x >> 3) >= per_xcd)) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t wrow = b * 1024 + wave * 256;
    Tile t[4];
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (wrow + p * 64 < nrows) tile_load(t[p], vals, col16, rowptr, wrow + p * 64, lane);
    if (XWIN) {
        u32x4 *d4 = reinterpret_cast<u32x4 *>(smem);
        const uint32_t c0 = b * 1024 > 2048 ? b * 1024 - 2048 : 0;
        const u32x4 *s4 = reinterpret_cast<const u32x4 *>(x + min(c0, nrows - 5120));
        u32x4 r[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) r[k] = s4[threadIdx.x + k * 256];
#pragma unroll
        for (int k = 0; k < 10; ++k) d4[threadIdx.x + k * 256] = r[k];
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t r0 = wrow + k * 64;
        if (r0 >= nrows) break;
        if (k + PF < 4 && r0 + PF * 64 < nrows) tile_load(t[(k + PF) & 3], vals, col16, rowptr, r0 + PF * 64, lane);
        double s = tile_use(t[k & 3]);
        if (XWIN) s += reinterpret_cast<const double *>(smem)[(lane * 37 + k) & 4095];
        if (r0 + lane < nrows) y[r0 + lane] = s;
    }
}


// footprint with the 8 XCDs interleaved in chunks of C consecutive super-tiles (XCD x takes chunks x, x + 8, ...): all
// XCDs stay inside one moving window of 8 * C super-tiles instead of walking 8 runs an eighth of the arrays apart.
template <int PF, bool XWIN>
__global__ __launch_bounds__(256, 2) void footprint_chunk(const double *__restrict__ vals, const uint16_t *__restrict__ col16,
                                                          const uint32_t *__restrict__ rowptr, const double *__restrict__ x,
                                                          double *__restrict__ y, uint32_t nrows, uint32_t nblocks, uint32_t C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t b = ((slot / C) * 8u + xcd) * C + slot % C;
    if (b >= nblocks) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t wrow = b * 1024 + wave * 256;
    Tile t[4];
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (wrow + p * 64 < nrows) tile_load(t[p], vals, col16, rowptr, wrow + p * 64, lane);
    if (XWIN) {
        u32x4 *d4 = reinterpret_cast<u32x4 *>(smem);
        const uint32_t c0 = b * 1024 > 2048 ? b * 1024 - 2048 : 0;
        const u32x4 *s4 = reinterpret_cast<const u32x4 *>(x + min(c0, nrows - 5120));
        u32x4 r[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) r[k] = s4[threadIdx.x + k * 256];
#pragma unroll
        for (int k = 0; k < 10; ++k) d4[threadIdx.x + k * 256] = r[k];
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t r0 = wrow + k * 64;
        if (r0 >= nrows) break;
        if (k + PF < 4 && r0 + PF * 64 < nrows) tile_load(t[(k + PF) & 3], vals, col16, rowptr, r0 + PF * 64, lane);
        double s = tile_use(t[k & 3]);
        if (XWIN) s += reinterpret_cast<const double *>(smem)[(lane * 37 + k) & 4095];
        if (r0 + lane < nrows) y[r0 + lane] = s;
    }
}


// Which second stream does the values array's "class" need?  The chunk-interleaved footprint (C = 64) with parts
// switched off: MASK bit 0 = the 16-bit columns are read, bit 1 = the row pointers, bit 2 = y is written, bit 3 = the
// x window is staged.
template <int MASK, int YMODE = 1>   // YMODE: how y is stored -- 1 plain, 2 non-temporal, 3 agent-scope atomic store (written through), 4 the super-tile's 1024 rows in one burst out of LDS
__global__ __launch_bounds__(256, 2) void footprint_parts(const double *__restrict__ vals, const uint16_t *__restrict__ col16,
                                                          const uint32_t *__restrict__ rowptr, const double *__restrict__ x,
                                                          double *__restrict__ y, uint32_t nrows, uint32_t nblocks, uint32_t C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t b = ((slot / C) * 8u + xcd) * C + slot % C;
    if (b >= nblocks) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t wrow = b * 1024 + wave * 256;
    f64x2 v[2][7];
    uint32_t c[2][7], r0v[2], r1v[2];
    auto load = [&](int s, uint32_t row) {
        const size_t e0 = (size_t)row * 14 + lane * 2;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            v[s][j] = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(vals + e0 + j * 128));
            c[s][j] = (MASK & 1) ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(col16 + e0 + j * 128)) : 0u;
        }
        r0v[s] = (MASK & 2) ? rowptr[row + lane] : 0u;
        r1v[s] = (MASK & 2) ? rowptr[row + lane + 1] : 1u;
    };
    load(0, wrow);
    if (MASK & 8) {
        u32x4 *d4 = reinterpret_cast<u32x4 *>(smem);
        const uint32_t c0 = b * 1024 > 2048 ? b * 1024 - 2048 : 0;
        const u32x4 *s4 = reinterpret_cast<const u32x4 *>(x + min(c0, nrows - 5120));
        u32x4 r[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) r[k] = s4[threadIdx.x + k * 256];
#pragma unroll
        for (int k = 0; k < 10; ++k) d4[threadIdx.x + k * 256] = r[k];
        __syncthreads();
    }
    double keep = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t r0 = wrow + k * 64;
        if (r0 >= nrows) break;
        if (k + 1 < 4 && r0 + 64 < nrows) load((k + 1) & 1, r0 + 64);
        double s = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) s += v[k & 1][j].x + v[k & 1][j].y + (double)c[k & 1][j];
        s += (double)(r1v[k & 1] - r0v[k & 1]);
        if (MASK & 8) s += reinterpret_cast<const double *>(smem)[(lane * 37 + k) & 4095];
        if (MASK & 4) {
            if (r0 + lane < nrows) {
                if (YMODE == 1) y[r0 + lane] = s;
                else if (YMODE == 2) __builtin_nontemporal_store(s, y + r0 + lane);
                else if (YMODE == 3) __hip_atomic_store(y + r0 + lane, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else reinterpret_cast<double *>(smem + 40960)[wave * 256 + k * 64 + lane] = s;
            }
        } else keep += s;
    }
    if ((MASK & 4) && YMODE == 4) {
        __syncthreads();
        const f64x2 *src = reinterpret_cast<const f64x2 *>(smem + 40960);
        f64x2 *dst = reinterpret_cast<f64x2 *>(y + (size_t)b * 1024);
        for (uint32_t i = threadIdx.x; i < 512 && (size_t)b * 1024 + i * 2 + 1 < nrows; i += 256) dst[i] = src[i];
    }
    if (!(MASK & 4) && keep == 1.2345e300) y[0] = keep;
}


// The footprint with values and 16-bit columns MERGED into one array: per step of 128 entries 1024 bytes of values
// followed by 256 bytes of columns (one stream instead of two).  MASK as footprint_parts (bit 0 is implied).
template <int MASK>
__global__ __launch_bounds__(256, 2) void footprint_merged(const unsigned char *__restrict__ vc, const uint32_t *__restrict__ rowptr,
                                                           const double *__restrict__ x, double *__restrict__ y, uint32_t nrows,
                                                           uint32_t nblocks, uint32_t C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t b = ((slot / C) * 8u + xcd) * C + slot % C;
    if (b >= nblocks) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t wrow = b * 1024 + wave * 256;
    f64x2 v[2][7];
    uint32_t c[2][7], r0v[2], r1v[2];
    auto load = [&](int s, uint32_t row) {
        const size_t e0 = (size_t)row * 14 + lane * 2;            // row * 14 is a multiple of 2; 64 rows = 7 steps
        const unsigned char *p = vc + (e0 >> 7) * 1280 + (e0 & 127) * 8;
        const unsigned char *q = vc + (e0 >> 7) * 1280 + 1024 + (e0 & 127) * 2;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            v[s][j] = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p + j * 1280));
            c[s][j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(q + j * 1280));
        }
        r0v[s] = (MASK & 2) ? rowptr[row + lane] : 0u;
        r1v[s] = (MASK & 2) ? rowptr[row + lane + 1] : 1u;
    };
    load(0, wrow);
    if (MASK & 8) {
        u32x4 *d4 = reinterpret_cast<u32x4 *>(smem);
        const uint32_t c0 = b * 1024 > 2048 ? b * 1024 - 2048 : 0;
        const u32x4 *s4 = reinterpret_cast<const u32x4 *>(x + min(c0, nrows - 5120));
        u32x4 r[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) r[k] = s4[threadIdx.x + k * 256];
#pragma unroll
        for (int k = 0; k < 10; ++k) d4[threadIdx.x + k * 256] = r[k];
        __syncthreads();
    }
    double keep = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t r0 = wrow + k * 64;
        if (r0 >= nrows) break;
        if (k + 1 < 4 && r0 + 64 < nrows) load((k + 1) & 1, r0 + 64);
        double s = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) s += v[k & 1][j].x + v[k & 1][j].y + (double)c[k & 1][j];
        s += (double)(r1v[k & 1] - r0v[k & 1]);
        if (MASK & 8) s += reinterpret_cast<const double *>(smem)[(lane * 37 + k) & 4095];
        if (MASK & 4) { if (r0 + lane < nrows) y[r0 + lane] = s; }
        else keep += s;
    }
    if (!(MASK & 4) && keep == 1.2345e300) y[0] = keep;
}

// The same loads, but a fixed grid of workgroups each walking a contiguous run of super-tiles (what the sliding
// kernel does), two tiles ahead across super-tile boundaries, no x window: does the walk itself cost bandwidth?
template <int PF>
__global__ __launch_bounds__(256, 2) void footprint_walk(const double *__restrict__ vals, const uint16_t *__restrict__ col16,
                                                         const uint32_t *__restrict__ rowptr, double *__restrict__ y,
                                                         uint32_t nrows, uint32_t nsteps, uint32_t per_xcd, uint32_t chunk,
                                                         uint32_t G) {
    // G workgroups share a run of G * chunk steps: member m takes its steps m, m + G, m + 2G, ... (G = 1: a run each)
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, grp = slot / G, mem = slot % G;
    const uint32_t run_end = min((xcd + 1u) * per_xcd, nsteps);
    const uint32_t g0 = xcd * per_xcd + grp * chunk * G;
    if (g0 + mem >= run_end) return;
    const uint32_t g1 = min(g0 + chunk * G, run_end);
    const uint32_t i0 = 0, i1 = (g1 - g0 - mem + G - 1u) / G;      // this member's steps, numbered 0 ...
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    Tile t[PF + 1];
    auto gstep = [&](uint32_t i) { return g0 + mem + min(i, i1 - 1u) * G; };
    auto row_of = [&](uint32_t i) { return min((gstep(i) * 4u + wave) * 64u, nrows - 64u); };
#pragma unroll
    for (int p = 0; p < PF; ++p) tile_load(t[p], vals, col16, rowptr, row_of(i0 + p), lane);
    auto step = [&](uint32_t i, auto kc) {
        constexpr int K = decltype(kc)::value;
        tile_load(t[(K + PF) % (PF + 1)], vals, col16, rowptr, row_of(i + PF), lane);
        const double s = tile_use(t[K]);
        const uint32_t r0 = (gstep(i) * 4u + wave) * 64u;
        if (r0 + lane < nrows) y[r0 + lane] = s;
    };
    for (uint32_t i = i0; i < i1; i += PF + 1) {
        step(i, std::integral_constant<int, 0>{});
        if (i + 1 < i1) step(i + 1, std::integral_constant<int, 1>{});
        if constexpr (PF >= 2) if (i + 2 < i1) step(i + 2, std::integral_constant<int, 2>{});
    }
}

template <typename F>
static double time_us(F launch, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms * 1e3 / iters;
}

template <int PF, bool XWIN>
static void run_footprint(const char *name, size_t lds, const double *vals, const uint16_t *col16, const uint32_t *rowptr,
                          const double *x, double *y, uint32_t nrows) {
    const uint32_t nblocks = (nrows + 1023) / 1024, per_xcd = (nblocks + 7) / 8;
    auto kern = footprint<PF, XWIN>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const double us = time_us([&] {
        hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(256), lds, 0, vals, col16, rowptr, x, y, nrows, nblocks, per_xcd);
    }, 50);
    const double bytes = (double)nrows * 14 * 10 + 4.0 * nrows + 8.0 * nrows + (XWIN ? 8.0 * nrows : 0.0);
    printf("footprint %-34s lds %6zu B  %7.1f us  %7.1f GB/s moved (HBM-level bytes %.3f GB)\n", name, lds, us, bytes / us / 1e3,
           bytes / 1e9);
    fflush(stdout);
}

// --place: does the time depend on WHICH allocations the two big streams live in?  6 value arrays x 6 column arrays
// (odd-sized dummies between them), every pair timed.
static int place_main() {
    const uint32_t nrows = 10'000'000;
    const size_t nnz = (size_t)nrows * 14 + 4096;
    const int K = 6;
    double *vals[K], *x, *y;
    uint16_t *col16[K];
    uint32_t *rowptr;
    void *dummy[2 * K];
    for (int i = 0; i < K; ++i) {
        CK(hipMalloc(&vals[i], nnz * 8)); CK(hipMemset(vals[i], 1, nnz * 8));
        CK(hipMalloc(&dummy[2 * i], (size_t)(i + 1) * 7'654'321));
        CK(hipMalloc(&col16[i], nnz * 2)); CK(hipMemset(col16[i], 1, nnz * 2));
        CK(hipMalloc(&dummy[2 * i + 1], (size_t)(i + 3) * 3'456'789));
    }
    CK(hipMalloc(&rowptr, ((size_t)nrows + 65) * 4)); CK(hipMemset(rowptr, 0, ((size_t)nrows + 65) * 4));
    CK(hipMalloc(&x, (size_t)nrows * 8)); CK(hipMalloc(&y, (size_t)nrows * 8));
    CK(hipMemset(x, 0, (size_t)nrows * 8));
    const uint32_t nblocks = (nrows + 1023) / 1024, per_xcd = (nblocks + 7) / 8;
    auto kern = footprint<2, true>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        printf("rep %d: rows = value array %%p..., columns = col16 array; us per launch\n", rep);
        for (int i = 0; i < K; ++i) {
            printf("vals %p:", (void *)vals[i]);
            for (int j = 0; j < K; ++j) {
                const double us = time_us([&] {
                    hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(256), 72 * 1024, 0, vals[i], col16[j], rowptr, x, y, nrows, nblocks, per_xcd);
                }, 20);
                printf(" %6.1f", us);
            }
            printf("\n");
            fflush(stdout);
        }
    }
    for (int j = 0; j < K; ++j) printf("col16[%d] %p\n", j, (void *)col16[j]);
    return 0;
}

// --stride: for each of 5 value arrays, the time as a function of the blocks per XCD run (i.e. of the distance
// between the 8 streams that walk the array), and with no XCD runs at all.
static int stride_main() {
    const uint32_t nrows = 10'000'000;
    const size_t nnz = (size_t)nrows * 14 + 4096;
    const int K = 5;
    double *vals[K], *x, *y;
    uint16_t *col16;
    uint32_t *rowptr;
    void *dummy[K];
    for (int i = 0; i < K; ++i) {
        CK(hipMalloc(&vals[i], nnz * 8 + (32u << 20))); CK(hipMemset(vals[i], 1, nnz * 8));
        CK(hipMalloc(&dummy[i], (size_t)(i + 1) * 7'654'321));
    }
    CK(hipMalloc(&col16, nnz * 2)); CK(hipMemset(col16, 1, nnz * 2));
    CK(hipMalloc(&rowptr, ((size_t)nrows + 65) * 4)); CK(hipMemset(rowptr, 0, ((size_t)nrows + 65) * 4));
    CK(hipMalloc(&x, (size_t)nrows * 8)); CK(hipMalloc(&y, (size_t)nrows * 8));
    CK(hipMemset(x, 0, (size_t)nrows * 8));
    const uint32_t nblocks = (nrows + 1023) / 1024, base = (nblocks + 7) / 8;
    auto kern = footprint<2, true>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int deltas[] = {0, 1, 2, 3, 4, 6, 8, 16, 32, 64, -1};
    printf("columns: per-XCD run = ceil(nblocks / 8) + {0, 1, 2, 3, 4, 6, 8, 16, 32, 64} blocks, then launch order (no runs);"
           " then the array shifted by 4 KB, 64 KB, 1 MB, 16 MB at run + 0\n");
    for (int i = 0; i < K; ++i) {
        printf("vals %p:", (void *)vals[i]);
        for (int dl : deltas) {
            const uint32_t per_xcd = dl < 0 ? 0u : base + (uint32_t)dl;
            const uint32_t grid = dl < 0 ? nblocks : per_xcd * 8;
            const double us = time_us([&] {
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 72 * 1024, 0, vals[i], col16, rowptr, x, y, nrows, nblocks, per_xcd);
            }, 20);
            printf(" %6.1f", us);
        }
        printf("  |");
        for (size_t shift : {(size_t)4096, (size_t)65536, (size_t)1 << 20, (size_t)16 << 20}) {
            const double *v = (const double *)((const char *)vals[i] + shift);
            const double us = time_us([&] {
                hipLaunchKernelGGL(kern, dim3(base * 8), dim3(256), 72 * 1024, 0, v, col16, rowptr, x, y, nrows, nblocks, base);
            }, 20);
            printf(" %6.1f", us);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}

// --quick <nrows> <per_row>: one JSON line for bench.py -- the rate at which the HBM delivers the stream kernel's
// footprint for that shape (f64 values, 16-bit columns, x staged through LDS, y written; rows of per_row <= 14 * k
// are approximated by the byte count: the kernel below is the 14-per-row one scaled to the same bytes), and the
// plain 16-byte copy / read rates.
static int quick_main(uint32_t nrows_req, uint32_t per_row) {
    // same bytes as the request: the footprint kernel streams 14 entries per row
    const uint64_t want_entries = (uint64_t)nrows_req * per_row;
    const uint32_t nrows = (uint32_t)std::min<uint64_t>(want_entries / 14, 40'000'000ull);
    const size_t nnz = (size_t)nrows * 14 + 4096;
    double *vals, *x, *y;
    uint16_t *col16;
    uint32_t *rowptr;
    CK(hipMalloc(&vals, nnz * 8)); CK(hipMalloc(&col16, nnz * 2)); CK(hipMalloc(&rowptr, ((size_t)nrows + 65) * 4));
    CK(hipMalloc(&x, (size_t)nrows * 8 + 65536)); CK(hipMalloc(&y, (size_t)nrows * 8));
    CK(hipMemset(vals, 1, nnz * 8)); CK(hipMemset(col16, 1, nnz * 2)); CK(hipMemset(rowptr, 0, ((size_t)nrows + 65) * 4));
    CK(hipMemset(x, 0, (size_t)nrows * 8 + 65536));
    const uint32_t nblocks = (nrows + 1023) / 1024, per_xcd = (nblocks + 7) / 8;
    auto kern = footprint<2, true>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep)
        best = std::min(best, time_us([&] {
            hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(256), 72 * 1024, 0, vals, col16, rowptr, x, y, nrows, nblocks, per_xcd);
        }, 30));
    const double bytes = (double)nrows * 14 * 10 + 4.0 * nrows + 8.0 * nrows + 8.0 * nrows;
    const size_t n16 = nnz * 8 / 16;
    double copy_us = 1e30, read_us = 1e30;
    u32x4 *b;
    uint32_t *flag;
    CK(hipMalloc(&b, n16 * 16)); CK(hipMalloc(&flag, 4));
    for (int rep = 0; rep < 2; ++rep) {
        copy_us = std::min(copy_us, time_us([&] { hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, (const u32x4 *)vals, b, n16); }, 10));
        read_us = std::min(read_us, time_us([&] { hipLaunchKernelGGL(read16, dim3(8192), dim3(256), 0, 0, (const u32x4 *)vals, flag, n16); }, 10));
    }
    printf("{\"footprint_us\": %.2f, \"footprint_bytes\": %.0f, \"footprint_gbs\": %.1f, \"copy16_gbs\": %.1f, \"read16_gbs\": %.1f, "
           "\"rows\": %u}\n", best, bytes, bytes / best / 1e3, 2.0 * n16 * 16 / copy_us / 1e3, 1.0 * n16 * 16 / read_us / 1e3, nrows);
    return 0;
}

// --walk: the fixed-grid walk against one workgroup per super-tile, same arrays
static int walk_main() {
    const uint32_t nrows = 10'000'000;
    const size_t nnz = (size_t)nrows * 14 + 4096;
    double *vals, *x, *y;
    uint16_t *col16;
    uint32_t *rowptr;
    CK(hipMalloc(&vals, nnz * 8)); CK(hipMalloc(&col16, nnz * 2)); CK(hipMalloc(&rowptr, ((size_t)nrows + 65) * 4));
    CK(hipMalloc(&x, (size_t)nrows * 8)); CK(hipMalloc(&y, (size_t)nrows * 8));
    CK(hipMemset(vals, 1, nnz * 8)); CK(hipMemset(col16, 1, nnz * 2)); CK(hipMemset(rowptr, 0, ((size_t)nrows + 65) * 4));
    CK(hipMemset(x, 0, (size_t)nrows * 8));
    const uint32_t nblocks = (nrows + 1023) / 1024, per_xcd_b = (nblocks + 7) / 8;
    const uint32_t nsteps = (nrows + 255) / 256, per_xcd = (nsteps + 7) / 8;
    auto k1 = footprint<2, false>;
    auto kw = footprint_walk<2>;
    CK(hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)kw, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < 3; ++rep) {
        double us = time_us([&] { hipLaunchKernelGGL(k1, dim3(per_xcd_b * 8), dim3(256), 72 * 1024, 0, vals, col16, rowptr, x, y, nrows, nblocks, per_xcd_b); }, 30);
        printf("one workgroup per super-tile (9766 workgroups)      %7.1f us\n", us);
        for (uint32_t grid : {512u, 1024u}) {
            for (uint32_t G : {1u, 2u, 4u, 8u, 16u, 64u}) {
                const uint32_t slots = grid / 8, chunk = (per_xcd + slots - 1) / slots;
                us = time_us([&] { hipLaunchKernelGGL(kw, dim3(slots * 8), dim3(256), 72 * 1024, 0, vals, col16, rowptr, y, nrows, nsteps, per_xcd, chunk, G); }, 30);
                printf("walk: %5u workgroups x %4u steps of 256 rows, %2u workgroups interleaved per run   %7.1f us\n", slots * 8, chunk, G, us);
            }
        }
        fflush(stdout);
    }
    return 0;
}


// --map [gb]: a speed map of the device's memory.  Value arrays of 1.12 GB are allocated one after the other until
// `gb` GB are held (default: 85 % of the free memory), each is timed as the VALUES stream of the config-3 footprint
// (columns, row pointers, x, y fixed) and with a plain read; then everything is freed, allocated again in the same
// order and timed again (does the pattern belong to the position in the allocation order, i.e. to the physical place?).
// For the slowest and the fastest array: the footprint over each eighth of the rows alone (512 MB flushed through the
// caches in between), i.e. whether a slow array is slow everywhere.
static int map_main(double gb) {
    const uint32_t nrows = 10'000'000;
    const size_t nnz = (size_t)nrows * 14 + 4096;
    size_t fre = 0, tot = 0;
    CK(hipMemGetInfo(&fre, &tot));
    double *x, *y;
    uint16_t *col16;
    uint32_t *rowptr, *flag;
    u32x4 *flush;
    const size_t flush_n16 = (size_t)(512u << 20) / 16;
    CK(hipMalloc(&col16, nnz * 2)); CK(hipMemset(col16, 1, nnz * 2));
    CK(hipMalloc(&rowptr, ((size_t)nrows + 65) * 4)); CK(hipMemset(rowptr, 0, ((size_t)nrows + 65) * 4));
    CK(hipMalloc(&x, (size_t)nrows * 8)); CK(hipMalloc(&y, (size_t)nrows * 8)); CK(hipMalloc(&flag, 64));
    CK(hipMalloc(&flush, flush_n16 * 16)); CK(hipMemset(flush, 0, flush_n16 * 16));
    CK(hipMemset(x, 0, (size_t)nrows * 8));
    if (gb <= 0) gb = fre * 0.85 / 1e9;
    const int K = (int)std::min<double>(gb * 1e9 / (nnz * 8.0), 260.0);
    printf("free %.1f GB of %.1f GB; %d value arrays of %.2f GB\n", fre / 1e9, tot / 1e9, K, nnz * 8 / 1e9);
    const uint32_t nblocks = (nrows + 1023) / 1024, per_xcd = (nblocks + 7) / 8;
    auto kern = footprint<2, true>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<double *> vals(K, nullptr);
    std::vector<double> us_fp(K), us_rd(K);
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < K; ++i) { CK(hipMalloc(&vals[i], nnz * 8)); }
        for (int i = 0; i < K; ++i) CK(hipMemsetAsync(vals[i], 1, nnz * 8, 0));
        CK(hipDeviceSynchronize());
        printf("pass %d: index  address         footprint us   read16 us\n", pass);
        for (int i = 0; i < K; ++i) {
            us_fp[i] = time_us([&] {
                hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(256), 72 * 1024, 0, vals[i], col16, rowptr, x, y, nrows, nblocks, per_xcd);
            }, 12);
            us_rd[i] = time_us([&] { hipLaunchKernelGGL(read16, dim3(8192), dim3(256), 0, 0, (const u32x4 *)vals[i], flag, nnz * 8 / 16); }, 8);
            printf("  %3d  %p  %7.1f  %7.1f\n", i, (void *)vals[i], us_fp[i], us_rd[i]);
            fflush(stdout);
        }
        int lo = 0, hi = 0;
        for (int i = 0; i < K; ++i) { if (us_fp[i] < us_fp[lo]) lo = i; if (us_fp[i] > us_fp[hi]) hi = i; }
        printf("pass %d: fastest %d (%.1f us), slowest %d (%.1f us)\n", pass, lo, us_fp[lo], hi, us_fp[hi]);
        for (int which : {lo, hi}) {
            printf("  array %d by eighths of the rows (us):", which);
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int s8 = 0; s8 < 8; ++s8) {
                const uint32_t r0 = (nrows / 8 / 1024) * 1024 * s8, nr = (nrows / 8 / 1024) * 1024;
                const uint32_t nb = nr / 1024, px = (nb + 7) / 8;
                double sum = 0;
                for (int it = 0; it < 6; ++it) {
                    hipLaunchKernelGGL(read16, dim3(8192), dim3(256), 0, 0, (const u32x4 *)flush, flag, flush_n16);
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(kern, dim3(px * 8), dim3(256), 72 * 1024, 0, vals[which] + (size_t)r0 * 14, col16 + (size_t)r0 * 14,
                                       rowptr + r0, x + r0, y + r0, nr, nb, px);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (it) sum += ms * 1e3;
                }
                printf(" %6.1f", sum / 5);
            }
            printf("\n");
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        for (int i = 0; i < K; ++i) CK(hipFree(vals[i]));
        fflush(stdout);
    }
    return 0;
}

// --map-vmm <align_mb> <offset_mb> <piece_mb> [gb]: the same map with the value arrays built through the virtual-memory
// API: virtual range reserved at `align_mb` alignment, the array mapped at + offset_mb, physical memory created in
// pieces of piece_mb (0 = one handle for the whole array; pieces are power-of-two sized, so the buddy allocator hands
// out naturally aligned blocks).  Theory under test: the class of an array is the alignment of (virtual - physical)
// address, i.e. the largest page-table fragment the driver can use for it.
static int map_vmm_main(size_t align_mb, size_t offset_mb, size_t piece_mb, double gb) {
    const uint32_t nrows = 10'000'000;
    const size_t nnz = (size_t)nrows * 14 + 4096;
    size_t fre = 0, tot = 0;
    CK(hipMemGetInfo(&fre, &tot));
    double *x, *y;
    uint16_t *col16;
    uint32_t *rowptr, *flag;
    CK(hipMalloc(&col16, nnz * 2)); CK(hipMemset(col16, 1, nnz * 2));
    CK(hipMalloc(&rowptr, ((size_t)nrows + 65) * 4)); CK(hipMemset(rowptr, 0, ((size_t)nrows + 65) * 4));
    CK(hipMalloc(&x, (size_t)nrows * 8)); CK(hipMalloc(&y, (size_t)nrows * 8)); CK(hipMalloc(&flag, 64));
    CK(hipMemset(x, 0, (size_t)nrows * 8));
    if (gb <= 0) gb = 100;
    const size_t MB = (size_t)1 << 20;
    const size_t piece = piece_mb ? piece_mb * MB : ((nnz * 8 + 2 * MB - 1) / (2 * MB)) * (2 * MB);
    const size_t mapped = ((nnz * 8 + piece - 1) / piece) * piece;
    const int K = (int)std::min<double>(gb * 1e9 / (double)mapped, 260.0);
    printf("vmm: alignment %zu MB, offset %zu MB, pieces of %zu MB (%zu per array), %d arrays\n", align_mb, offset_mb, piece / MB,
           mapped / piece, K);
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof acc);
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const uint32_t nblocks = (nrows + 1023) / 1024, per_xcd = (nblocks + 7) / 8;
    auto kern = footprint<2, true>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<void *> bases(K, nullptr);
    std::vector<std::vector<hipMemGenericAllocationHandle_t>> handles(K);
    const size_t reserve = mapped + offset_mb * MB;
    for (int i = 0; i < K; ++i) {
        CK(hipMemAddressReserve(&bases[i], reserve, align_mb * MB, nullptr, 0));
        char *at = (char *)bases[i] + offset_mb * MB;
        for (size_t off = 0; off < mapped; off += piece) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, piece, &prop, 0));
            handles[i].push_back(h);
            CK(hipMemMap(at + off, piece, 0, h, 0));
        }
        CK(hipMemSetAccess(at, mapped, &acc, 1));
        CK(hipMemsetAsync(at, 1, nnz * 8, 0));
    }
    CK(hipDeviceSynchronize());
    double lo = 1e9, hi = 0;
    for (int i = 0; i < K; ++i) {
        double *v = (double *)((char *)bases[i] + offset_mb * MB);
        const double us = time_us([&] {
            hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(256), 72 * 1024, 0, v, col16, rowptr, x, y, nrows, nblocks, per_xcd);
        }, 12);
        const double rd = time_us([&] { hipLaunchKernelGGL(read16, dim3(8192), dim3(256), 0, 0, (const u32x4 *)v, flag, nnz * 8 / 16); }, 8);
        printf("  %3d  %p  %7.1f  %7.1f\n", i, (void *)v, us, rd);
        lo = std::min(lo, us); hi = std::max(hi, us);
        fflush(stdout);
    }
    printf("vmm: alignment %zu MB, offset %zu MB, pieces %zu MB: footprint %.1f ... %.1f us over %d arrays\n", align_mb, offset_mb,
           piece / MB, lo, hi, K);
    for (int i = 0; i < K; ++i) {
        CK(hipMemUnmap((char *)bases[i] + offset_mb * MB, mapped));
        for (auto h : handles[i]) CK(hipMemRelease(h));
        CK(hipMemAddressFree(bases[i], reserve));
    }
    return 0;
}

// --map-chunk [gb]: per value array (hipMalloc, as --map): the footprint with 8 XCD runs (the library's order), with the
// XCDs interleaved in chunks of C = 1, 4, 16, 64, 256 super-tiles, and the sliding kernel's walk (512 workgroups, a run each).
static int map_chunk_main(double gb) {
    const uint32_t nrows = 10'000'000;
    const size_t nnz = (size_t)nrows * 14 + 4096;
    double *x, *y;
    uint16_t *col16;
    uint32_t *rowptr;
    CK(hipMalloc(&col16, nnz * 2)); CK(hipMemset(col16, 1, nnz * 2));
    CK(hipMalloc(&rowptr, ((size_t)nrows + 65) * 4)); CK(hipMemset(rowptr, 0, ((size_t)nrows + 65) * 4));
    CK(hipMalloc(&x, (size_t)nrows * 8)); CK(hipMalloc(&y, (size_t)nrows * 8));
    CK(hipMemset(x, 0, (size_t)nrows * 8));
    if (gb <= 0) gb = 60;
    const int K = (int)std::min<double>(gb * 1e9 / (nnz * 8.0), 260.0);
    const uint32_t nblocks = (nrows + 1023) / 1024, per_xcd = (nblocks + 7) / 8;
    const uint32_t nsteps = (nrows + 255) / 256, per_xcd_s = (nsteps + 7) / 8;
    auto k0 = footprint<2, true>;
    auto kc = footprint_chunk<2, true>;
    auto kw = footprint_walk<2>;
    CK(hipFuncSetAttribute((const void *)k0, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)kc, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)kw, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<double *> vals(K, nullptr);
    for (int i = 0; i < K; ++i) { CK(hipMalloc(&vals[i], nnz * 8)); CK(hipMemsetAsync(vals[i], 1, nnz * 8, 0)); }
    CK(hipDeviceSynchronize());
    printf("index  8 runs | chunks of 1, 4, 16, 64, 256 | walk 512 x run, 512 x (8 interleaved), 512 x (64 interleaved)   (us)\n");
    for (int i = 0; i < K; ++i) {
        printf("  %3d  %6.1f |", i, time_us([&] {
            hipLaunchKernelGGL(k0, dim3(per_xcd * 8), dim3(256), 72 * 1024, 0, vals[i], col16, rowptr, x, y, nrows, nblocks, per_xcd); }, 12));
        for (uint32_t C : {1u, 4u, 16u, 64u, 256u})
            printf(" %6.1f", time_us([&] {
                hipLaunchKernelGGL(kc, dim3(per_xcd * 8 + 8 * C), dim3(256), 72 * 1024, 0, vals[i], col16, rowptr, x, y, nrows, nblocks, C); }, 12));
        printf(" |");
        for (uint32_t G : {1u, 8u, 64u}) {
            const uint32_t slots = 64, chunk = (per_xcd_s + slots - 1) / slots;
            printf(" %6.1f", time_us([&] {
                hipLaunchKernelGGL(kw, dim3(slots * 8), dim3(256), 72 * 1024, 0, vals[i], col16, rowptr, y, nrows, nsteps, per_xcd_s, chunk, G); }, 12));
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}

template <int MASK, int YMODE = 1>
static double parts_us(const double *vals, const uint16_t *col16, const uint32_t *rowptr, const double *x, double *y, uint32_t nrows) {
    const uint32_t nblocks = (nrows + 1023) / 1024, per_xcd = (nblocks + 7) / 8, C = 64;
    auto k = footprint_parts<MASK, YMODE>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return time_us([&] { hipLaunchKernelGGL(k, dim3(per_xcd * 8 + 8 * C), dim3(256), 72 * 1024, 0, vals, col16, rowptr, x, y, nrows, nblocks, C); }, 12);
}
// --map-parts [gb]: per value array, the chunk-interleaved footprint with its parts switched on one by one; then, for
// the first slow and the first fast array, the full footprint against 6 different allocations of y / of the columns.
static int map_parts_main(double gb, int method) {   // method 0 hipMalloc, 1 hipExtMallocWithFlags(Uncached), 2 (Finegrained), 3 (Contiguous)
    const uint32_t nrows = 10'000'000;
    const size_t nnz = (size_t)nrows * 14 + 4096;
    double *x, *y;
    uint16_t *col16;
    uint32_t *rowptr;
    CK(hipMalloc(&col16, nnz * 2)); CK(hipMemset(col16, 1, nnz * 2));
    CK(hipMalloc(&rowptr, ((size_t)nrows + 65) * 4)); CK(hipMemset(rowptr, 0, ((size_t)nrows + 65) * 4));
    CK(hipMalloc(&x, (size_t)nrows * 8)); CK(hipMalloc(&y, (size_t)nrows * 8));
    CK(hipMemset(x, 0, (size_t)nrows * 8));
    if (gb <= 0) gb = 60;
    const int K = (int)std::min<double>(gb * 1e9 / (nnz * 8.0), 260.0);
    std::vector<double *> vals(K, nullptr);
    for (int i = 0; i < K; ++i) {
        if (method == 0) CK(hipMalloc(&vals[i], nnz * 8));
        else CK(hipExtMallocWithFlags((void **)&vals[i], nnz * 8, method == 1 ? hipDeviceMallocUncached : method == 2 ? hipDeviceMallocFinegrained : hipDeviceMallocContiguous));
        CK(hipMemsetAsync(vals[i], 1, nnz * 8, 0));
    }
    CK(hipDeviceSynchronize());
    printf("values arrays: %s\n", method == 0 ? "hipMalloc" : method == 1 ? "hipExtMallocWithFlags(Uncached)" : method == 2 ? "hipExtMallocWithFlags(Finegrained)" : "hipExtMallocWithFlags(Contiguous)");
    printf("index  values only | + columns | + row pointers | + y written | + x window (all) | values + y | values + x window  (us)\n");
    std::vector<double> full(K);
    for (int i = 0; i < K; ++i) {
        const double a = parts_us<0>(vals[i], col16, rowptr, x, y, nrows), b = parts_us<1>(vals[i], col16, rowptr, x, y, nrows),
                     c = parts_us<3>(vals[i], col16, rowptr, x, y, nrows), d = parts_us<7>(vals[i], col16, rowptr, x, y, nrows),
                     e = parts_us<15>(vals[i], col16, rowptr, x, y, nrows), f = parts_us<4>(vals[i], col16, rowptr, x, y, nrows),
                     g = parts_us<8>(vals[i], col16, rowptr, x, y, nrows);
        full[i] = e;
        printf("  %3d  %6.1f  %6.1f  %6.1f  %6.1f  %6.1f  |  %6.1f  %6.1f\n", i, a, b, c, d, e, f, g);
        fflush(stdout);
    }
    int lo = 0, hi = 0;
    for (int i = 0; i < K; ++i) { if (full[i] < full[lo]) lo = i; if (full[i] > full[hi]) hi = i; }
    // other allocations of y and of the columns against the slowest and the fastest values array
    double *ys[6];
    uint16_t *cs[6];
    for (int j = 0; j < 6; ++j) {
        void *pad;
        CK(hipMalloc(&pad, (size_t)(j + 1) * 12'345'678));
        CK(hipMalloc(&ys[j], (size_t)nrows * 8));
        CK(hipMalloc(&cs[j], nnz * 2)); CK(hipMemset(cs[j], 1, nnz * 2));
    }
    for (int which : {lo, hi}) {
        printf("values array %d (%.1f us): the whole footprint with y stored plain / non-temporal / written through (sc1) / 8 KB bursts out of LDS:"
               " %6.1f %6.1f %6.1f %6.1f   values + y only: %6.1f %6.1f %6.1f %6.1f\n", which, full[which],
               parts_us<15, 1>(vals[which], col16, rowptr, x, y, nrows), parts_us<15, 2>(vals[which], col16, rowptr, x, y, nrows),
               parts_us<15, 3>(vals[which], col16, rowptr, x, y, nrows), parts_us<15, 4>(vals[which], col16, rowptr, x, y, nrows),
               parts_us<4, 1>(vals[which], col16, rowptr, x, y, nrows), parts_us<4, 2>(vals[which], col16, rowptr, x, y, nrows),
               parts_us<4, 3>(vals[which], col16, rowptr, x, y, nrows), parts_us<4, 4>(vals[which], col16, rowptr, x, y, nrows));
    }
    for (int which : {lo, hi}) {
        printf("values array %d (%.1f us): y in 6 other allocations:", which, full[which]);
        for (int j = 0; j < 6; ++j) printf(" %6.1f", parts_us<15>(vals[which], col16, rowptr, x, ys[j], nrows));
        printf("   columns in 6 other allocations:");
        for (int j = 0; j < 6; ++j) printf(" %6.1f", parts_us<15>(vals[which], cs[j], rowptr, x, y, nrows));
        printf("\n");
    }
    return 0;
}

template <int MASK>
static double merged_us(const unsigned char *vc, const uint32_t *rowptr, const double *x, double *y, uint32_t nrows) {
    const uint32_t nblocks = (nrows + 1023) / 1024, per_xcd = (nblocks + 7) / 8, C = 64;
    auto k = footprint_merged<MASK>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return time_us([&] { hipLaunchKernelGGL(k, dim3(per_xcd * 8 + 8 * C), dim3(256), 72 * 1024, 0, vc, rowptr, x, y, nrows, nblocks, C); }, 12);
}
// --map-merged [gb]: per index a values array (1.12 GB) and, right after it, a merged values + columns array (1.4 GB):
// values + columns as two streams (columns array fixed) against the one merged stream; and the whole footprint each way
static int map_merged_main(double gb) {
    const uint32_t nrows = 10'000'000;
    const size_t nnz = (size_t)nrows * 14 + 4096;
    const size_t mbytes = ((nnz + 127) / 128) * 1280;
    double *x, *y;
    uint16_t *col16;
    uint32_t *rowptr;
    CK(hipMalloc(&col16, nnz * 2)); CK(hipMemset(col16, 1, nnz * 2));
    CK(hipMalloc(&rowptr, ((size_t)nrows + 65) * 4)); CK(hipMemset(rowptr, 0, ((size_t)nrows + 65) * 4));
    CK(hipMalloc(&x, (size_t)nrows * 8)); CK(hipMalloc(&y, (size_t)nrows * 8));
    CK(hipMemset(x, 0, (size_t)nrows * 8));
    if (gb <= 0) gb = 60;
    const int K = (int)std::min<double>(gb * 1e9 / (nnz * 8.0 + mbytes), 120.0);
    std::vector<double *> vals(K, nullptr);
    std::vector<unsigned char *> vc(K, nullptr);
    for (int i = 0; i < K; ++i) {
        CK(hipMalloc(&vals[i], nnz * 8)); CK(hipMemsetAsync(vals[i], 1, nnz * 8, 0));
        CK(hipMalloc(&vc[i], mbytes)); CK(hipMemsetAsync(vc[i], 1, mbytes, 0));
    }
    CK(hipDeviceSynchronize());
    printf("index  two streams: values + columns | whole footprint   ||  merged stream: values + columns | whole footprint  (us)\n");
    for (int i = 0; i < K; ++i) {
        printf("  %3d  %6.1f  %6.1f  ||  %6.1f  %6.1f\n", i, parts_us<1>(vals[i], col16, rowptr, x, y, nrows),
               parts_us<15>(vals[i], col16, rowptr, x, y, nrows), merged_us<1>(vc[i], rowptr, x, y, nrows),
               merged_us<15>(vc[i], rowptr, x, y, nrows));
        fflush(stdout);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "--map-merged") return map_merged_main(argc > 2 ? atof(argv[2]) : 0.0);
    if (argc > 1 && std::string(argv[1]) == "--map-parts") return map_parts_main(argc > 2 ? atof(argv[2]) : 0.0, argc > 3 ? atoi(argv[3]) : 0);
    if (argc > 1 && std::string(argv[1]) == "--map-chunk") return map_chunk_main(argc > 2 ? atof(argv[2]) : 0.0);
    if (argc > 4 && std::string(argv[1]) == "--map-vmm") return map_vmm_main(atoll(argv[2]), atoll(argv[3]), atoll(argv[4]), argc > 5 ? atof(argv[5]) : 0.0);
    if (argc > 1 && std::string(argv[1]) == "--map") return map_main(argc > 2 ? atof(argv[2]) : 0.0);
    if (argc > 1 && std::string(argv[1]) == "--walk") return walk_main();
    if (argc > 3 && std::string(argv[1]) == "--quick") return quick_main((uint32_t)atoll(argv[2]), (uint32_t)atoll(argv[3]));
    if (argc > 1 && std::string(argv[1]) == "--place") return place_main();
    if (argc > 1 && std::string(argv[1]) == "--stride") return stride_main();
    const uint32_t nrows = 10'000'000;
    const size_t nnz = (size_t)nrows * 14 + 4096;
    double *vals, *x, *y;
    uint16_t *col16;
    uint32_t *rowptr;
    CK(hipMalloc(&vals, nnz * 8)); CK(hipMalloc(&col16, nnz * 2)); CK(hipMalloc(&rowptr, ((size_t)nrows + 65) * 4));
    CK(hipMalloc(&x, (size_t)nrows * 8)); CK(hipMalloc(&y, (size_t)nrows * 8));
    CK(hipMemset(vals, 1, nnz * 8)); CK(hipMemset(col16, 1, nnz * 2)); CK(hipMemset(rowptr, 0, ((size_t)nrows + 65) * 4));
    CK(hipMemset(x, 0, (size_t)nrows * 8));
    // plain copy / read of 1.6 GB
    {
        const size_t n16 = (size_t)100'000'000;   // 1.6 GB
        u32x4 *a, *b;
        uint32_t *flag;
        CK(hipMalloc(&a, n16 * 16)); CK(hipMalloc(&b, n16 * 16)); CK(hipMalloc(&flag, 4));
        CK(hipMemset(a, 3, n16 * 16));
        for (int grid : {2048, 4096, 8192}) {
            double us = time_us([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, a, b, n16); }, 20);
            printf("copy16  grid %5d  %7.1f us  %7.1f GB/s (read + write)\n", grid, us, 2.0 * n16 * 16 / us / 1e3);
            us = time_us([&] { hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, a, flag, n16); }, 20);
            printf("read16  grid %5d  %7.1f us  %7.1f GB/s (read only)\n", grid, us, 1.0 * n16 * 16 / us / 1e3);
        }
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(flag));
    }
    // the stream kernel's footprint
    for (int rep = 0; rep < 2; ++rep) {
        for (size_t lds : {(size_t)72 * 1024, (size_t)50 * 1024, (size_t)38 * 1024, (size_t)18 * 1024}) {
            run_footprint<1, false>("1 tile ahead, no x", lds, vals, col16, rowptr, x, y, nrows);
            run_footprint<2, false>("2 tiles ahead, no x", lds, vals, col16, rowptr, x, y, nrows);
            run_footprint<3, false>("3 tiles ahead, no x", lds, vals, col16, rowptr, x, y, nrows);
        }
        run_footprint<1, true>("1 tile ahead, x window staged", 72 * 1024, vals, col16, rowptr, x, y, nrows);
        run_footprint<2, true>("2 tiles ahead, x window staged", 72 * 1024, vals, col16, rowptr, x, y, nrows);
        run_footprint<3, true>("3 tiles ahead, x window staged", 72 * 1024, vals, col16, rowptr, x, y, nrows);
    }
    return 0;
}
