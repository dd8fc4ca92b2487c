// wide_scatter.hip -- micro-benchmark for VERDICT r03 item 1(c): what does ONE radix pass cost when its digit is wider than
// 8 bits?  (Config 5 needs 16 sort bits: two 8-bit passes move 3.25 GB; a 12- or 13-bit first pass would let a larger group
// kernel finish the rest and save a whole pass -- if the memory system can take the short runs.)
//
// What is measured is the MEMORY side only: 50M entries of (key u32, aux u32, value f64) are read tile by tile (coalesced)
// and written to where a stable radix pass on BITS bits would put them.  Ranking costs nothing here by construction: the
// keys of a tile are a permutation of 0 .. TILE-1 (affine, different per tile), the digit of an entry is key >> (log2 TILE
// - BITS), so every tile holds exactly TILE >> BITS entries of every digit and the destination is arithmetic:
//     dest = digit * (N >> BITS) + tile * (TILE >> BITS) + (key & (TILE >> BITS) - 1)
// -- the address pattern of the real scatter (digit-major runs, one run per digit and tile, tiles dealt to the XCDs the way
// radix_scatter deals them) without histogram, scan or ballots.  Two forms:
//   direct : every lane stores its entry straight to its destination (what a wide digit amounts to anyway);
//   staged : the tile is first permuted in LDS into digit order and copied out linearly, so a digit's run leaves the
//            workgroup as consecutive lanes (radix_scatter's form; runs of TILE >> BITS entries).
//   staged2: the same in two rounds through HALF the LDS -- keys and aux words first, the values behind them in the same space --
//            so that a tile of 8192 entries (runs twice as long) still leaves room for two workgroups per CU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/wide_scatter.hip -o tools/micro/wide_scatter
// Run:   tools/micro/wide_scatter [entries]        (prints one line per (BITS, TILE, form): us per pass, GB/s of 32 B/entry)
//        rocprofv3 --pmc WRITE_SIZE FETCH_SIZE ... -- tools/micro/wide_scatter   (kernel names carry BITS / TILE / form)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int TILE>
__global__ __launch_bounds__(kThreads) void make_keys(uint32_t *key, uint32_t *aux, double *val, uint32_t ntiles) {
    const uint32_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    const uint32_t a = hash32(tile * 2u + 1u) | 1u, b = hash32(tile ^ 0x9e3779b9u);
    for (uint32_t p = threadIdx.x; p < (uint32_t)TILE; p += kThreads) {
        const uint64_t i = (uint64_t)tile * TILE + p;
        key[i] = (a * p + b) & (uint32_t)(TILE - 1);   // a odd: a permutation of 0 .. TILE-1
        aux[i] = (uint32_t)i;
        val[i] = (double)i;
    }
}

template <int BITS, int TILE, int STAGED>
__global__ __launch_bounds__(kThreads) void wide_scatter(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ ain,
                                                          const double *__restrict__ vin, uint32_t *__restrict__ kout,
                                                          uint32_t *__restrict__ aout, double *__restrict__ vout,
                                                          uint32_t ntiles, uint32_t per_xcd) {
    constexpr int ITEMS = TILE / kThreads;
    constexpr uint32_t LOG_TILE = TILE == 4096 ? 12 : TILE == 8192 ? 13 : 14;
    constexpr uint32_t PER = (uint32_t)TILE >> BITS;          // entries of one digit in one tile (>= 1)
    static_assert(BITS <= (int)LOG_TILE, "at least one entry per digit and tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);   // XCD-contiguous, as radix_scatter
    if (tile >= ntiles) return;
    const uint64_t t0 = (uint64_t)tile * TILE;
    const uint64_t bucket = (uint64_t)ntiles * PER;           // entries of one digit in all tiles
    uint32_t k[ITEMS], a[ITEMS];
    double v[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint64_t i = t0 + (uint64_t)j * kThreads + threadIdx.x;
        k[j] = kin[i]; a[j] = ain[i]; v[j] = vin[i];
    }
    if (STAGED == 2) {
        uint32_t *s_key = reinterpret_cast<uint32_t *>(smem), *s_aux = s_key + TILE;
        double *s_val = reinterpret_cast<double *>(smem);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { s_key[k[j]] = k[j]; s_aux[k[j]] = a[j]; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t lp = j * kThreads + threadIdx.x;
            const uint64_t gp = (uint64_t)(lp / PER) * bucket + (uint64_t)tile * PER + (lp % PER);
            kout[gp] = s_key[lp]; aout[gp] = s_aux[lp];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) s_val[k[j]] = v[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t lp = j * kThreads + threadIdx.x;
            const uint64_t gp = (uint64_t)(lp / PER) * bucket + (uint64_t)tile * PER + (lp % PER);
            vout[gp] = s_val[lp];
        }
    } else if (STAGED) {
        double *s_val = reinterpret_cast<double *>(smem);
        uint32_t *s_key = reinterpret_cast<uint32_t *>(s_val + TILE), *s_aux = s_key + TILE;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { s_key[k[j]] = k[j]; s_aux[k[j]] = a[j]; s_val[k[j]] = v[j]; }   // key = place in digit order
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t lp = j * kThreads + threadIdx.x;
            const uint64_t gp = (uint64_t)(lp / PER) * bucket + (uint64_t)tile * PER + (lp % PER);
            kout[gp] = s_key[lp]; aout[gp] = s_aux[lp]; vout[gp] = s_val[lp];
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint64_t gp = (uint64_t)(k[j] / PER) * bucket + (uint64_t)tile * PER + (k[j] % PER);
            kout[gp] = k[j]; aout[gp] = a[j]; vout[gp] = v[j];
        }
    }
}

struct Buffers {
    uint32_t *kin, *ain, *kout, *aout;
    double *vin, *vout;
};

template <int BITS, int TILE, int STAGED>
static void run(const Buffers &b, uint64_t n, int reps) {
    const uint32_t ntiles = (uint32_t)(n / TILE), per_xcd = (ntiles + 7) / 8;
    const size_t lds = STAGED == 2 ? (size_t)TILE * 8 : STAGED ? (size_t)TILE * 16 : 0;
    auto kern = wide_scatter<BITS, TILE, STAGED>;
    if (lds > 48 * 1024) CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((make_keys<TILE>), dim3(ntiles), dim3(kThreads), 0, 0, b.kin, b.ain, b.vin, ntiles);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(kThreads), lds, 0, b.kin, b.ain, b.vin, b.kout, b.aout, b.vout, ntiles, per_xcd);
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(kThreads), lds, 0, b.kin, b.ain, b.vin, b.kout, b.aout, b.vout, ntiles, per_xcd);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // check a sample: the output must be the stable sort by digit (every digit's run holds its tiles in order)
    const uint64_t used = (uint64_t)ntiles * TILE;
    std::vector<uint32_t> ko(4096), ao(4096);
    CHECK(hipMemcpy(ko.data(), b.kout + used / 2, 4096 * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ao.data(), b.aout + used / 2, 4096 * 4, hipMemcpyDeviceToHost));
    constexpr uint32_t PER = (uint32_t)TILE >> BITS;
    bool ok = true;
    for (int i = 0; i < 4096 && ok; ++i) {
        const uint64_t gp = used / 2 + i, bucket = (uint64_t)ntiles * PER;
        const uint32_t digit = (uint32_t)(gp / bucket), tile = (uint32_t)((gp % bucket) / PER);
        ok = ko[i] / PER == digit && ao[i] / TILE == tile;
    }
    const double us = ms * 1e3 / reps, gb = (double)used * 32.0 / 1e9;
    printf("bits %2d  tile %5d  %-7s  runs of %4u entries  %8.1f us per pass  %7.1f GB/s (16 B read + 16 B written per entry)  %s\n", BITS,
           TILE, STAGED == 2 ? "staged2" : STAGED ? "staged" : "direct", PER, us, gb / (us * 1e-6) , ok ? "ok" : "WRONG ORDER");
    fflush(stdout);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 50000000ull;
    n = n / 16384 * 16384;
    const int reps = 10;
    Buffers b;
    CHECK(hipMalloc((void **)&b.kin, n * 4)); CHECK(hipMalloc((void **)&b.ain, n * 4)); CHECK(hipMalloc((void **)&b.vin, n * 8));
    CHECK(hipMalloc((void **)&b.kout, n * 4)); CHECK(hipMalloc((void **)&b.aout, n * 4)); CHECK(hipMalloc((void **)&b.vout, n * 8));
    printf("# %llu entries, %d timed passes each; XCD-contiguous tiles\n", (unsigned long long)n, reps);
    if (argc > 2) {   // round 4, late: longer runs for the 8-bit digit
        run<8, 4096, 1>(b, n, reps);
        run<8, 4096, 2>(b, n, reps);
        run<8, 8192, 1>(b, n, reps);
        run<8, 8192, 2>(b, n, reps);
        return 0;
    }
    run<8, 4096, 1>(b, n, reps);
    run<8, 4096, 0>(b, n, reps);
    run<10, 4096, 1>(b, n, reps);
    run<10, 4096, 0>(b, n, reps);
    run<11, 4096, 1>(b, n, reps);
    run<11, 4096, 0>(b, n, reps);
    run<12, 4096, 1>(b, n, reps);
    run<12, 4096, 0>(b, n, reps);
    run<11, 8192, 1>(b, n, reps);
    run<12, 8192, 1>(b, n, reps);
    run<12, 8192, 0>(b, n, reps);
    run<13, 8192, 1>(b, n, reps);
    run<13, 8192, 0>(b, n, reps);
    return 0;
}
