// Micro-benchmark (development tool): WHY do identical kernels on identical bytes run 250 / 265 / 285 us depending on
// which allocation the 1.12 GB values array lives in (profiles/r02/placement_of_the_values_array.txt)?
// Hypothesis: physical contiguity of the backing VRAM (PTE fragment size -> TLB reach), not the bytes' addresses.
//   * method 0 hipMalloc                      * method 1 hipExtMallocWithFlags(hipDeviceMallocContiguous)
//   * method 2 hipMemCreate (ONE physical handle) + hipMemMap     * method 3 hipMemCreate per 2 MB chunk, mapped in a row
//   * method 4 hipMallocAsync (default pool)
// per method: NB buffers of 1.12 GB kept alive, each timed with a pure read and with the config-3 footprint walk;
// then a TLB probe: dependent loads at strides 4 KB ... 32 MB (cycles per hop).  `--frag` first fragments VRAM
// (many small blocks, every other one freed) to see whether the classes can be provoked.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/placement.hip -o tools/micro/placement
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #e, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) double f64x2;

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
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

// every workgroup reads a contiguous run (what a walking kernel does: 2048 fronts far apart)
__global__ __launch_bounds__(256) void read_runs(const u32x4 *__restrict__ src, uint32_t *__restrict__ out, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = min(lo + per, n);
    u32x4 acc = {0, 0, 0, 0};
    size_t i = lo + threadIdx.x;
    for (; i + 768 < hi; i += 1024) {
        u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256),
              c = __builtin_nontemporal_load(src + i + 512), d = __builtin_nontemporal_load(src + i + 768);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < hi; i += 256) acc ^= src[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

// dependent loads: p = buf[p]; one lane.  cycles per hop.
__global__ void chase(const uint64_t *__restrict__ buf, uint64_t start, uint32_t hops, uint64_t *__restrict__ out) {
    uint64_t p = start;
    for (uint32_t i = 0; i < 64; ++i) p = __builtin_nontemporal_load(buf + p);   // warm the first few
    const uint64_t t0 = wall_clock64();
    for (uint32_t i = 0; i < hops; ++i) p = __builtin_nontemporal_load(buf + p);
    const uint64_t t1 = wall_clock64();
    out[0] = t1 - t0;
    out[1] = p;
}
__global__ void chase_init(uint64_t *buf, size_t nwords, size_t stride_words) {
    // buf[k * stride] = ((k + 1) * stride) % span, plus a 64-byte wobble so that lines differ
    const size_t nk = nwords / stride_words;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < nk; k += (size_t)gridDim.x * blockDim.x)
        buf[k * stride_words] = ((k + 1) % nk) * stride_words;
}

template <typename F>
static double time_us(F launch, int iters, int warm = 3) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < warm; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3 / iters;
}

struct Buf {
    void *p = nullptr;
    size_t bytes = 0;
    int method = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

static size_t g_gran_min = 0, g_gran_rec = 0;

static bool alloc_buf(Buf &b, int method, size_t bytes, size_t chunk = 0) {
    b.method = method;
    b.bytes = bytes;
    hipError_t e = hipSuccess;
    if (method == 0) e = hipMalloc(&b.p, bytes);
    else if (method == 1) e = hipExtMallocWithFlags(&b.p, bytes, hipDeviceMallocContiguous);
    else if (method == 4) { e = hipMallocAsync(&b.p, bytes, 0); if (e == hipSuccess) e = hipStreamSynchronize(0); }
    else if (method == 5) e = hipExtMallocWithFlags(&b.p, bytes, hipDeviceMallocUncached);
    else {
        hipMemAllocationProp prop;
        memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        const size_t gran = g_gran_rec ? g_gran_rec : (2u << 20);
        const size_t piece = method == 2 ? ((bytes + gran - 1) / gran) * gran : (chunk ? chunk : ((size_t)2 << 20));
        const size_t total = ((bytes + piece - 1) / piece) * piece;
        b.bytes = total;
        e = hipMemAddressReserve(&b.p, total, 0, nullptr, 0);
        if (e != hipSuccess) { printf("reserve: %s\n", hipGetErrorString(e)); return false; }
        for (size_t off = 0; off < total; off += piece) {
            hipMemGenericAllocationHandle_t h;
            e = hipMemCreate(&h, piece, &prop, 0);
            if (e != hipSuccess) { printf("hipMemCreate(%zu): %s\n", piece, hipGetErrorString(e)); return false; }
            b.handles.push_back(h);
            e = hipMemMap((char *)b.p + off, piece, 0, h, 0);
            if (e != hipSuccess) { printf("hipMemMap: %s\n", hipGetErrorString(e)); return false; }
        }
        hipMemAccessDesc acc;
        memset(&acc, 0, sizeof acc);
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = 0;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(b.p, total, &acc, 1);
    }
    if (e != hipSuccess) { printf("alloc method %d: %s\n", method, hipGetErrorString(e)); (void)hipGetLastError(); return false; }
    return true;
}
static void free_buf(Buf &b) {
    if (!b.p) return;
    if (b.method == 0 || b.method == 1 || b.method == 5) CK(hipFree(b.p));
    else if (b.method == 4) { CK(hipFreeAsync(b.p, 0)); CK(hipStreamSynchronize(0)); }
    else {
        CK(hipMemUnmap(b.p, b.bytes));
        for (auto h : b.handles) CK(hipMemRelease(h));
        CK(hipMemAddressFree(b.p, b.bytes));
    }
    b.p = nullptr;
    b.handles.clear();
}

static const char *mname(int m) {
    switch (m) {
    case 0: return "hipMalloc";
    case 1: return "hipExtMalloc(Contiguous)";
    case 2: return "hipMemCreate(one handle)";
    case 3: return "hipMemCreate(chunks)";
    case 4: return "hipMallocAsync";
    case 5: return "hipExtMalloc(Uncached)";
    }
    return "?";
}

static uint32_t *g_out;
static uint64_t *g_chase_out;

static void probe(Buf &b, const char *tag, bool tlb) {
    const size_t n16 = (size_t)1120000000 / 16;          // the values array of config 3
    const double us_r = time_us([&] { hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const u32x4 *)b.p, g_out, n16); }, 20);
    const double us_w = time_us([&] { hipLaunchKernelGGL(read_runs, dim3(2048), dim3(256), 0, 0, (const u32x4 *)b.p, g_out, n16); }, 20);
    printf("%-26s %-10s %p  read16 %7.1f us %6.0f GB/s   runs %7.1f us %6.0f GB/s", mname(b.method), tag, b.p, us_r,
           1.12e3 / us_r * 1e3, us_w, 1.12e3 / us_w * 1e3);
    if (tlb) {
        printf("  | hop ns @");
        const size_t strides[] = {4096 + 64, 65536 + 64, (2u << 20) + 64, (32u << 20) + 64};
        for (size_t s : strides) {
            const size_t sw = s / 8, nwords = (size_t)1120000000 / 8;
            hipLaunchKernelGGL(chase_init, dim3(1024), dim3(256), 0, 0, (uint64_t *)b.p, nwords, sw);
            CK(hipDeviceSynchronize());
            // flush caches / TLBs with a read of the whole buffer first
            hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const u32x4 *)b.p, g_out, n16);
            const uint32_t hops = (uint32_t)std::min<size_t>(nwords / sw, 4096);
            hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, (const uint64_t *)b.p, (uint64_t)0, hops, g_chase_out);
            CK(hipDeviceSynchronize());
            uint64_t r[2];
            CK(hipMemcpy(r, g_chase_out, 16, hipMemcpyDeviceToHost));
            printf(" %zuK: %.0f", s >> 10, (double)r[0] * 10.0 / hops);   // wall_clock64 = 100 MHz
        }
    }
    printf("\n");
    fflush(stdout);
}

int main(int argc, char **argv) {
    bool frag = false, tlb = true;
    int nb = 6;
    std::vector<int> methods = {0, 1, 2, 3, 4};
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--frag") frag = true;
        else if (a == "--no-tlb") tlb = false;
        else if (a == "--nb" && i + 1 < argc) nb = atoi(argv[++i]);
        else if (a == "--methods" && i + 1 < argc) { methods.clear(); for (char *c = argv[++i]; *c; ++c) methods.push_back(*c - '0'); }
    }
    CK(hipSetDevice(0));
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    (void)hipMemGetAllocationGranularity(&g_gran_min, &prop, hipMemAllocationGranularityMinimum);
    (void)hipMemGetAllocationGranularity(&g_gran_rec, &prop, hipMemAllocationGranularityRecommended);
    size_t fre = 0, tot = 0;
    CK(hipMemGetInfo(&fre, &tot));
    printf("granularity min %zu recommended %zu; free %.1f GB of %.1f GB\n", g_gran_min, g_gran_rec, fre / 1e9, tot / 1e9);
    CK(hipMalloc(&g_out, 64));
    CK(hipMalloc(&g_chase_out, 64));
    const size_t bytes = (size_t)1120000000 + 2048;

    std::vector<void *> shards;
    if (frag) {
        // fragment: 6000 blocks of 3 MB (odd size), free every other one
        for (int i = 0; i < 6000; ++i) { void *p; CK(hipMalloc(&p, 3u << 20)); shards.push_back(p); }
        for (size_t i = 0; i < shards.size(); i += 2) { CK(hipFree(shards[i])); shards[i] = nullptr; }
        printf("fragmented: 3000 x 3 MB holes\n");
    }
    for (int m : methods) {
        std::vector<Buf> bufs(nb);
        std::vector<void *> dummies;
        int got = 0;
        for (int i = 0; i < nb; ++i) {
            if (!alloc_buf(bufs[i], m, bytes)) break;
            ++got;
            void *d;                                    // odd-sized dummies between, as a library's small arrays would be
            CK(hipMalloc(&d, (size_t)(40 + 7 * i) << 20));
            dummies.push_back(d);
        }
        for (int rep = 0; rep < 2; ++rep)
            for (int i = 0; i < got; ++i) {
                char tag[32];
                snprintf(tag, sizeof tag, "buf%d r%d", i, rep);
                probe(bufs[i], tag, tlb && rep == 0);
            }
        for (int i = 0; i < got; ++i) free_buf(bufs[i]);
        for (void *d : dummies) CK(hipFree(d));
    }
    if (std::find(methods.begin(), methods.end(), 3) != methods.end()) {
        // chunk-size sweep of the VMM route: is the class a function of the physical piece size?
        const size_t chunks[] = {(size_t)2 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)1 << 30};
        for (size_t c : chunks) {
            Buf b;
            if (!alloc_buf(b, 3, bytes, c)) continue;
            char tag[32];
            snprintf(tag, sizeof tag, "%zuMB", c >> 20);
            probe(b, tag, tlb);
            free_buf(b);
        }
    }
    for (void *p : shards) if (p) CK(hipFree(p));
    printf("done\n");
    return 0;
}
