// Micro-benchmark (development tool): are the "placement classes" of profiles/r02 a PAIRWISE property of two streams?
// (round 3 found: the time of the config-3 footprint depends on where the values array lies RELATIVE to y and to the
// column array; a read of one array alone runs at the same rate everywhere.)
//   1. N buffers of 1 GiB (hipMalloc, allocation order = roughly physical order); buffer 0 is the reference.
//   2. for every buffer i: one kernel streams 512 MiB of the reference and 512 MiB of buffer i side by side
//      (read + read, and read + write); buffers of the reference's own class show up slower.
//   3. one representative per class: the full class x class matrix (is "same class = slow" transitive?).
//   4. inside one buffer: the same probe per 64 MiB piece (does a class boundary cut through an allocation?).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/rank_probe.hip -o tools/micro/rank_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <string>
#include <vector>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #e, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// every workgroup iteration: 4 KiB of a and 4 KiB of b (16 bytes per lane each), interleaved
template <int WRITE_B>
__global__ __launch_bounds__(256) void two_streams(const u32x4 *__restrict__ a, u32x4 *__restrict__ b, size_t n16, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + stride < n16; i += 2 * stride) {
        const u32x4 a0 = __builtin_nontemporal_load(a + i), a1 = __builtin_nontemporal_load(a + i + stride);
        if (WRITE_B) {
            b[i] = a0; b[i + stride] = a1;
        } else {
            const u32x4 b0 = __builtin_nontemporal_load(b + i), b1 = __builtin_nontemporal_load(b + i + stride);
            acc ^= b0 ^ b1;
        }
        acc ^= a0 ^ a1;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

static hipEvent_t e0, e1;
static uint32_t *g_out;

template <int WRITE_B>
static double pair_us(const void *a, void *b, size_t bytes, int iters = 6) {
    const size_t n16 = bytes / 16;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(two_streams<WRITE_B>, dim3(4096), dim3(256), 0, 0, (const u32x4 *)a, (u32x4 *)b, n16, g_out);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(two_streams<WRITE_B>, dim3(4096), dim3(256), 0, 0, (const u32x4 *)a, (u32x4 *)b, n16, g_out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / iters;
}

// --spacer: what does it cost to step through the device's memory with big allocations (hipMalloc of 4 ... 32 GiB, not
// touched), and does a 512 MiB buffer allocated after a spacer of S GiB land in another class than the one before it?
static int spacer_main() {
    const size_t GiB = (size_t)1 << 30, probe = (size_t)256 << 20;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMalloc(&g_out, 64));
    char *ref;
    CK(hipMalloc(&ref, GiB)); CK(hipMemset(ref, 1, GiB));
    for (size_t S : {(size_t)4, (size_t)8, (size_t)16, (size_t)32}) {
        std::vector<void *> hold;
        printf("spacers of %zu GiB: ", S);
        for (int k = 0; k < 6; ++k) {
            char *cand;
            CK(hipMalloc(&cand, (size_t)512 << 20)); CK(hipMemset(cand, 1, (size_t)512 << 20));
            const double rw = pair_us<1>(ref, cand, probe, 4), rr = pair_us<0>(ref, cand, probe, 4);
            hold.push_back(cand);
            void *sp = nullptr;
            hipEvent_t a, b;
            (void)a; (void)b;
            const auto t0 = std::chrono::steady_clock::now();
            CK(hipMalloc(&sp, S * GiB));
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            hold.push_back(sp);
            printf(" [rw %.0f rr %.0f | hipMalloc %.1f ms]", rw, rr, ms);
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (void *p : hold) CK(hipFree(p));
        printf("  (freeing all: %.1f ms)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        fflush(stdout);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "--spacer") return spacer_main();
    const size_t GiB = (size_t)1 << 30, probe = (size_t)512 << 20;
    size_t fre = 0, tot = 0;
    CK(hipMemGetInfo(&fre, &tot));
    int N = argc > 1 ? atoi(argv[1]) : (int)(fre * 0.9 / GiB);
    N = std::min(N, 280);
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMalloc(&g_out, 64));
    std::vector<char *> buf(N);
    for (int i = 0; i < N; ++i) { CK(hipMalloc(&buf[i], GiB)); CK(hipMemsetAsync(buf[i], 1, GiB, 0)); }
    CK(hipDeviceSynchronize());
    printf("%d buffers of 1 GiB; probe: 512 MiB of buffer 0 beside 512 MiB of buffer i (us): read+read, read+write\n", N);
    std::vector<double> rr(N), rw(N);
    for (int i = 1; i < N; ++i) {
        rr[i] = pair_us<0>(buf[0], buf[i], probe);
        rw[i] = pair_us<1>(buf[0], buf[i], probe);
        printf("  %3d  %7.1f  %7.1f\n", i, rr[i], rw[i]);
        fflush(stdout);
    }
    // classes by the read + write time against buffer 0: split at the midpoint of the range
    double lo = 1e30, hi = 0;
    for (int i = 1; i < N; ++i) { lo = std::min(lo, rw[i]); hi = std::max(hi, rw[i]); }
    printf("read+write against buffer 0: %.1f ... %.1f us\n", lo, hi);
    // representatives: greedy -- a buffer joins the first representative it is SLOW with, else founds a class
    std::vector<int> reps = {0};
    std::vector<int> cls(N, -1);
    cls[0] = 0;
    const double mid = 0.5 * (lo + hi);
    for (int i = 1; i < N && (hi - lo) > 0.03 * lo; ++i) {
        for (size_t k = 0; k < reps.size() && cls[i] < 0; ++k) {
            const double t = reps[k] == 0 ? rw[i] : pair_us<1>(buf[reps[k]], buf[i], probe, 4);
            if (t > mid) cls[i] = (int)k;
        }
        if (cls[i] < 0) { if (reps.size() < 6) { reps.push_back(i); cls[i] = (int)reps.size() - 1; } else cls[i] = 9; }
    }
    printf("classes in allocation order: ");
    for (int i = 0; i < N; ++i) printf("%d", cls[i] < 0 ? 0 : cls[i]);
    printf("\nrepresentatives:");
    for (int r : reps) printf(" %d", r);
    printf("\nclass x class, read+write / read+read (us):\n");
    for (int a : reps) {
        for (int b : reps) {
            // a second member of the class where there is one (a buffer beside ITSELF is another matter)
            int b2 = b;
            for (int i = 0; i < N; ++i) if (i != a && i != b && cls[i] == cls[b]) { b2 = i; break; }
            printf("  %6.1f/%6.1f", pair_us<1>(buf[a], buf[b2], probe, 4), pair_us<0>(buf[a], buf[b2], probe, 4));
        }
        printf("\n");
    }
    // inside buffers: 64 MiB pieces of buffer j against 64 MiB of buffer 0 (caches flushed by the size of the loop: 16 pieces)
    for (int j : {1, N / 2, N - 1}) {
        printf("buffer %d by 64 MiB pieces (read 64 MiB of buffer 0 + write the piece; 16 pieces one after the other, us each):", j);
        for (int p = 0; p < 16; ++p) printf(" %5.1f", pair_us<1>(buf[0] + ((size_t)(p * 37 % 16) << 26), buf[j] + ((size_t)p << 26), (size_t)64 << 20, 8));
        printf("\n");
    }
    return 0;
}
