// Micro-benchmark (development tool): rate of ds_add_f64 / ds_add_f32 for address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

template <typename T>
__global__ __launch_bounds__(1024) void k(const uint32_t *__restrict__ idx, T *__restrict__ out, int iters, int n) {
    extern __shared__ unsigned char smem[];
    T *w = reinterpret_cast<T *>(smem);
    for (int i = threadIdx.x; i < 6144; i += blockDim.x) w[i] = T(0);
    __syncthreads();
    uint32_t a[8];
    for (int j = 0; j < 8; ++j) a[j] = idx[(blockIdx.x * 8 + j) * 1024 % n + threadIdx.x];
    for (int it = 0; it < iters; ++it)
        for (int j = 0; j < 8; ++j)
            __hip_atomic_fetch_add(&w[a[j]], T(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = w[0] + w[5];
}

template <typename T>
static void run(const char *name, const std::vector<uint32_t> &h) {
    uint32_t *d; T *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 4096 * sizeof(T));
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200, blocks = 1024;
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(1024), 6144 * sizeof(T), 0, d, o, 10, (int)h.size());
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(1024), 6144 * sizeof(T), 0, d, o, iters, (int)h.size());
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double atomics = (double)blocks * 1024 * 8 * iters;
    printf("%-44s %s  %8.1f G atomics/s  (%.2f per clk per CU at 2.4 GHz)\n", name, sizeof(T) == 8 ? "f64" : "f32",
           atomics / ms / 1e6, atomics / ms / 1e6 / 256 / 2.4);
    hipFree(d); hipFree(o);
}

int main() {
    const int n = 1 << 20;
    std::mt19937 rng(1);
    std::vector<uint32_t> rnd(n), lin(n), same(n), band(n), spread16(n);
    for (int i = 0; i < n; ++i) {
        rnd[i] = rng() % 6144;
        lin[i] = (i % 1024) + (i / 1024 % 5) * 1024;        // lane l -> consecutive elements
        same[i] = (i % 64) * 32 % 6144;                       // all lanes of a wave on one bank pair
        band[i] = (uint32_t)((i * 7 / 14) % 1024 + rng() % 4096) % 6144;  // like a banded CSC column
        spread16[i] = ((rng() % 384) * 16 + (i % 16)) % 6144;  // random rows, lane % 16 = element % 16
    }
    run<double>("random element", rnd);
    run<double>("consecutive elements (conflict-free)", lin);
    run<double>("one bank pair for the whole wave", same);
    run<double>("banded-CSC-like", band);
    run<double>("random, element % 16 == lane % 16", spread16);
    run<float>("random element", rnd);
    run<float>("consecutive elements (conflict-free)", lin);
    return 0;
}
