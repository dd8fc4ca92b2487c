// Micro-benchmark (development tool): what does a device-scope ticket (one atomicAdd on ONE address per workgroup, at
// its start) cost a launch of 39 063 short workgroups (the COO group kernel of config 5: ~29 us per workgroup, 7 per CU)?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/ticket.hip -o tools/micro/ticket
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>   // 0: blockIdx, 1: one ticket counter, 2: a counter per XCD-slot (blockIdx & 7)
__global__ __launch_bounds__(256, 7) void work(uint32_t *ticket, uint32_t *out, uint32_t busy_ticks, uint32_t *order) {
    __shared__ uint32_t s_id;
    uint32_t id = blockIdx.x;
    if (MODE) {
        if (threadIdx.x == 0) s_id = atomicAdd(&ticket[MODE == 2 ? (blockIdx.x & 7u) * 64 : 0], 1u);
        __syncthreads();
        id = s_id;
    }
    const uint64_t t0 = wall_clock64();
    uint32_t acc = id;
    while (wall_clock64() - t0 < busy_ticks) { acc = acc * 1664525u + 1013904223u; __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0) { out[id % 4096] = acc; if (order) order[blockIdx.x] = id; }
}

template <int MODE>
static void run(const char *name, uint32_t n, uint32_t busy, uint32_t *ticket, uint32_t *out, uint32_t *order) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        CK(hipMemset(ticket, 0, 4096));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(work<MODE>, dim3(n), dim3(256), 0, 0, ticket, out, busy, order);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-28s %6u workgroups x %5.1f us busy: %8.1f us\n", name, n, busy / 100.0, best * 1e3);
    if (order && MODE == 1) {   // how far does the ticket order stray from blockIdx order?
        uint32_t *h = (uint32_t *)malloc(n * 4);
        CK(hipMemcpy(h, order, n * 4, hipMemcpyDeviceToHost));
        long maxd = 0; double sum = 0;
        for (uint32_t i = 0; i < n; ++i) { long d = labs((long)h[i] - (long)i); if (d > maxd) maxd = d; sum += d; }
        printf("   ticket vs blockIdx: mean |diff| %.1f, max %ld\n", sum / n, maxd);
        free(h);
    }
}

int main() {
    uint32_t *ticket, *out, *order;
    CK(hipMalloc(&ticket, 4096)); CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&order, 1 << 20));
    for (uint32_t busy : {0u, 500u, 2900u}) {        // wall_clock64: 100 MHz -> 0 / 5 / 29 us
        run<0>("blockIdx", 39063, busy, ticket, out, order);
        run<1>("one ticket counter", 39063, busy, ticket, out, order);
        run<2>("a counter per blockIdx & 7", 39063, busy, ticket, out, order);
    }
    run<0>("blockIdx", 400000, 0, ticket, out, nullptr);
    run<1>("one ticket counter", 400000, 0, ticket, out, nullptr);
    return 0;
}
