// cblock_proto.hip -- development prototype (not part of the library): CSR y = A x for bands far wider than LDS,
// with the entries re-ordered into COLUMN BLOCKS.  Rows are taken in blocks of RB; a block's columns span a few
// panels of PC columns; the block's entries are stored panel-major, inside a panel in (row, column) order, with a
// 16-bit count per (row, panel).  A workgroup walks its block's panels in ascending order: stage the panel of x in
// LDS, then a lane per row continues that row's running sum over the row's entries in the panel -- the reference's
// left-to-right order (src/csr/ops/mul.rs:31-38), each product rounded once: bit-identical to the sequential loop.
//
//   hipcc -O3 --offload-arch=gfx950 cblock_proto.hip -L../../spal_synth -lspal_synth -o cblock_proto
//   ./cblock_proto [rows] [window] [per_row]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../spal_synth/spal_synth.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PC, int RB, int THREADS>
__global__ __launch_bounds__(THREADS) void cblock_spmv(const uint16_t *__restrict__ cnts, const uint32_t *__restrict__ wavebase,
                                                      const double *__restrict__ cval, const uint16_t *__restrict__ ccol,
                                                      const uint32_t *__restrict__ segbase, const uint32_t *__restrict__ pfirst,
                                                      const double *__restrict__ x, double *__restrict__ y, uint32_t nrows,
                                                      uint32_t ncols, uint32_t nblocks, uint32_t per_xcd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *xp = reinterpret_cast<double *>(smem);
    constexpr int RPTH = RB / THREADS;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t b = xcd * per_xcd + slot;
    if (slot >= per_xcd || b >= nblocks) return;
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t s0 = segbase[b], np = segbase[b + 1] - s0, p0 = pfirst[b];
    const uint32_t r0 = b * RB;
    double acc[RPTH];
    bool init[RPTH];
#pragma unroll
    for (int j = 0; j < RPTH; ++j) { acc[j] = 0.0; init[j] = false; }
    for (uint32_t p = 0; p < np; ++p) {
        const uint32_t seg = s0 + p;
        uint32_t cnt[RPTH], off[RPTH];
#pragma unroll
        for (int j = 0; j < RPTH; ++j) {
            cnt[j] = cnts[(size_t)seg * RB + j * THREADS + t];
            off[j] = wavebase[(size_t)seg * (RB / 64) + (j * THREADS + t) / 64];   // (uniform in the wave)
        }
        const uint64_t c0 = (uint64_t)(p0 + p) * PC;
        for (uint32_t i = t * 2; i < PC; i += THREADS * 2) {
            const uint64_t c = c0 + i;
            double2 v;
            v.x = c < ncols ? x[c] : 0.0;
            v.y = c + 1 < ncols ? x[c + 1] : 0.0;
            *reinterpret_cast<double2 *>(xp + i) = v;
        }
        __syncthreads();
        // entries of a 64-row slice are stored k-major (all first entries of the rows that have one, then all second
        // ones, ...): the lanes that still have an entry read consecutive addresses
        const uint64_t lt = (1ull << lane) - 1ull;
        // the first KU entries of every row slot: addresses from ballots alone, so all their loads go out together
        constexpr int KU = 6;
        double v[RPTH][KU];
        uint32_t c[RPTH][KU];
        uint32_t base[RPTH];
#pragma unroll
        for (int j = 0; j < RPTH; ++j) {
            base[j] = off[j];
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const bool on = (uint32_t)k < cnt[j];
                const uint64_t m = __ballot(on);
                const uint32_t at = on ? base[j] + (uint32_t)__popcll(m & lt) : 0u;
                v[j][k] = cval[at];
                c[j][k] = ccol[at];
                base[j] += (uint32_t)__popcll(m);
            }
        }
#pragma unroll
        for (int j = 0; j < RPTH; ++j) {
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                if ((uint32_t)k < cnt[j]) {
                    const double pr = v[j][k] * xp[c[j][k]];
                    acc[j] = init[j] ? acc[j] + pr : pr;
                    init[j] = true;
                }
            }
            for (uint32_t k = KU;; ++k) {      // rows with more entries in this panel
                const bool on = k < cnt[j];
                const uint64_t m = __ballot(on);
                if (!m) break;
                if (on) {
                    const uint32_t at = base[j] + (uint32_t)__popcll(m & lt);
                    const double pr = cval[at] * xp[ccol[at]];
                    acc[j] = init[j] ? acc[j] + pr : pr;
                    init[j] = true;
                }
                base[j] += (uint32_t)__popcll(m);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < RPTH; ++j) {
        const uint32_t r = r0 + j * THREADS + t;
        if (r < nrows) y[r] = init[j] ? acc[j] : 0.0;
    }
}

template <int PC, int RB, int THREADS>
static void run(uint64_t n, const std::vector<uint64_t> &rp, const std::vector<uint64_t> &ci, const std::vector<double> &va,
                const std::vector<double> &x, const std::vector<double> &yref) {
    const uint32_t nb = (uint32_t)((n + RB - 1) / RB);
    std::vector<uint32_t> segbase(nb + 1, 0), pfirst(nb, 0);
    for (uint32_t b = 0; b < nb; ++b) {
        uint64_t lo = ~0ull, hi = 0;
        for (uint64_t r = (uint64_t)b * RB; r < std::min<uint64_t>(n, (uint64_t)(b + 1) * RB); ++r)
            if (rp[r] < rp[r + 1]) { lo = std::min(lo, ci[rp[r]]); hi = std::max(hi, ci[rp[r + 1] - 1]); }
        if (lo == ~0ull) { pfirst[b] = 0; segbase[b + 1] = segbase[b]; continue; }
        pfirst[b] = (uint32_t)(lo / PC);
        segbase[b + 1] = segbase[b] + (uint32_t)(hi / PC - lo / PC + 1);
    }
    const uint32_t nseg = segbase[nb];
    std::vector<uint16_t> cnts((size_t)nseg * RB, 0);
    for (uint64_t r = 0; r < n; ++r) {
        const uint32_t b = (uint32_t)(r / RB);
        for (uint64_t e = rp[r]; e < rp[r + 1]; ++e) ++cnts[(size_t)(segbase[b] + (uint32_t)(ci[e] / PC) - pfirst[b]) * RB + r % RB];
    }
    std::vector<uint32_t> wavebase((size_t)nseg * (RB / 64));
    uint64_t run_ = 0;
    for (size_t i = 0; i < cnts.size(); ++i) {
        if (i % 64 == 0) wavebase[i / 64] = (uint32_t)run_;
        run_ += cnts[i];
    }
    const uint64_t nnz = rp[n];
    std::vector<double> cval(nnz + 64, 0.0);
    std::vector<uint16_t> ccol(nnz + 64, 0);
    {
        // per (segment, 64-row slice): rows' entries of this panel, then written k-major
        std::vector<std::vector<std::pair<double, uint16_t>>> rows(64);
        for (uint32_t b = 0; b < nb; ++b) {
            for (uint32_t p = 0; p < segbase[b + 1] - segbase[b]; ++p) {
                const uint32_t seg = segbase[b] + p, pn = pfirst[b] + p;
                for (uint32_t w = 0; w < RB / 64; ++w) {
                    uint32_t mx = 0;
                    for (uint32_t l = 0; l < 64; ++l) {
                        rows[l].clear();
                        const uint64_t r = (uint64_t)b * RB + w * 64 + l;
                        if (r >= n) continue;
                        // entries of row r inside panel pn (columns ascend: a binary search would do; rows are short)
                        for (uint64_t e = rp[r]; e < rp[r + 1]; ++e)
                            if (ci[e] / PC == pn) rows[l].push_back({va[e], (uint16_t)(ci[e] - (uint64_t)pn * PC)});
                        mx = std::max<uint32_t>(mx, (uint32_t)rows[l].size());
                    }
                    uint32_t at = wavebase[(size_t)seg * (RB / 64) + w];
                    for (uint32_t k = 0; k < mx; ++k)
                        for (uint32_t l = 0; l < 64; ++l)
                            if (k < rows[l].size()) { cval[at] = rows[l][k].first; ccol[at] = rows[l][k].second; ++at; }
                }
            }
        }
    }
    const double fmt_bytes = (double)nnz * 10 + (double)cnts.size() * 2 + (double)wavebase.size() * 4;
    printf("PC=%d RB=%d THREADS=%d: %u blocks, %u segments (%.2f panels per block), format %.1f MB (+ x panels %.1f MB from L2 per product)\n",
           PC, RB, THREADS, nb, nseg, (double)nseg / nb, fmt_bytes / 1e6, (double)nseg * PC * 8 / 1e6);
    uint16_t *d_cnts, *d_ccol;
    uint32_t *d_wb, *d_sb, *d_pf;
    double *d_cval, *d_x, *d_y;
    CK(hipMalloc(&d_cnts, cnts.size() * 2)); CK(hipMalloc(&d_ccol, ccol.size() * 2)); CK(hipMalloc(&d_wb, wavebase.size() * 4));
    CK(hipMalloc(&d_sb, segbase.size() * 4)); CK(hipMalloc(&d_pf, pfirst.size() * 4)); CK(hipMalloc(&d_cval, cval.size() * 8));
    CK(hipMalloc(&d_x, n * 8)); CK(hipMalloc(&d_y, n * 8));
    CK(hipMemcpy(d_cnts, cnts.data(), cnts.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ccol, ccol.data(), ccol.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wb, wavebase.data(), wavebase.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sb, segbase.data(), segbase.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pf, pfirst.data(), pfirst.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cval, cval.data(), cval.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x, x.data(), n * 8, hipMemcpyHostToDevice));
    auto kern = cblock_spmv<PC, RB, THREADS>;
    const size_t lds = (size_t)PC * 8;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const uint32_t per_xcd = (nb + 7) / 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i)
            hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(THREADS), lds, 0, d_cnts, d_wb, d_cval, d_ccol, d_sb, d_pf, d_x, d_y,
                               (uint32_t)n, (uint32_t)n, nb, per_xcd);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double B = (double)nnz * 12 + 4.0 * (n + 1) + 16.0 * n;
        printf("   %.1f us per product = %.1f %% of 8 TB/s (algorithmic bytes)\n", ms * 100, 100 * B / (ms / 10 * 1e-3) / 8e12);
    }
    std::vector<double> y(n);
    CK(hipMemcpy(y.data(), d_y, n * 8, hipMemcpyDeviceToHost));
    uint64_t bad = 0;
    for (uint64_t r = 0; r < n; ++r) bad += memcmp(&y[r], &yref[r], 8) != 0;
    printf("   rows that differ from the sequential loop bit for bit: %llu of %llu\n", (unsigned long long)bad, (unsigned long long)n);
    hipFree(d_cnts); hipFree(d_ccol); hipFree(d_wb); hipFree(d_sb); hipFree(d_pf); hipFree(d_cval); hipFree(d_x); hipFree(d_y);
}

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull;
    const uint64_t window = argc > 2 ? strtoull(argv[2], nullptr, 10) : 65536ull;
    const uint32_t per_row = argc > 3 ? (uint32_t)atoi(argv[3]) : 14u;
    std::vector<uint64_t> rp(n + 1), ci(n * per_row);
    std::vector<double> va(n * per_row), x(n), yref(n);
    if (spal_synth_banded_csr_rows_f64(n, n, per_row, window, 0x5EED0003ull, 0, n, rp.data(), ci.data(), va.data()) ||
        spal_synth_vector_f64(n, 0xC0FFEEull, x.data())) {
        fprintf(stderr, "synth: %s\n", spal_synth_last_error());
        return 1;
    }
    for (uint64_t r = 0; r < n; ++r) {
        double acc = 0.0;
        bool first = true;
        for (uint64_t e = rp[r]; e < rp[r + 1]; ++e) {
            const double p = va[e] * x[ci[e]];
            acc = first ? p : acc + p;
            first = false;
        }
        yref[r] = acc;
    }
    printf("banded %llu x %llu, %u per row, window %llu\n", (unsigned long long)n, (unsigned long long)n, per_row, (unsigned long long)window);
    run<8192, 4096, 1024>(n, rp, ci, va, x, yref);
    run<16384, 4096, 1024>(n, rp, ci, va, x, yref);
    run<8192, 2048, 512>(n, rp, ci, va, x, yref);
    run<8192, 8192, 1024>(n, rp, ci, va, x, yref);
    return 0;
}
