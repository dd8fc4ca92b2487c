// csr_panel.hpp -- super-tiles whose x window is wider than LDS, taken in column PANELS (gfx950).
//
// A band of 16 384 or 65 536 columns under 1024 rows spans 68 / 260 pages of x: more than the 24 (48 KB) a
// workgroup stages beside its product strips.  csr_spmv_stream then gathers x from global memory / L2, 128-byte
// lines for 8-byte reads: 31 % of the HBM roofline at W = 16 384.  Here the workgroup keeps the ENTRIES of its 16
// tiles in registers (4 tiles per wave, loaded once, coalesced; 16-bit columns relative to the span's first column) and walks the window in panels of 40 pages (80 KB:
// the panel and the product strips, needed only after the last panel, share the workgroup's LDS): stage panel p in LDS, multiply every entry whose column lies in it (the product replaces the value in its
// register), next panel.  Every entry belongs to exactly one panel, so after the last one all products exist; they
// then go through the wave's LDS strip tile by tile and lane l sums row l left to right -- the reference's order
// of additions (src/csr/ops/mul.rs:31-38) with each product rounded once, exactly as in csr_spmv_stream: rows
// handled here stay BIT-IDENTICAL to the sequential CPU result.  x is read from L2 in whole pages (every page of
// the span once per super-tile, 16-byte coalesced) instead of one line per entry.
#pragma once
#include "csr_kernels.hpp"
#include "csr_slide.hpp"

namespace spal {

constexpr int kPanelTiles = 4;   // tiles a wave holds in registers (= the stream plan's tiles per wave)

template <typename T, int RPT>
__global__ __launch_bounds__(kStreamBlock, 2) void csr_spmv_panel(
    const uint32_t *__restrict__ rowptr, const uint16_t *__restrict__ col16, const T *__restrict__ vals,
    const T *__restrict__ x, T *__restrict__ y, const uint32_t *__restrict__ ptiles, const uint2 *__restrict__ pwin,
    const uint4 *__restrict__ desc, uint32_t npanel_tiles, uint32_t nrows, uint32_t ncols, uint32_t panel_pages,
    uint32_t flags) {
    using pair_t = typename Pair<T>::type;
    using vec_t = typename SlideVec<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char spal_smem[];
    constexpr uint32_t V = 16 / sizeof(T), VP = kPageCols / V;
    constexpr uint32_t kRows = stream_rows(kPanelTiles, RPT);
    const bool nt_store = flags & 1u;
    // the panel window and, after the last panel, the waves' product strips: the same LDS
    T *prod_all = reinterpret_cast<T *>(spal_smem);
    T *xw = prod_all;
    vec_t *xw4 = reinterpret_cast<vec_t *>(xw);
    if (blockIdx.x >= npanel_tiles) return;
    const uint32_t b = ptiles[blockIdx.x];
    const uint2 win = pwin[blockIdx.x];                  // {first page, pages} of the super-tile's column span
    const uint4 d = desc[b];
    const uint32_t skip_bits = desc_skip_bits(d), ulen = desc_ulen(d);
    const uint32_t row0 = b * kRows, row1 = min(row0 + kRows, nrows);
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    T *prod = prod_all + wave * stream_strip<false>();
    const uint32_t wrow = row0 + wave * (kPanelTiles * (uint32_t)RPT);
    const uint32_t ovmask = wave_skip_mask<kPanelTiles>(skip_bits, wave);

    // the wave's tiles, loaded once
    const uint32_t tb_lane = rowptr[min(wrow + min(lane, (uint32_t)kPanelTiles) * (uint32_t)RPT, row1)];
    uint32_t tb[kPanelTiles + 1];
#pragma unroll
    for (int k = 0; k <= kPanelTiles; ++k) tb[k] = __builtin_amdgcn_readlane(tb_lane, k);
    StreamTile<T> t[kPanelTiles];   // (16-bit columns, relative to the first column of the super-tile's span)
#pragma unroll
    for (int k = 0; k < kPanelTiles; ++k)
        stream_load<T, RPT>(t[k], rowptr, col16, vals, min(wrow + k * (uint32_t)RPT, row1), row1, tb[k], tb[k + 1], lane, ulen);

    // the panels of the window, ascending.  (Staging panel p + 1 asynchronously -- LDS-DMA into a second buffer --
    // was measured and lost: twice the panels at half the size, 553 vs 476 us at W = 16 384; the pass over the
    // entries per panel, not the staging, is what a panel costs.  profiles/r02/wide_bands_column_panels.txt)
    for (uint32_t p0 = 0; p0 < win.y; p0 += panel_pages) {
        const uint32_t np = min(panel_pages, win.y - p0), f = win.x + p0;
        if (p0) __syncthreads();                          // every wave is done with the previous panel
        for (uint32_t j0 = threadIdx.x; j0 < np * VP; j0 += 4u * kStreamBlock) {
            vec_t r[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t j = min(j0 + k * kStreamBlock, np * VP - 1u);
                r[k] = slide_load_vec_tail<T>(x, f + j / VP, j % VP, ncols);
            }
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t j = j0 + k * kStreamBlock;
                if (j < np * VP) xw4[j] = r[k];
            }
        }
        __syncthreads();
        const uint32_t base = p0 * kPageCols, width = np * kPageCols;   // (columns are relative to the span's first page)
#pragma unroll
        for (int k = 0; k < kPanelTiles; ++k) {
#pragma unroll
            for (int j = 0; j < kStreamSteps; ++j) {
                if ((uint32_t)j < t[k].steps) {           // uniform
                    const uint32_t c0 = (t[k].c[j] & 0xffffu) - base, c1 = (t[k].c[j] >> 16) - base;
                    const T x0 = xw[min(c0, width - 1u)], x1 = xw[min(c1, width - 1u)];
                    // an entry's column lies in exactly one panel: its product is formed once, there
                    if (c0 < width) t[k].v[j].x = t[k].v[j].x * x0;
                    if (c1 < width) t[k].v[j].y = t[k].v[j].y * x1;
                }
            }
        }
    }
    __syncthreads();   // the last panel is done with: the product strips take its place in LDS
    // products -> strip -> left-to-right row sums, a tile at a time
#pragma unroll
    for (int k = 0; k < kPanelTiles; ++k) {
        const uint32_t r0 = wrow + k * (uint32_t)RPT;
        if (r0 >= row1) break;                            // wave-uniform
        if ((ovmask >> k) & 1u) continue;                 // oversized tiles: csr_spmv_overflow
#pragma unroll
        for (int j = 0; j < kStreamSteps; ++j)
            if ((uint32_t)j < t[k].steps) reinterpret_cast<pair_t *>(prod)[j * 64 + lane] = t[k].v[j];
        __builtin_amdgcn_wave_barrier();
        strip_sums_to_y<T, RPT, false>(t[k], prod, y, r0, row1, lane, nt_store);
    }
}

}  // namespace spal
