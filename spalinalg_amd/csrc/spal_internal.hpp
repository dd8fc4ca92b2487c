// spal_internal.hpp -- shared declarations of libspal_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spal.h"

namespace spal {

// ---- thread-local error message -------------------------------------------
std::string &last_error_ref();
int fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define SPAL_HIP_TRY(expr)                                                          \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            int st_ = (e_ == hipErrorOutOfMemory) ? SPAL_ERR_OUT_OF_MEMORY          \
                      : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice)     \
                          ? SPAL_ERR_NO_DEVICE                                      \
                          : SPAL_ERR_HIP;                                           \
            return ::spal::fail(st_, "%s failed: %s (%s:%d)", #expr,                \
                                hipGetErrorString(e_), __FILE__, __LINE__);         \
        }                                                                           \
    } while (0)

#define SPAL_TRY(expr)                    \
    do {                                  \
        int st_ = (expr);                 \
        if (st_ != SPAL_OK) return st_;   \
    } while (0)

// Selects `device` for the calling thread, restores the previous one on exit.
struct DeviceGuard {
    int prev = -1;
    int status = SPAL_OK;
    explicit DeviceGuard(int device);
    ~DeviceGuard();
};

// ---- device memory ---------------------------------------------------------------
// hipMalloc / hipFree of hundreds of MB cost milliseconds (and on some hosts
// tens of ms) each; a handle that is assembled, used and destroyed in a loop
// would pay that every time.  dev_alloc / dev_free keep a small per-device
// cache of freed blocks (exact-fit-ish reuse, bounded by SPAL_CACHE_BYTES, default a quarter
// of the device's memory, at least 8 GiB).  dev_free synchronises the device first,
// like hipFree does, so a cached block is never handed out while work that used
// it is still in flight.  Small blocks are rounded up to a power of two.
// Stored entries a handle accepts: entry offsets are 32-bit on the device and the kernels compute
// `offset + a batch` (at most a few thousand entries past the end, clamped afterwards) in 32 bits.
constexpr uint64_t kMaxEntries = 0xffffffffull - 65536ull;

// ---- placement blocks: blocks of 1 GiB found by ONE walk per process and device, kept, and shared by every
// handle's vectors and 16-bit columns as pieces (first fit, 4 KB granules).  Two are kept when the walk met two classes of
// region: the place where the first handle ran fastest and the one where it ran slowest -- a later handle, whose values lie
// wherever they lie, times itself in each (no hipMalloc) and takes the better.
struct PlaceBlock { void *base; size_t size; };
int place_block_count(int device);
PlaceBlock place_block(int device, int index);
void *place_alloc(int device, int index, size_t bytes);        // a piece of block `index`, or nullptr
void place_free(int device, void *ptr);                        // a piece (nullptr is fine)
void place_adopt(int device, void *base, size_t size);         // a hipMalloc'ed block becomes a placement block
size_t place_free_bytes(int device);                           // what the retained blocks still have to give
void place_trim();                                             // blocks without pieces back to the driver
bool place_walked(int device);
void place_set_walked(int device);
hipError_t dev_alloc(void **ptr, size_t bytes);   // on the current device
hipError_t dev_free(void *ptr);                   // on the current device; NULL is fine
// a device block that returns to the allocator when it goes out of scope (early returns included)
struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)dev_free(p); }
    hipError_t alloc(size_t bytes) { return dev_alloc((void **)&p, bytes ? bytes : 1); }
    template <typename U> U *as() { return reinterpret_cast<U *>(p); }
    void *release() { void *q = p; p = nullptr; return q; }
};
void dev_cache_trim();                            // releases every cached block
hipError_t stream_acquire(hipStream_t *out);      // a non-blocking stream of the current device (pooled)
void stream_release(hipStream_t s);               // synchronises it and returns it to the pool

// ---- host helpers ------------------------------------------------------------
unsigned host_threads();
// fn(begin, end, tid) over [0, n) split into contiguous chunks, one per thread.
void parallel_for(uint64_t n, const std::function<void(uint64_t, uint64_t, unsigned)> &fn,
                  uint64_t min_chunk = 1u << 16);

// Compressed-format invariants (src/csr.rs:144-156); reason ordinal or 0.
int compressed_validate(uint64_t nrows, uint64_t ncols, bool major_is_rows,
                        const uint64_t *ptr, uint64_t ptr_len, const uint64_t *ind,
                        uint64_t ind_len, uint64_t val_len);
const char *invariant_text(int reason, bool csr);

// ---- device matrices ---------------------------------------------------------
struct CsrPlan {
    // kernel family: 1 = "vector" (L lanes per row, shuffle reduction, optional
    // LDS-staged x window); 2 = "stream" (lane per row, products through LDS,
    // 16-bit columns = position in the paged LDS x window; vector fallback per super-tile)
    int kernel = 0;          // 0 = not planned yet
    int user_kernel = 0;     // 0 = auto
    bool user_persistent = false;  // `persistent` was set by the caller or by the autotune
    int lanes_per_row = 0;   // L in {2,4,8,16,32,64}
    int unroll = 1;          // row groups in flight per wave iteration
    int vec_col16 = 0;       // vector kernel (long rows, LDS windows): reads 16-bit window-relative columns
    int vec_col16_allowed = 1;
    int long_rows = 0;       // vector kernel: batched rest-of-row loop (mean row length above 64)
    int threads = 512;       // workgroup size: 512 or 1024
    int tiles_per_wave = 4;  // stream kernel: tiles per wave (4 or 8)
    int rows_per_tile = 64;  // stream kernel: rows of a wave-tile (64, 32 or 16)
    int persistent = 0;      // stream kernel: fixed grid walking chunks of super-tiles
    int nt_store = 0;        // stream kernel: non-temporal stores of y
    int skew = 0;            // stream kernel: skewed product strips (rows a multiple of 128 bytes long: LDS bank conflicts)
    bool user_skew = false;
    int stream_row_max = 128; // stream kernel: tiles with a longer row go to the overflow kernel (a lane sums a row)
    int window_pages = 0;    // stream kernel: page budget of a super-tile's LDS x window (0 = automatic)
    int stream_global = 1;   // stream kernel: super-tiles whose pages exceed the LDS budget gather x from global
    int uniform_rows = 1;    // stream kernel: super-tiles whose rows all have one length do not read rowptr
    int prefetch = 1;        // stream kernel: tiles of loads a wave keeps in flight ahead of the one it works on (1 or 2)
    // the sliding-window kernel (csr_slide.hpp): band-like stream plans
    int slide_user = -1;     // -1 auto, 0 never
    int slide_on = 1;        // launch it (0: the one-super-tile-per-workgroup kernels on the same ring plan)
    int slide = 0;           // the plan is eligible and built for it
    int ring_pages = 0;      // the LDS x window is a ring of this many pages (col16 = (page % ring) * 256 + column in page)
    uint32_t slide_steps = 0;  // steps of 4 * rows_per_tile rows
    int panel_pages = 192;   // wide bands: super-tiles with a column span of at most this many pages go to csr_spmv_panel (0: never)
    int panel_on = 1;        // launch it (0: those super-tiles gather x from global memory in the stream kernels)
    int panel_window_pages = 0;  // pages of one panel (LDS)
    int panel_window_user = 0;   // option "panel_window" (0 = 80 KB)
    int slide_run = 0;       // steps per run (0 = one run per workgroup: fully persistent)
    int xcd_chunk = 32;      // one-super-tile stream kernel: chunks of super-tiles per XCD (0 = one run per XCD)
    // the column-blocked kernel (csr_cblock.hpp): matrices whose columns are not local
    int cblock_user = -1;    // option "cblock": -1 = when the plan finds most rows gathering x from beyond L2, 0 never, 1 always (tests)
    int cblock = 0;          // the tiled copy is built
    int cblock_pending = 0;  // the matrix qualifies; the copy is built by the first product (handles assembled on the device:
                             // the assembly call does not pay for a second copy nobody may use)
    int cblock_on = 1;       // launch it (autotune: 0 when the stream kernels measured faster)
    int cblock_rows = 0;     // rows of a row block (one workgroup each)
    int cblock_strip = 0;    // entries of the product strip (the fullest tile fits)
    int cblock_rows_user = 0, cblock_shift_user = 0;   // options "cblock_rows", "cblock_shift"
    int cblock_form_user = -1;   // option "cblock_form": -1 = by the entries per run, 0 = entry-parallel, 1 = rows form
    int cblock_form = 0;         // 0: csr_spmv_cblock (entry-parallel, rows16), 1: csr_spmv_cblock_rows (threads own rows, cnt8)
    float cblock_run = 0.f;      // entries per (row, column block) that holds any
    int cblock_shift = 0;    // a column block holds 2^shift columns
    int cblock_nbc = 0;      // column blocks
    uint32_t cblock_nrb = 0; // row blocks = workgroups
    double nonlocal_row_fraction = 0.0;   // rows of super-tiles that gather x from global memory over a span no panel holds
    int slide_S = 0;         // 128-entry steps of the largest streamed tile: every tile issues that many loads
    int slide_uniform = 0;   // 1 + the length of every row when all streamed rows have one length (rowptr is not read), else 0
    int all_rows_uniform = 0; // ... and EVERY row of the matrix has it (rowptr[r] = r * length): tile bounds are arithmetic
    int arith_bounds = 1;    // option "arith_bounds": use that (0: load the tiles' bounds from rowptr as before)
    int slide_fill_ok = 1;   // the sliding kernel's tiles are full enough for its fixed count of loads per tile (csr_slide plan)
    int slide_fill_user = -1; // -1 = by the tiles' average fill, 1 = "slide_on" was asked for by name
    int slide_even = 1;      // option "slide_even": one run per workgroup -> the XCD's steps split evenly over all its workgroups
    int blockwin = -1;       // option "blockwin": -1 = timed against the row split where that is built, 0 never, 1 whenever the windows fit (tests)
    int row_split = -1;      // option "row_split": -1 = when long rows keep a tenth of the 64-row tiles from streaming, 0 never, 1 always (tests)
    int split_threshold = 128;   // option "row_split_threshold": rows above it are "long"
    int place_tries = 8;     // autotune: blocks of 1 GiB the 16-bit columns are tried in, at most (see csr_autotune)
    int split_tiles_on = 1;  // sliding kernel: tiles above 1024 entries whose halves fit go through the strip twice
    int diag = 0;            // ablation builds (-DSPAL_DIAG) only: parts of the stream kernel switched off
    int persistent_blocks = 0;    // its grid; 0 = what the device holds at once (LDS per workgroup decides: f64 band 512)
    int rows_per_block = 0;  // R
    int lds_x = 0;           // stage the block's x window in LDS
    uint32_t lds_entries = 0;  // LDS window capacity (elements) when lds_x
    uint32_t nblocks = 0;
    double lds_row_fraction = 0.0;  // rows whose block window fits
    double stream_row_fraction = 0.0;  // rows handled by the stream path (kernel 2)
    double uniform_row_fraction = 0.0; // rows in super-tiles that do not read rowptr (all rows one length)
    bool user_rows_per_block = false, user_lanes = false, user_lds = false,
         user_unroll = false, user_threads = false, user_rows_per_tile = false;
};

// implemented in spal_coo.hip: stable sort of the entries by their minor index
// (compressed-by-major -> compressed-by-minor); outputs are hipMalloc'ed with
// *out_cap entries (nnz + over-read margin) and owned by the caller
int transpose_device(int device, int elem_size, uint64_t nmajor, uint64_t nminor, uint64_t nnz,
                     const uint32_t *d_ptr, const uint32_t *d_ind, const void *d_val,
                     hipStream_t st, uint32_t **out_ptr, uint32_t **out_ind, void **out_val,
                     uint64_t *out_cap);
// implemented in spal_csc.hip: handle around device arrays it takes ownership of
int csc_adopt_device(int device, int elem_size, uint64_t nrows, uint64_t ncols, uint64_t nnz,
                     uint32_t *d_colptr, uint32_t *d_rowind, void *d_values, spal_csc **out);
}  // namespace spal

// The opaque handle types of spal.h.
struct spal_csr {
    int device = 0;
    int elem_size = 8;  // 8 = f64, 4 = f32
    uint64_t nrows = 0, ncols = 0, nnz = 0;
    uint32_t *d_rowptr = nullptr;  // nrows + 1
    uint32_t *d_colind = nullptr;  // nnz
    void *d_values = nullptr;      // nnz * elem_size
    uint16_t *d_col16 = nullptr;   // nnz (+pad): page slot * 256 + column inside the page, streamable super-tiles
    uint32_t *d_ovtiles = nullptr; // [count][first rows of the n_ovtiles tiles the stream kernels skip (csr_spmv_overflow)]
    uint32_t n_ovtiles = 0;
    uint32_t *d_pages = nullptr;   // blocks * page budget: ascending page ids of super-tiles whose pages are not one run
    uint32_t *d_ptiles = nullptr;  // panel kernel: the super-tiles it takes
    uint2 *d_pwin = nullptr;       // ... and {first page, pages} of their column spans
    uint32_t n_ptiles = 0;
    uint2 *d_sdesc = nullptr;      // sliding kernel: per step {first page, pages | skip << 8 | flags << 16 | split << 20}
    uint32_t *d_ovtiles_slide = nullptr;   // first rows of the tiles left to csr_spmv_overflow when the sliding kernel runs
    uint32_t n_ovtiles_slide = 0, n_split_tiles = 0;   // ... and the number it takes itself, in two halves
    uint4 *d_desc = nullptr;       // per row block: Stream {first page / offset into d_pages, pages, mode, contiguous};
                                   // VectorLds {window base column, window length, mode, 0}
    uint64_t cap_entries = 0;      // allocated entries of d_colind / d_values (>= nnz + pad)
    // column-blocked copy (csr_cblock.hpp): entries ordered (row block, column block, row, column)
    void *d_cb_val = nullptr;
    uint32_t *d_cb_col = nullptr, *d_cb_tile = nullptr;   // columns; first entry of every tile (+ the end)
    uint8_t *d_cb_cnt = nullptr;   // entries per (tile, row): the builder's scratch
    uint16_t *d_cb_row = nullptr;  // row of every entry inside its row block
    float cblock_us[2] = {0.f, 0.f};   // autotune: per launch {stream kernels, column-blocked kernel}
    // ROW SPLIT (skewed row lengths: a few long rows poison the 64-row tiles of the stream kernel): the matrix is multiplied as
    // A = A_short + A_long -- `split_short` is a complete handle of its own over a compacted copy of the rows of at most
    // `split_threshold` entries (the long rows are empty rows in it: it writes every row of y), `d_split_rows` lists the
    // long rows, which csr_spmv_row_list then overwrites out of THIS handle's arrays.  No temporaries: concurrent products stay safe.
    spal_csr *split_short = nullptr;
    uint32_t *d_split_rows = nullptr;
    uint32_t split_nlong = 0, split_nheavy = 0;   // listed rows (longest first); the first split_nheavy hold more than 1024 entries
    uint64_t split_long_entries = 0;
    int split_child = 0;           // this handle IS the short part of a split (never splits again)
    // BLOCK WINDOW kernel (spal_csr_blockwin.hip): blocks of bw_rows rows, a window of at most bw_cols columns each; bw_on: the products run it
    uint32_t *d_bworder = nullptr; // 32-byte records {block, first entry, one past the last, window's first column, columns, -, -, -} in the order the blocks are dealt
    uint32_t bw_blocks = 0, bw_rows = 0, bw_cols = 0, bw_grid = 0;   // (bw_grid: resident workgroups, 8 x the walks per XCD)
    uint32_t bw_list_rows = 0;     // rows of 8 records in d_bworder (a walk's blocks are J rows apart)
    int bw_on = 0;
    float bw_us[2] = {0.f, 0.f};   // setup: per product {what it was timed against, the block-window kernel}
    int plan_pending = 0;          // a device-assembled handle: the product kernels' plan is built by whoever needs it first (csr_ensure_plan)
    int cblock_lazy = 0;           // build the tiled copy with the first product, not with the plan (csr_adopt_device)
    int cblock_failed = 0;         // building it failed (out of memory, ...): the stream kernels run instead
    // spal_csr_alloc_vectors: the block of 1 GiB (or more) that holds the caller's x and y, found by the placement walk
    void *d_vec_block = nullptr;       // x and y: a piece of one of the process's placement blocks (place_*), or a block of its own
    int vec_block_owned = 0;           // 1: hipMalloc'ed for this handle alone (small matrices: nothing to place)
    size_t vec_x_off = 0, vec_y_off = 0;
    float walk_us[2] = {0.f, 0.f};     // fastest / slowest candidate of the walk (per product)
    int walk_blocks = 0;               // NEW blocks of 1 GiB this handle's call took from the device (0 once the process has its blocks)
    int walk_probes = 0;               // places timed (new blocks + the process's retained ones)
    int walk_max = 8;                  // option "walk_blocks": blocks of 1 GiB a walk may hold at once = how far it reaches
    int col16_placed = 0;              // d_col16 is a piece of a placement block (place_free, not dev_free)
    std::mutex mu_cb;
    // autotune: microseconds per launch of {plain, persistent} x {plain, non-temporal y stores}
    float tuned_us[4] = {0.f, 0.f, 0.f, 0.f};
    float place_us[2] = {0.f, 0.f};   // autotune: per launch before / after re-placing the values array
    int place_tried = 0;
    spal::CsrPlan plan;
    std::vector<uint2> win_base;   // host copy of the per-256-row column windows (planner cache)
    // a device-assembled handle: {first column, one past the last} of every group of 2^win_group_bits rows as the assembly's
    // group kernel saw them, still on the device -- fetched and folded into win_base when the plan is built (csr_plan_build)
    uint2 *d_win_groups = nullptr;
    uint32_t win_groups = 0, win_group_bits = 0;
    // host-convenience staging (spal_csr_spmv_*): guarded by mu
    std::mutex mu;
    void *d_x = nullptr, *d_y = nullptr;
    hipStream_t stream = nullptr;
    // More stored entries than 32-bit device offsets address (the reference's are usize, src/csr.rs:66-72): the
    // handle is then a list of ROW BLOCKS, each a complete handle of its own (own plan, offsets relative to the
    // block's first entry) over the same columns; this parent owns no matrix arrays.  A product launches the blocks
    // one after the other on the caller's stream, block b writing y[part_row0[b] ...).
    std::vector<spal_csr *> parts;
    std::vector<uint64_t> part_row0, part_entry0;   // first row / first entry of every block, then nrows / nnz
};

struct spal_csc {
    int device = 0;
    int elem_size = 8;
    uint64_t nrows = 0, ncols = 0, nnz = 0;
    uint32_t *d_colptr = nullptr;  // ncols + 1
    uint32_t *d_rowind = nullptr;  // nnz (+pad)
    void *d_values = nullptr;      // nnz (+pad)
    uint32_t *d_meta = nullptr;    // nnz (+pad): (row - window base) | (col - tile base) << 16
    uint4 *d_desc = nullptr;       // per super-tile of columns {window base row, length, mode, offset of its window in d_windows}
    // two-phase flush (flush == 1): LDS-mode super-tiles store their y windows here; csc_window_reduce
    // adds, for every chunk of 1024 rows, the windows that overlap it (ascending super-tile = column order)
    void *d_windows = nullptr;     // sum of the LDS-mode window lengths, elements
    uint32_t *d_chunk_ptr = nullptr, *d_chunk_blk = nullptr;  // CSR-like cover lists: chunk -> super-tiles
    uint32_t nchunks = 0;
    uint64_t windows_entries = 0;
    int flush = 0;                 // 0 = neighbour hand-off when the plan allows it (else global atomics), 1 = windows + ordered reduce, 2 = global atomics
    // neighbour hand-off (`ordered`): row windows ascend and only adjacent super-tiles overlap, so every row of y is
    // STORED by the first super-tile that covers it and updated by the next one after a flag: no memset, no atomics
    int ordered = 0;
    uint32_t *d_prev_hi = nullptr; // per super-tile: end of the previous super-tile's window (where its own rows begin)
    uint32_t *d_flags = nullptr;   // per super-tile: the launch number whose owned rows are in y; [nblocks + 1] = the ticket counter
    uint32_t *h_gave_up = nullptr, *d_gave_up = nullptr;   // one word of mapped host memory: a super-tile hit its spin bound
    uint32_t ticket_next = 0;      // value of the ticket counter when the next launch begins (guarded by mu_launch)
    uint32_t spin_bound = 1u << 22;
    int use_ticket = -1;           // option "ticket": -1 = ticket_auto
    uint32_t uniform_cols = 0;     // 1 + the length of every column when all columns have one length (the kernel computes colptr), else 0
    int ticket_auto = 0;           // plan: the launch has more workgroups than the device holds at once
    int handoff_timeouts = 0;      // products of this handle that hit the spin bound (their y was invalid)
    uint32_t epoch = 0;            // launch number (guarded by mu, with ev_last)
    hipEvent_t ev_last = nullptr;  // launches of one handle run one after the other (they share d_flags): stream order, or this event across streams
    hipStream_t last_stream = nullptr;
    int last_stream_valid = 0;
    std::mutex mu_launch;          // ... chained under this lock
    // row tiles (spal_csc_rowtiles.hip): the entries a second time, ordered (row tile, column, row) -- a workgroup owns rows
    // of y outright, no hand-off; built where every tile's window of x fits LDS, and then the scatter path's default
    int rowtiles = 0;              // the copy is built
    int rowtiles_failed = 0;       // building it failed (out of memory, ...): the column tiles run instead
    int rowtiles_user = -1;        // option "row_tiles": -1 / 1 = where it qualifies, 0 = never (the column tiles run)
    uint32_t rt_rows = 0, rt_ntiles = 0, rt_xcap = 0;   // rows of a tile, tiles, widest x window (elements)
    uint32_t rt_rows_user = 0;     // option "row_tile_rows": 0 = the tallest of 4096 / 2048 / 1024 that fits
    void *d_rt_val = nullptr;      // nnz (+pad)
    uint32_t *d_rt_meta = nullptr; // nnz (+pad): (row - tile's first row) | (col - window's first column) << 16
    uint32_t *d_rt_ptr = nullptr;  // ntiles + 1
    uint4 *d_rt_desc = nullptr;    // {first row, rows, first column of the x window, columns}
    int all_lds = 0;               // every super-tile with entries is in LDS mode: y needs no memset
    int cols_per_block = 1024;     // columns of a super-tile: 4096 / 2048 / 1024 (the widest whose row windows fit LDS)
    int user_cols = 0;             // option "cols_per_block" (0 = automatic)
    uint32_t nblocks = 0;
    uint32_t lds_entries = 0;      // largest LDS y window (elements); 0 = global scatter only
    double lds_col_fraction = 0.0;
    int use_lds = 1;
    int kernel = 2;                // 1 = atomic scatter (LDS-privatised / global), 2 = transposed (CSR kernels; default)
    spal_csr *as_csr = nullptr;    // kernel 2: the same matrix as CSR, built on the device on first use
    int lanes_per_col = 0;
    std::mutex mu;
    void *d_x = nullptr, *d_y = nullptr;
    hipStream_t stream = nullptr;
};

struct spal_coo {
    int device = 0;
    int elem_size = 8;
    uint64_t nrows = 0, ncols = 0, len = 0;
    uint32_t *d_rows = nullptr, *d_cols = nullptr;
    void *d_vals = nullptr;
    void *d_work = nullptr;   // sort buffers + scratch of the assembly, allocated at upload
    // a HINT only: the fullest group of rows [0] / columns [1] the last assembly met (it picks the group kernel's LDS
    // capacity without a host round trip; the kernel verifies it, see coo_assemble_t).  Nothing else about the triplets
    // is kept between assemblies.
    uint32_t cap_hint[2] = {0, 0};
    int loop_hint[2] = {0, 0};   // the last assembly by rows [0] / columns [1] met a row beyond the network form's reach
    int last_row_sort = 0;
    size_t work_bytes = 0;
    void *h_back = nullptr;   // pinned host memory the assembly's few results come back into
    size_t h_back_bytes = 0;
    std::mutex mu;            // serialises assemblies on one handle (shared workspace)
    int last_group_rows = 0, last_group_cap = 0;  // geometry of the last assembly's local sort (0 = general route)
    int last_relaunches = 0;        // group kernel launched again because the capacity hint was too small
    int last_lookback_gave_up = 0;  // assemblies of this handle whose look-back hit its spin bound (backstop taken)
    int last_ticket = 0, last_packed = 0, last_offsets = 0;   // how ids were handed out (0 blockIdx, 1 one counter, 8 class counters); packed payload; offsets from the passes' counts
};

namespace spal {
// implemented in spal_csr.hip
int csr_plan_build(spal_csr *a);
int csr_launch(spal_csr *a, const void *x_dev, void *y_dev, hipStream_t stream);
int csr_ensure_plan(spal_csr *a, hipStream_t launch_stream, bool from_launch);
// implemented in spal_csr_slide.hip: the sliding-window kernel for a plan with plan.slide set
hipError_t launch_slide(const spal_csr *a, const void *x, void *y, hipStream_t st);
// ... and the column-panel kernel over a->d_ptiles
hipError_t launch_panel(const spal_csr *a, const void *x, void *y, hipStream_t st);
// implemented in spal_csc_rowtiles.hip: CSC scatter over row tiles
int csc_rowtiles_plan(spal_csc *a);
void csc_rowtiles_free(spal_csc *a);
hipError_t launch_csc_rowtiles(const spal_csc *a, const void *x, void *y, hipStream_t st);
// implemented in spal_csr_cblock.hip: the column-blocked kernel (tiled copy of the matrix, x slices kept in L2)
int cblock_plan(spal_csr *a, bool force);
void cblock_free(spal_csr *a);
hipError_t launch_cblock(const spal_csr *a, const void *x, void *y, hipStream_t st);
// spal_csr_blockwin.hip: skewed row lengths with columns near the rows -- entries streamed, one LDS window of x per row block
int blockwin_plan(spal_csr *a);        // measures the windows; a->bw_rows != 0 when the matrix fits the kernel
void blockwin_free(spal_csr *a);
hipError_t blockwin_launch(const spal_csr *a, const void *x, void *y, hipStream_t st);
// builds a handle around device arrays it takes ownership of (used by the COO
// assembly, which produces CSR directly on the device)
// (cap_entries = allocated entries of d_colind / d_values; re-allocated with
// padding when smaller than nnz + the kernels' over-read margin)
// (win256: optional {first column, one past the last} of every 256 rows, if the caller has it; or d_win_groups: the same
//  per group of 2^win_group_bits <= 256 rows, on the device, ownership passes to the handle)
int csr_adopt_device(int device, int elem_size, uint64_t nrows, uint64_t ncols,
                     uint64_t nnz, uint64_t cap_entries, uint32_t *d_rowptr,
                     uint32_t *d_colind, void *d_values, spal_csr **out,
                     const std::vector<uint2> *win256 = nullptr, bool eager_copies = false, bool lazy_plan = false,
                     uint2 *d_win_groups = nullptr, uint32_t win_groups = 0, uint32_t win_group_bits = 0);
}  // namespace spal
