/*
 * spal.h -- C ABI of libspal_hip.so: the MI355X (gfx950) SpMV / assembly path
 * that sits underneath spalinalg's public CsrMatrix / CscMatrix / CooMatrix
 * types (reference: lokyhark/spalinalg, paths below relative to its root).
 *
 * The reference has no FFI of its own; every entry point here
 * names the reference interface it would be bound under.  The Rust-side
 * binding a maintainer adds is shown in INTEGRATION.md and rust_shim/.
 *
 * Conventions
 *  - plain C: pointers + sizes only, no C++ types, no exceptions, no torch.
 *  - `usize` of the reference == uint64_t here (x86-64 Linux).
 *  - host arrays are BORROWED for the duration of the call; device copies are
 *    owned by the opaque handle and released by *_destroy.
 *  - every function returns a spal_status; on failure a thread-local message
 *    is available from spal_last_error().  The reference's convention is to
 *    panic on a contract violation (assert!, src/csr.rs:144-156); a binding
 *    turns any non-zero status into panic!() to keep that behaviour.
 *  - scalars: exactly the two `Scalar` impls, f32 and f64 (src/scalar.rs:56-57).
 *  - there is NO CPU fallback: without a usable HIP device every compute
 *    entry point fails with SPAL_ERR_NO_DEVICE / SPAL_ERR_HIP.
 */
#ifndef SPAL_H
#define SPAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum spal_status {
    SPAL_OK = 0,
    SPAL_ERR_INVALID_ARGUMENT = 1, /* null pointer, x.len() != ncols (src/csr/ops/mul.rs:9) ... */
    SPAL_ERR_INVARIANT = 2,        /* the reference constructor would panic (src/csr.rs:144-156) */
    SPAL_ERR_HIP = 3,              /* a HIP runtime call failed */
    SPAL_ERR_OUT_OF_MEMORY = 4,
    SPAL_ERR_UNSUPPORTED = 5,      /* shape does not fit the device's 32-bit index format */
    SPAL_ERR_NO_DEVICE = 6,
    SPAL_ERR_INDEX_OUT_OF_BOUNDS = 7 /* COO entry outside the matrix (src/coo.rs:432-433) */
} spal_status;

/* Opaque device-resident matrices. */
typedef struct spal_csr *spal_csr_t; /* mirrors CsrMatrix<T>, src/csr.rs:66-72 */
typedef struct spal_csc *spal_csc_t; /* mirrors CscMatrix<T>, src/csc.rs:66-72 */
typedef struct spal_coo *spal_coo_t; /* mirrors CooMatrix<T>, src/coo.rs:53-57 (SoA on device) */
typedef struct spal_mg *spal_mg_t;          /* the GPUs of one node + their RCCL communicators */
typedef struct spal_mg_csr *spal_mg_csr_t;  /* a CsrMatrix partitioned by rows over them */

/* ---- library ---------------------------------------------------------- */
const char *spal_last_error(void);   /* thread-local, never NULL */
const char *spal_version(void);
int spal_device_count(int *count);   /* 0 devices is SPAL_OK with *count = 0 */

/* ---- host-side checks (no device needed) ------------------------------- */
/* The assertions of CsrMatrix::new (src/csr.rs:144-156), in order.  On
 * SPAL_ERR_INVARIANT *reason (may be NULL) receives the 1-based ordinal of
 * the first assertion that fails:  1 nrows>0, 2 ncols>0, 3 rowptr.len()==
 * nrows+1, 4 rowptr[0]==0, 5 colind.len()==rowptr[nrows], 6 values.len()==
 * rowptr[nrows], 7 rowptr sorted, 8 colind in range, 9 colind strictly
 * increasing inside each row. */
int spal_csr_validate(uint64_t nrows, uint64_t ncols,
                      const uint64_t *rowptr, uint64_t rowptr_len,
                      const uint64_t *colind, uint64_t colind_len,
                      uint64_t values_len, int *reason);
/* CscMatrix::new (src/csc.rs:144-156): same ordinals with colptr / rowind. */
int spal_csc_validate(uint64_t nrows, uint64_t ncols,
                      const uint64_t *colptr, uint64_t colptr_len,
                      const uint64_t *rowind, uint64_t rowind_len,
                      uint64_t values_len, int *reason);
/* Contiguous row ranges with balanced stored entries, for the row-partitioned
 * multi-GPU product.  bounds has nparts+1 entries,
 * bounds[0] = 0, bounds[nparts] = nrows, non-decreasing. */
int spal_partition_rows(const uint64_t *rowptr, uint64_t nrows,
                        uint32_t nparts, uint64_t *bounds);

/* ---- CSR: y = A * x ------------------------------------------------------
 * Replaces the reference's only route to A*x, `&a * &x_as_matrix`
 * (`impl Mul for &CsrMatrix<T>`, src/csr/ops/mul.rs:5-59); bound as
 * `impl Mul<&[T]> for &CsrMatrix<T>`.
 * create: borrows rowptr()/colind()/values() (src/csr.rs:228-258), checks
 * the constructor's invariants, narrows indices to 32 bits and uploads.
 * Limits: nrows, ncols < 2^32.  The number of stored entries is not limited
 * (the reference's offsets are usize, src/csr.rs:66-72): beyond 2^32 - 65537
 * entries the handle keeps the matrix as row blocks with 32-bit offsets each
 * (spal_csr_describe reports "kernel": "row_blocks" and the cuts); products,
 * download, options and autotune work on the whole; spal_csr_to_csc is then
 * refused (SPAL_ERR_UNSUPPORTED), and so are CSC / COO handles of that size. */
int spal_csr_create_f64(int device, uint64_t nrows, uint64_t ncols,
                        const uint64_t *rowptr, uint64_t rowptr_len,
                        const uint64_t *colind, uint64_t colind_len,
                        const double *values, uint64_t values_len,
                        spal_csr_t *out);
int spal_csr_create_f32(int device, uint64_t nrows, uint64_t ncols,
                        const uint64_t *rowptr, uint64_t rowptr_len,
                        const uint64_t *colind, uint64_t colind_len,
                        const float *values, uint64_t values_len,
                        spal_csr_t *out);
int spal_csr_destroy(spal_csr_t a);
/* nrows()/ncols()/nnz() (src/csr.rs:200-222, :287-289); elem_size 4 or 8. */
int spal_csr_shape(spal_csr_t a, uint64_t *nrows, uint64_t *ncols,
                   uint64_t *nnz, int *elem_size);
/* Host convenience: H2D x, kernel, D2H y.  x_len must equal ncols and y_len
 * nrows (mirrors assert_eq! at src/csr/ops/mul.rs:9).  y is fully
 * overwritten; rows without stored entries give 0.0. */
int spal_csr_spmv_f64(spal_csr_t a, const double *x, uint64_t x_len,
                      double *y, uint64_t y_len);
int spal_csr_spmv_f32(spal_csr_t a, const float *x, uint64_t x_len,
                      float *y, uint64_t y_len);
/* Timed path: x (ncols) and y (nrows) are DEVICE pointers on the handle's
 * device; the launch is enqueued on `stream` (a hipStream_t, NULL = the
 * default stream) and not synchronised.  Safe to call concurrently on one
 * handle (read-only). */
int spal_csr_spmv_dev_f64(spal_csr_t a, const double *x_dev, double *y_dev,
                          void *stream);
int spal_csr_spmv_dev_f32(spal_csr_t a, const float *x_dev, float *y_dev,
                          void *stream);
/* Copies the device matrix back into caller arrays shaped like the
 * reference's fields (rowptr nrows+1, colind nnz, values nnz). */
int spal_csr_download_f64(spal_csr_t a, uint64_t *rowptr, uint64_t *colind,
                          double *values);
int spal_csr_download_f32(spal_csr_t a, uint64_t *rowptr, uint64_t *colind,
                          float *values);
/* Kernel plan knobs (tuning / tests).  Keys: "kernel" (0 = auto, 1 = vector,
 * 2 = stream), "rows_per_block", "lanes_per_row", "lds_x" (-1 auto/0/1),
 * "unroll", "threads", "col16" (long rows: 16-bit columns, default 1) (vector
 * kernel); "rows_per_tile" (0 auto, 256, 128, 64, 32, 24, 16, 12, 8),
 * "tiles_per_wave", "persistent", "persistent_blocks" (0 = what the device
 * holds at once), "nt_store", "stream_global", "window_pages" (0 auto; LDS x
 * window budget in 256-column pages), "stream_row_max" (64-row tiles with a
 * longer row go to the overflow kernel; default 128), "skew" (-1 auto, 0, 1:
 * skewed LDS product strips), "slide" (-1 auto / 0: the sliding-window kernel
 * and its ring-addressed x window for band-like plans), "slide_on" (0 / 1:
 * launch it), "uniform_rows" (0 / 1: super-tiles whose rows all have one length
 * do not read rowptr), "prefetch" (1 / 2 tiles of loads ahead), "place_tries"
 * (autotune), "split_tiles" (1 default / 0: the sliding kernel computes tiles
 * above 1024 entries whose halves fit in two passes instead of leaving them to
 * the overflow kernel), "xcd_chunk", "walk_blocks" (stream kernel); "cblock"
 * (-1 auto / 0 / 1: the column-blocked kernels for columns anywhere),
 * "cblock_form" (-1 by the entries a row holds per column block / 0 entry-
 * parallel / 1 rows form), "cblock_rows", "cblock_shift" (0 auto; rows of a row
 * block, log2 of the columns of a column block); "row_split" (-1 auto / 0 / 1:
 * skewed row lengths -- rows above "row_split_threshold" entries, default 128,
 * are multiplied apart from the rest), "blockwin" (-1 / 0 / 1: the block-window
 * kernel for skewed rows whose columns stay near the rows -- -1: timed against
 * the row split at setup, 1: whenever every row block's window of x fits LDS).
 * Unknown key or a value the kernels are not instantiated for:
 * SPAL_ERR_INVALID_ARGUMENT. */
int spal_csr_set_option(spal_csr_t a, const char *key, int64_t value);
/* Setup-time autotune: runs the planned kernel's variants (the stream kernel
 * with one workgroup per super-tile vs. its walking form -- the sliding-window
 * kernel on band-like plans, else the persistent form -- each with plain or
 * non-temporal y stores; the column-blocked kernel against the stream kernels
 * where the plan built both) `iters` times each on the caller's device vectors
 * and keeps the fastest; then copies the 16-bit column array into
 * "place_tries" (default 8; up to three times as many while none is 3 % better)
 * blocks of 1 GiB taken one after the other from the
 * device's memory and keeps the place where the kernel ran fastest (two streams
 * out of one class of region disturb each other).  All variants
 * produce identical y.  Synchronises `stream`.  Cost at config 3 with
 * iters = 30: 0.1 ... 0.15 s. */
int spal_csr_autotune_f64(spal_csr_t a, const double *x_dev, double *y_dev,
                          void *stream, int iters);
int spal_csr_autotune_f32(spal_csr_t a, const float *x_dev, float *y_dev,
                          void *stream, int iters);
/* Device vectors x (ncols elements) and y (nrows elements) for products with THIS handle, placed so that the stores
 * of y do not collide with the matrix stream.  Why: on MI355X the time of a product depends on where y lies in the
 * device's memory RELATIVE to the matrix arrays (streams out of one class of region disturb each other: +-5 % at
 * config 3) -- a property of the pair that neither side can fix alone, and `hipMalloc` gives no say.
 * The FIRST such call in a process (per device) walks the device's memory in blocks of 1 GiB ("walk_blocks", default 8 =
 * at most 8 GiB held during the walk; ~3 ms each), times the handle's kernel into a candidate y in every block, KEEPS the
 * block where it ran fastest and -- when the walk met a second class of region -- the one where it ran slowest, as the
 * process's placement blocks, and frees the rest.  Every later call (any handle) only times the handle in the kept blocks
 * (two probes, no hipMalloc) and takes its vectors as a PIECE of the better one; spal_csr_autotune_* takes the 16-bit
 * columns from the same blocks.  Memory: at most two blocks of 1 GiB per process and device, shared by all handles
 * (spal_csr_describe: "placement_blocks", "placement_free_bytes"; a handle reports the new blocks its call took in
 * "vectors_walk_blocks" -- 0 once the process has its blocks -- and the places it timed in "vectors_probes").  The
 * vectors belong to the handle (their piece returns to the block with spal_csr_destroy; a second call returns the same
 * pointers); any other device memory works as x / y too, only possibly slower.  Matrices below 256 MB and handles above
 * 2^32 - 65537 entries (row blocks): a plain allocation of exactly the vectors' size.  Synchronises `stream`.
 * Replaces nothing in the reference: its `Vec<T>` has no placement.  Rust shim: `DeviceCsr::vectors()`. */
int spal_csr_alloc_vectors(spal_csr_t a, void **x_dev, void **y_dev, void *stream);
/* Writes a one-line JSON description of the active plan into buf: "kernel"
 * ("stream" | "vector"), "index_bits" (16: window-relative columns), the
 * geometry ("rows_per_tile", "rows_per_block", "lanes_per_row", ...),
 * "lds_window_bytes", "stream_row_fraction" (rows in tiles that stream),
 * "overflow_tiles" (tiles left to the overflow kernel), "skew", "persistent",
 * "slide", "ring_pages", "uniform_row_fraction", "nt_store", "autotune_us"
 * (one workgroup per super-tile, walking form, each then with non-temporal y
 * stores), "placement_us" (before / after placing the 16-bit columns),
 * "vectors_walk_us" (fastest / slowest block of spal_csr_alloc_vectors). */
int spal_csr_describe(spal_csr_t a, char *buf, size_t buf_len);

/* ---- CSC: y = A * x (atomic scatter) --------------------------------------
 * Replaces `&a * &x_as_matrix` for `impl Mul for &CscMatrix<T>`
 * (src/csc/ops/mul.rs:5-60); bound as `impl Mul<&[T]> for &CscMatrix<T>`. */
int spal_csc_create_f64(int device, uint64_t nrows, uint64_t ncols,
                        const uint64_t *colptr, uint64_t colptr_len,
                        const uint64_t *rowind, uint64_t rowind_len,
                        const double *values, uint64_t values_len,
                        spal_csc_t *out);
int spal_csc_create_f32(int device, uint64_t nrows, uint64_t ncols,
                        const uint64_t *colptr, uint64_t colptr_len,
                        const uint64_t *rowind, uint64_t rowind_len,
                        const float *values, uint64_t values_len,
                        spal_csc_t *out);
int spal_csc_destroy(spal_csc_t a);
int spal_csc_shape(spal_csc_t a, uint64_t *nrows, uint64_t *ncols,
                   uint64_t *nnz, int *elem_size);
int spal_csc_spmv_f64(spal_csc_t a, const double *x, uint64_t x_len,
                      double *y, uint64_t y_len);
int spal_csc_spmv_f32(spal_csc_t a, const float *x, uint64_t x_len,
                      float *y, uint64_t y_len);
int spal_csc_spmv_dev_f64(spal_csc_t a, const double *x_dev, double *y_dev,
                          void *stream);
int spal_csc_spmv_dev_f32(spal_csc_t a, const float *x_dev, float *y_dev,
                          void *stream);
int spal_csc_download_f64(spal_csc_t a, uint64_t *colptr, uint64_t *rowind,
                          double *values);
int spal_csc_download_f32(spal_csc_t a, uint64_t *colptr, uint64_t *rowind,
                          float *values);
/* Keys: "kernel" 1 = atomic scatter (over ROW tiles where every tile's window
 * of x fits LDS beside its rows -- a workgroup owns rows of y, nothing is shared
 * between workgroups or launches; "row_tiles" -1 auto / 0 / 1 --, else over
 * column tiles: LDS-privatised where the row window of a
 * 1024-column block fits LDS, global atomics otherwise), 2 = transposed: the
 * matrix is converted to CSR on the device once and the CSR kernels run
 * (deterministic; bit-identical to the reference's k-ascending order), 0 = auto
 * (= 2).  "lds" 0/1, "cols_per_block" (0 auto / 1024 / 2048 / 4096 columns per
 * super-tile), "flush" tune kernel 1's column tiles (a flush other than 0 also
 * selects them).  flush 0 (default): where the super-tiles'
 * row windows ascend and overlap their neighbours' only (bands), every row of y
 * is stored by the first super-tile that covers it and completed by the next one
 * behind a flag -- no zero fill of y, no global atomics; launches of one handle
 * are then chained by an event (any streams), and on a stream that is being
 * captured into a graph the atomics form runs instead.  Otherwise, or with flush
 * 2: y is zeroed and window rows are added with global atomics.  flush 1: windows
 * stored per super-tile, then an ordered reduce.  spal_csc_describe reports the
 * form in "flush". */
int spal_csc_set_option(spal_csc_t a, const char *key, int64_t value);
/* as spal_csr_autotune_* for the transposed route (kernel 2); no-op for kernel 1 */
int spal_csc_autotune_f64(spal_csc_t a, const double *x_dev, double *y_dev,
                          void *stream, int iters);
int spal_csc_autotune_f32(spal_csc_t a, const float *x_dev, float *y_dev,
                          void *stream, int iters);
int spal_csc_describe(spal_csc_t a, char *buf, size_t buf_len);
/* Health of the device-pointer products (spal_csc_spmv_dev_*) issued so far; call it after synchronising their stream.
 * *invalid_products = products of this handle whose neighbour hand-off (kernel 1 over column tiles, flush 0) hit its
 * spin bound: their y was NOT valid.  The library also reports such a product by failing the NEXT product on the
 * handle (SPAL_ERR_HIP) and flushes with atomics from then on; a caller whose last product it was learns it here.
 * Always 0 for the transposed route, the row tiles and the atomics forms (nothing is handed over there).
 * Stream lifetime with the hand-off form: the handle remembers the stream of its last product and, when the next product
 * comes on ANOTHER stream, records an event on the remembered one -- so a stream that carried a product of this handle
 * must stay alive until the handle has been used on another stream or destroyed (or use one stream per handle). */
int spal_csc_status(spal_csc_t a, int *invalid_products);

/* ---- CSR <-> CSC on the device ------------------------------------------------
 * Replace `impl From<&CscMatrix<T>> for CsrMatrix<T>` (src/csr/conv/csc.rs:4-52)
 * and `impl From<&CsrMatrix<T>> for CscMatrix<T>` (src/csc/conv/csr.rs:4-52),
 * i.e. the counting sort of CsrMatrix::transpose (src/csr.rs:358-406): a stable
 * sort of the entries by their minor index.  Entries are only moved, so the
 * result equals the reference's exactly.  The input handle is unchanged; the
 * output is a new, independent handle. */
int spal_csc_to_csr(spal_csc_t a, spal_csr_t *out);
int spal_csr_to_csc(spal_csr_t a, spal_csc_t *out);

/* ---- COO -> CSR assembly on the device -------------------------------------
 * Replaces `impl From<&CooMatrix<T>> for CsrMatrix<T>`
 * (src/csr/conv/coo.rs:4-115): stable order by (row, col), duplicates summed
 * left to right in insertion order, results equal to zero dropped.
 * Triplets are passed as three arrays (Rust does not fix the layout of
 * Vec<(usize, usize, T)>, so a binding unzips `coo.iter()`, src/coo.rs:491). */
int spal_coo_upload_f64(int device, uint64_t nrows, uint64_t ncols, uint64_t len,
                        const uint64_t *rows, const uint64_t *cols,
                        const double *vals, spal_coo_t *out);
int spal_coo_upload_f32(int device, uint64_t nrows, uint64_t ncols, uint64_t len,
                        const uint64_t *rows, const uint64_t *cols,
                        const float *vals, spal_coo_t *out);
int spal_coo_destroy(spal_coo_t c);
/* Device-resident assembly (the timed path): enqueues on `stream`, returns a
 * new CSR handle.  Synchronises the stream once (the output size is data
 * dependent). */
int spal_coo_assemble_csr(spal_coo_t c, void *stream, spal_csr_t *out);
/* The handle spal_coo_assemble_csr / spal_coo_to_csr_* return is the complete matrix `CsrMatrix::from(&coo)` produces
 * (shape, download, conversions); the plan of the PRODUCT kernels on it (tile heights, x windows, 16-bit columns: ~0.3 ms
 * of small kernels and host round trips at 50M entries) is built by whatever needs it first -- the first product,
 * spal_csr_set_option, spal_csr_autotune_*, spal_csr_alloc_vectors, spal_csr_describe -- or, explicitly, here.  A first
 * product cannot be captured into a graph: plan before capturing.  No-op on handles created from host arrays (planned at
 * create time).  SPAL_COO_EAGER_PLAN=1 plans inside the assembly call as rounds 1-3 did. */
int spal_csr_plan(spal_csr_t a);
/* Same assembly compressed by columns: replaces
 * `impl From<&CooMatrix<T>> for CscMatrix<T>` (src/csc/conv/coo.rs:4-115). */
int spal_coo_assemble_csc(spal_coo_t c, void *stream, spal_csc_t *out);
/* JSON description of the handle and of the route its last assembly took
 * ("local_sort" with its tile geometry, or "general"); for logs and tests. */
int spal_coo_describe(spal_coo_t c, char *buf, size_t buf_len);
/* One-call convenience: upload + assemble + free the COO copy. */
int spal_coo_to_csr_f64(int device, uint64_t nrows, uint64_t ncols, uint64_t len,
                        const uint64_t *rows, const uint64_t *cols,
                        const double *vals, spal_csr_t *out);
int spal_coo_to_csr_f32(int device, uint64_t nrows, uint64_t ncols, uint64_t len,
                        const uint64_t *rows, const uint64_t *cols,
                        const float *vals, spal_csr_t *out);

int spal_coo_to_csc_f64(int device, uint64_t nrows, uint64_t ncols, uint64_t len,
                        const uint64_t *rows, const uint64_t *cols,
                        const double *vals, spal_csc_t *out);
int spal_coo_to_csc_f32(int device, uint64_t nrows, uint64_t ncols, uint64_t len,
                        const uint64_t *rows, const uint64_t *cols,
                        const float *vals, spal_csc_t *out);

/* ---- row-partitioned y = A * x over the GPUs of one node, from one process ---
 * (SURVEY.md sections 8e, 8f-4).  Rows are cut into contiguous ranges with
 * balanced stored entries; every GPU holds its range and a full-length x buffer of
 * which it reads only its WINDOW, the columns its rows store.  The product of a
 * range is the same kernel as the single-GPU path, so results are identical.
 * Exchange steps (RCCL over xGMI, loaded lazily; ngpus == 1 needs no RCCL):
 *   x: spal_mg_csr_broadcast_x (ncclBroadcast of all of x from GPU 0) or
 *      spal_mg_csr_scatter_x (grouped ncclSend/ncclRecv: every GPU receives its
 *      window only);
 *   y: spal_mg_csr_gather_y (slices back to back into GPU 0's y, unequal sizes) or
 *      spal_mg_csr_spmv_resident (kernels + ncclAllGather: all of y on every GPU);
 *   iterative use: spal_mg_csr_spmv_halo (per step every GPU receives only the
 *      entries of y its rows read as columns).
 * devices == NULL means GPUs 0 .. ngpus-1.  transport: 0 = RCCL, 1 = peer copies
 * (hipMemcpyPeerAsync ordered by events; also accepts a device list with repeats,
 * i.e. several shards on one GPU), -1 = RCCL unless the list has repeats. */
int spal_mg_create(int ngpus, const int *devices, spal_mg_t *out);
int spal_mg_create_transport(int ngpus, const int *devices, int transport, spal_mg_t *out);
int spal_mg_destroy(spal_mg_t ctx);
int spal_mg_device_count(spal_mg_t ctx, int *ngpus);
int spal_mg_transport(spal_mg_t ctx, int *transport);
int spal_mg_csr_create_f64(spal_mg_t ctx, uint64_t nrows, uint64_t ncols,
                           const uint64_t *rowptr, uint64_t rowptr_len,
                           const uint64_t *colind, uint64_t colind_len,
                           const double *values, uint64_t values_len,
                           spal_mg_csr_t *out);
int spal_mg_csr_create_f32(spal_mg_t ctx, uint64_t nrows, uint64_t ncols,
                           const uint64_t *rowptr, uint64_t rowptr_len,
                           const uint64_t *colind, uint64_t colind_len,
                           const float *values, uint64_t values_len,
                           spal_mg_csr_t *out);
int spal_mg_csr_destroy(spal_mg_csr_t a);
/* the row boundaries in use: ngpus + 1 entries */
int spal_mg_csr_partition(spal_mg_csr_t a, uint64_t *bounds);
/* per GPU (ngpus entries each): its rows store columns in [need_lo, need_hi) only */
int spal_mg_csr_windows(spal_mg_csr_t a, uint64_t *need_lo, uint64_t *need_hi);
/* bytes one scatter_x / gather_y / halo exchange moves between GPUs (any may be NULL) */
int spal_mg_csr_exchange_bytes(spal_mg_csr_t a, uint64_t *x_scatter, uint64_t *y_gather,
                               uint64_t *halo);
/* host vectors: H2D x to GPU 0, x windows scattered (broadcast when the windows
 * cover most of x), local kernels, y gathered on GPU 0, D2H y */
int spal_mg_csr_spmv_f64(spal_mg_csr_t a, const double *x, uint64_t x_len,
                         double *y, uint64_t y_len);
int spal_mg_csr_spmv_f32(spal_mg_csr_t a, const float *x, uint64_t x_len,
                         float *y, uint64_t y_len);
/* resident (timed) path, everything asynchronous on the context's per-GPU
 * streams until spal_mg_csr_synchronize:
 *   write x into GPU 0's buffer (x_root; the pointer changes with every
 *   spmv_halo), then broadcast_x or scatter_x once;
 *   spmv_local any number of times (kernels only), gather_y: GPU 0's y (nrows
 *   elements, y_gathered) holds the result;
 *   or spmv_resident (kernels + all-gather): GPU 0's copy is at y_root as
 *   ngpus slices of slice_stride elements (slice g holds rows bounds[g] ..);
 *   or, square matrices, spmv_halo any number of times (x <- A * x with the halo
 *   exchange), then gather_y. */
int spal_mg_csr_x_root(spal_mg_csr_t a, void **x_dev);
int spal_mg_csr_broadcast_x(spal_mg_csr_t a);
int spal_mg_csr_scatter_x(spal_mg_csr_t a);
int spal_mg_csr_spmv_local(spal_mg_csr_t a);
int spal_mg_csr_gather_y(spal_mg_csr_t a);
int spal_mg_csr_y_gathered(spal_mg_csr_t a, void **y_dev);
int spal_mg_csr_spmv_halo(spal_mg_csr_t a);
int spal_mg_csr_spmv_resident(spal_mg_csr_t a);
int spal_mg_csr_y_root(spal_mg_csr_t a, void **y_dev, uint64_t *slice_stride);
int spal_mg_csr_synchronize(spal_mg_csr_t a);
/* HIP-event durations (ms, the longest over the GPUs) of the LAST x distribution
 * (broadcast_x / scatter_x), local kernels, halo exchange and y collection
 * (gather_y / the all-gather): ms[0..3]; -1 for a phase that has not run.
 * Synchronises. */
int spal_mg_csr_timing(spal_mg_csr_t a, double *ms);

/* ---- device memory helpers for callers without a HIP binding of their own
 * (the Rust shim, ctypes tests, the C++ tools). ---------------------------- */
int spal_dev_malloc(int device, size_t bytes, void **ptr);
int spal_dev_free(int device, void *ptr);
int spal_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes);
int spal_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes);
int spal_device_synchronize(int device);
/* Device blocks released by *_destroy / spal_dev_free are kept in a per-process
 * cache for reuse (bounded by the environment variable SPAL_CACHE_BYTES; default:
 * a quarter of the device's memory, at most half of what was free at first use).
 * spal_cache_trim hands every cached block back to the driver, e.g. before
 * another allocator in the process (torch) needs the memory -- and the placement
 * blocks of spal_csr_alloc_vectors that no live handle holds a piece of. */
int spal_cache_trim(void);

#ifdef __cplusplus
}
#endif
#endif /* SPAL_H */
