/*
 * spal_synth.h -- synthetic inputs of BASELINE.json's configs (SURVEY.md
 * section 8d) for bench.py, the tests and the tools.  NOT part of the product:
 * libspal_hip.so (include/spal.h) does not contain these; they live in their
 * own host-only library, libspal_synth.so (g++, no HIP).
 *
 * SplitMix64 based, bit-exact across implementations (tests/test_synth.py
 * restates them in pure Python); threaded.  All functions return 0 on success,
 * 1 on a bad argument (message: spal_synth_last_error()).
 */
#ifndef SPAL_SYNTH_H
#define SPAL_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *spal_synth_last_error(void);

/* banded: every row has exactly `per_row` distinct sorted columns drawn from a
 * window of `window` columns centred on the diagonal (window == ncols gives
 * the uniform "stress" distribution); values in [-1, 1).  Rows
 * [row_begin, row_end) only (a rank's shard of the row-partitioned product);
 * rowptr has row_end - row_begin + 1 entries and starts at 0. */
int spal_synth_banded_csr_rows_f64(uint64_t nrows, uint64_t ncols, uint32_t per_row,
                                   uint64_t window, uint64_t seed, uint64_t row_begin,
                                   uint64_t row_end, uint64_t *rowptr, uint64_t *colind,
                                   double *values);
int spal_synth_banded_csr_rows_f32(uint64_t nrows, uint64_t ncols, uint32_t per_row,
                                   uint64_t window, uint64_t seed, uint64_t row_begin,
                                   uint64_t row_end, uint64_t *rowptr, uint64_t *colind,
                                   float *values);

/* ragged (the robustness variant of SURVEY.md section 8d): the same per-row
 * stream, but its first draw gives the row's length, 1 + next() % 27 (mean
 * 14); then the columns as above, then the values.  Two calls: _rowptr fills
 * rowptr (rebased to 0) so the caller can size colind / values
 * (rowptr[row_end - row_begin] entries), _fill writes them. */
int spal_synth_ragged_rowptr(uint64_t nrows, uint64_t seed, uint64_t row_begin, uint64_t row_end,
                             uint64_t *rowptr);
int spal_synth_ragged_fill_f64(uint64_t nrows, uint64_t ncols, uint64_t window, uint64_t seed,
                               uint64_t row_begin, uint64_t row_end, const uint64_t *rowptr,
                               uint64_t *colind, double *values);
int spal_synth_ragged_fill_f32(uint64_t nrows, uint64_t ncols, uint64_t window, uint64_t seed,
                               uint64_t row_begin, uint64_t row_end, const uint64_t *rowptr,
                               uint64_t *colind, float *values);

/* sequential 2u-1 stream: x vectors */
int spal_synth_vector_f64(uint64_t n, uint64_t seed, double *x);
int spal_synth_vector_f32(uint64_t n, uint64_t seed, float *x);

/* `len` uniform random triplets row=r()%nrows, col=r()%ncols, val=2u-1; then,
 * deterministically, `dup_permille` per mille of the entries are overwritten
 * by exact copies of an earlier entry's (row, col) and `cancel_permille` per
 * mille by an earlier entry with the value negated. */
int spal_synth_coo_f64(uint64_t nrows, uint64_t ncols, uint64_t len, uint64_t seed,
                       uint32_t dup_permille, uint32_t cancel_permille, uint64_t *rows,
                       uint64_t *cols, double *vals);
int spal_synth_coo_f32(uint64_t nrows, uint64_t ncols, uint64_t len, uint64_t seed,
                       uint32_t dup_permille, uint32_t cancel_permille, uint64_t *rows,
                       uint64_t *cols, float *vals);

#ifdef __cplusplus
}
#endif
#endif /* SPAL_SYNTH_H */
