// spal_synth.cpp -- libspal_synth.so: the benchmark / test input generators
// (SURVEY.md section 8d).  Host only, no HIP; see spal_synth.h.
#include "spal_synth.h"

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace {

std::string &err() {
    static thread_local std::string msg;
    return msg;
}
int fail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err() = buf;
    return 1;
}

unsigned host_threads() {
    static unsigned n = [] {
        if (const char *e = getenv("SPAL_HOST_THREADS")) {
            int v = atoi(e);
            if (v > 0) return (unsigned)v;
        }
        unsigned h = std::thread::hardware_concurrency();
        return std::min(h ? h : 1u, 32u);
    }();
    return n;
}

// fn(begin, end) over [0, n) in contiguous chunks, one per thread
void parallel_for(uint64_t n, const std::function<void(uint64_t, uint64_t)> &fn, uint64_t min_chunk = 1u << 16) {
    if (n == 0) return;
    unsigned nt = host_threads();
    const uint64_t by_size = (n + min_chunk - 1) / min_chunk;
    if (by_size < nt) nt = (unsigned)by_size;
    if (nt <= 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    const uint64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const uint64_t b = std::min<uint64_t>(n, (uint64_t)t * per), e = std::min<uint64_t>(n, b + per);
        if (b >= e) break;
        th.emplace_back([&fn, b, e] { fn(b, e); });
    }
    for (auto &t : th) t.join();
}

// SplitMix64.  The state after k draws is seed + k*GAMMA, so any draw is
// addressable without the ones before it.
constexpr uint64_t GAMMA = 0x9E3779B97F4A7C15ull;
constexpr uint64_t ROW_MULT = 0xD1B54A32D192ED03ull;
constexpr uint32_t kRaggedMod = 27;   // row length 1 + next() % 27

inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct SplitMix {
    uint64_t s;
    explicit SplitMix(uint64_t seed) : s(seed) {}
    inline uint64_t next() { s += GAMMA; return mix64(s); }
};
inline uint64_t draw_at(uint64_t seed, uint64_t k) { return mix64(seed + (k + 1) * GAMMA); }
inline double unit_value(uint64_t r) { return 2.0 * ((double)(r >> 11) * (1.0 / 9007199254740992.0)) - 1.0; }
inline uint64_t row_state(uint64_t seed, uint64_t row) { return seed ^ (ROW_MULT * (row + 1)); }

inline uint64_t window_start(uint64_t row, uint64_t nrows, uint64_t ncols, uint64_t window) {
    const uint64_t centre = (uint64_t)(((unsigned __int128)row * ncols) / nrows);
    uint64_t w0 = centre < window / 2 ? 0 : centre - window / 2;
    if (w0 > ncols - window) w0 = ncols - window;
    return w0;
}

// `k` distinct columns of the row's window, ascending, then k values: the tail of a row's stream
template <typename T>
inline void row_entries(SplitMix &rng, uint64_t w0, uint64_t window, uint32_t k, uint64_t *colind, T *values) {
    uint64_t cols[256];
    uint32_t have = 0;
    while (have < k) {
        const uint64_t c = w0 + rng.next() % window;
        bool dup = false;
        for (uint32_t j = 0; j < have; ++j) dup |= (cols[j] == c);
        if (!dup) cols[have++] = c;
    }
    std::sort(cols, cols + k);
    for (uint32_t j = 0; j < k; ++j) {
        colind[j] = cols[j];
        values[j] = (T)unit_value(rng.next());
    }
}

template <typename T>
int gen_banded(uint64_t nrows, uint64_t ncols, uint32_t per_row, uint64_t window, uint64_t seed,
               uint64_t row_begin, uint64_t row_end, uint64_t *rowptr, uint64_t *colind, T *values) {
    if (!rowptr || !colind || !values) return fail("banded: null output");
    if (nrows == 0 || ncols == 0 || per_row == 0 || window == 0 || window > ncols || per_row > window || per_row > 256)
        return fail("banded: need 0 < per_row <= min(window, 256), window <= ncols");
    if (row_begin > row_end || row_end > nrows) return fail("banded: bad row range");
    parallel_for(row_end - row_begin, [&](uint64_t b, uint64_t e) {
        for (uint64_t local = b; local < e; ++local) {
            const uint64_t row = row_begin + local;
            SplitMix rng(row_state(seed, row));
            const uint64_t base = local * per_row;
            rowptr[local] = base;
            row_entries<T>(rng, window_start(row, nrows, ncols, window), window, per_row, colind + base, values + base);
        }
    }, 1u << 12);
    rowptr[row_end - row_begin] = (row_end - row_begin) * per_row;
    return 0;
}

int ragged_rowptr(uint64_t nrows, uint64_t seed, uint64_t row_begin, uint64_t row_end, uint64_t *rowptr) {
    if (!rowptr) return fail("ragged: null output");
    if (row_begin > row_end || row_end > nrows) return fail("ragged: bad row range");
    const uint64_t n = row_end - row_begin;
    // lengths first (the first draw of every row's stream), then a serial prefix sum (40 ms at 10M rows)
    parallel_for(n, [&](uint64_t b, uint64_t e) {
        for (uint64_t local = b; local < e; ++local)
            rowptr[local + 1] = 1 + draw_at(row_state(seed, row_begin + local), 0) % kRaggedMod;
    });
    rowptr[0] = 0;
    for (uint64_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
    return 0;
}

template <typename T>
int ragged_fill(uint64_t nrows, uint64_t ncols, uint64_t window, uint64_t seed, uint64_t row_begin,
                uint64_t row_end, const uint64_t *rowptr, uint64_t *colind, T *values) {
    if (!rowptr || !colind || !values) return fail("ragged: null array");
    if (nrows == 0 || ncols == 0 || window < kRaggedMod || window > ncols)
        return fail("ragged: need 27 <= window <= ncols");
    if (row_begin > row_end || row_end > nrows) return fail("ragged: bad row range");
    std::atomic<int> bad{0};
    parallel_for(row_end - row_begin, [&](uint64_t b, uint64_t e) {
        for (uint64_t local = b; local < e; ++local) {
            const uint64_t row = row_begin + local;
            SplitMix rng(row_state(seed, row));
            const uint32_t len = 1 + (uint32_t)(rng.next() % kRaggedMod);
            if (rowptr[local + 1] - rowptr[local] != len) { bad = 1; return; }
            row_entries<T>(rng, window_start(row, nrows, ncols, window), window, len, colind + rowptr[local],
                           values + rowptr[local]);
        }
    }, 1u << 12);
    return bad.load() ? fail("ragged: rowptr does not come from spal_synth_ragged_rowptr with these arguments") : 0;
}

template <typename T>
int gen_vector(uint64_t n, uint64_t seed, T *x) {
    if (!x && n) return fail("vector: null output");
    parallel_for(n, [&](uint64_t b, uint64_t e) {
        for (uint64_t j = b; j < e; ++j) x[j] = (T)unit_value(draw_at(seed, j));
    });
    return 0;
}

template <typename T>
int gen_coo(uint64_t nrows, uint64_t ncols, uint64_t len, uint64_t seed, uint32_t dup_permille,
            uint32_t cancel_permille, uint64_t *rows, uint64_t *cols, T *vals) {
    if (len && (!rows || !cols || !vals)) return fail("coo: null output");
    if (nrows == 0 || ncols == 0 || dup_permille + cancel_permille > 1000) return fail("coo: bad shape or rates");
    // base triplet i uses draws 3i, 3i+1, 3i+2 of the stream
    auto base_row = [&](uint64_t i) { return draw_at(seed, 3 * i) % nrows; };
    auto base_col = [&](uint64_t i) { return draw_at(seed, 3 * i + 1) % ncols; };
    auto base_val = [&](uint64_t i) { return (T)unit_value(draw_at(seed, 3 * i + 2)); };
    const uint64_t seed2 = mix64(seed ^ 0xA5A5A5A5DEADBEEFull);
    parallel_for(len, [&](uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            uint64_t r = base_row(i), c = base_col(i);
            T v = base_val(i);
            if (i > 0 && (dup_permille || cancel_permille)) {
                const uint64_t h = draw_at(seed2, i);
                const uint32_t t = (uint32_t)(h % 1000);
                const uint64_t j = (h >> 20) % i;  // an earlier entry's BASE triplet
                if (t < dup_permille) {
                    r = base_row(j); c = base_col(j);
                } else if (t < dup_permille + cancel_permille) {
                    r = base_row(j); c = base_col(j); v = -base_val(j);
                }
            }
            rows[i] = r; cols[i] = c; vals[i] = v;
        }
    });
    return 0;
}

}  // namespace

extern "C" {

const char *spal_synth_last_error(void) { return err().c_str(); }

int spal_synth_banded_csr_rows_f64(uint64_t nrows, uint64_t ncols, uint32_t per_row, uint64_t window, uint64_t seed,
                                   uint64_t row_begin, uint64_t row_end, uint64_t *rowptr, uint64_t *colind,
                                   double *values) {
    return gen_banded<double>(nrows, ncols, per_row, window, seed, row_begin, row_end, rowptr, colind, values);
}
int spal_synth_banded_csr_rows_f32(uint64_t nrows, uint64_t ncols, uint32_t per_row, uint64_t window, uint64_t seed,
                                   uint64_t row_begin, uint64_t row_end, uint64_t *rowptr, uint64_t *colind,
                                   float *values) {
    return gen_banded<float>(nrows, ncols, per_row, window, seed, row_begin, row_end, rowptr, colind, values);
}
int spal_synth_ragged_rowptr(uint64_t nrows, uint64_t seed, uint64_t row_begin, uint64_t row_end, uint64_t *rowptr) {
    return ragged_rowptr(nrows, seed, row_begin, row_end, rowptr);
}
int spal_synth_ragged_fill_f64(uint64_t nrows, uint64_t ncols, uint64_t window, uint64_t seed, uint64_t row_begin,
                               uint64_t row_end, const uint64_t *rowptr, uint64_t *colind, double *values) {
    return ragged_fill<double>(nrows, ncols, window, seed, row_begin, row_end, rowptr, colind, values);
}
int spal_synth_ragged_fill_f32(uint64_t nrows, uint64_t ncols, uint64_t window, uint64_t seed, uint64_t row_begin,
                               uint64_t row_end, const uint64_t *rowptr, uint64_t *colind, float *values) {
    return ragged_fill<float>(nrows, ncols, window, seed, row_begin, row_end, rowptr, colind, values);
}
int spal_synth_vector_f64(uint64_t n, uint64_t seed, double *x) { return gen_vector(n, seed, x); }
int spal_synth_vector_f32(uint64_t n, uint64_t seed, float *x) { return gen_vector(n, seed, x); }
int spal_synth_coo_f64(uint64_t nrows, uint64_t ncols, uint64_t len, uint64_t seed, uint32_t dup_permille,
                       uint32_t cancel_permille, uint64_t *rows, uint64_t *cols, double *vals) {
    return gen_coo<double>(nrows, ncols, len, seed, dup_permille, cancel_permille, rows, cols, vals);
}
int spal_synth_coo_f32(uint64_t nrows, uint64_t ncols, uint64_t len, uint64_t seed, uint32_t dup_permille,
                       uint32_t cancel_permille, uint64_t *rows, uint64_t *cols, float *vals) {
    return gen_coo<float>(nrows, ncols, len, seed, dup_permille, cancel_permille, rows, cols, vals);
}

}  // extern "C"
