// Host-only parts of libspal_hip (spal_host.cpp: the constructor invariants, the row
// partitioner) and the bench input generators (spal_synth/spal_synth.cpp) under ASan +
// UBSan, driven with edge-case and fuzzed inputs.  Built with g++ (no device code involved).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "spal.h"
#include "spal_synth.h"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

int main() {
    // the reference's should_panic cases (src/csr.rs:470-510) + accepted ones
    int reason = -1;
    { uint64_t rp[] = {0, 1, 1}, ci[] = {0}; CHECK(spal_csr_validate(0, 1, rp, 3, ci, 1, 1, &reason) == SPAL_ERR_INVARIANT && reason == 1); }
    { uint64_t rp[] = {0, 1}, ci[] = {0}; CHECK(spal_csr_validate(2, 1, rp, 2, ci, 1, 1, &reason) == SPAL_ERR_INVARIANT && reason == 3); }
    { uint64_t rp[] = {0, 2, 2}, ci[] = {1, 0}; CHECK(spal_csr_validate(2, 2, rp, 3, ci, 2, 2, &reason) == SPAL_ERR_INVARIANT && reason == 9); }
    { uint64_t rp[] = {0, 1, 3}, ci[] = {0, 1, 2}; CHECK(spal_csr_validate(2, 3, rp, 3, ci, 3, 3, &reason) == SPAL_OK && reason == 0); }
    { uint64_t cp[] = {0, 1, 1}, ri[] = {1}; CHECK(spal_csc_validate(1, 2, cp, 3, ri, 1, 1, &reason) == SPAL_ERR_INVARIANT && reason == 8); }
    CHECK(spal_csr_validate(1, 1, nullptr, 0, nullptr, 0, 0, &reason) == SPAL_ERR_INVARIANT);  // empty rowptr: len != nrows + 1
    // fuzz: random (mostly invalid) inputs must never read out of bounds
    std::mt19937_64 rng(7);
    for (int it = 0; it < 20000; ++it) {
        const uint64_t nr = rng() % 6, nc = rng() % 6;
        std::vector<uint64_t> rp(rng() % 8), ci(rng() % 9);
        uint64_t acc = 0;
        for (auto &v : rp) { if (rng() % 4) acc += rng() % 3; else acc = rng() % 9; v = acc; }
        if (!rp.empty() && rng() % 3) rp[0] = 0;
        for (auto &v : ci) v = rng() % 7;
        (void)spal_csr_validate(nr, nc, rp.data(), rp.size(), ci.data(), ci.size(), rng() % 3 ? ci.size() : rng() % 9, &reason);
        (void)spal_csc_validate(nr, nc, rp.data(), rp.size(), ci.data(), ci.size(), ci.size(), nullptr);
    }
    // partition
    {
        std::vector<uint64_t> rp(1001, 0);
        for (int i = 1; i <= 1000; ++i) rp[i] = rp[i - 1] + rng() % 40;
        for (uint32_t parts : {1u, 2u, 7u, 8u, 1000u, 1500u}) {
            std::vector<uint64_t> b(parts + 1, 99);
            CHECK(spal_partition_rows(rp.data(), 1000, parts, b.data()) == SPAL_OK);
            CHECK(b[0] == 0 && b[parts] == 1000);
            for (uint32_t g = 0; g < parts; ++g) CHECK(b[g] <= b[g + 1]);
        }
        uint64_t b1[2];
        CHECK(spal_partition_rows(rp.data(), 1000, 0, b1) == SPAL_ERR_INVALID_ARGUMENT);
    }
    // generators: shapes at the edges of their contracts
    {
        const uint64_t n = 4097;
        std::vector<uint64_t> rp(n + 1), ci(n * 14);
        std::vector<double> va(n * 14);
        CHECK(spal_synth_banded_csr_rows_f64(n, n, 14, 64, 1, 0, n, rp.data(), ci.data(), va.data()) == 0);
        CHECK(spal_csr_validate(n, n, rp.data(), n + 1, ci.data(), n * 14, n * 14, &reason) == SPAL_OK);
        CHECK(spal_synth_banded_csr_rows_f64(n, n, 14, n, 1, 0, n, rp.data(), ci.data(), va.data()) == 0);       // window == ncols
        CHECK(spal_synth_banded_csr_rows_f64(n, n, 14, n + 1, 1, 0, n, rp.data(), ci.data(), va.data()) != 0);
        CHECK(spal_synth_banded_csr_rows_f64(n, n, 257, n, 1, 0, n, rp.data(), ci.data(), va.data()) != 0);
        std::vector<uint64_t> srp(101), sci(100 * 14);
        std::vector<float> sva(100 * 14);
        CHECK(spal_synth_banded_csr_rows_f32(n, n, 14, 64, 1, n - 100, n, srp.data(), sci.data(), sva.data()) == 0);
        CHECK(spal_synth_banded_csr_rows_f32(n, n, 14, 64, 1, n - 100, n + 1, srp.data(), sci.data(), sva.data()) != 0);
        // ragged rows (1 ... 27 entries): a valid CsrMatrix, a slice equals the same rows of the whole
        std::vector<uint64_t> grp(n + 1);
        CHECK(spal_synth_ragged_rowptr(n, 9, 0, n, grp.data()) == 0);
        std::vector<uint64_t> gci(grp[n]);
        std::vector<double> gva(grp[n]);
        CHECK(spal_synth_ragged_fill_f64(n, n, 64, 9, 0, n, grp.data(), gci.data(), gva.data()) == 0);
        CHECK(spal_csr_validate(n, n, grp.data(), n + 1, gci.data(), grp[n], grp[n], &reason) == SPAL_OK);
        for (uint64_t r = 0; r < n; ++r) CHECK(grp[r + 1] - grp[r] >= 1 && grp[r + 1] - grp[r] <= 27);
        std::vector<uint64_t> hrp(101);
        CHECK(spal_synth_ragged_rowptr(n, 9, n - 100, n, hrp.data()) == 0);
        CHECK(hrp[100] == grp[n] - grp[n - 100]);
        std::vector<uint64_t> hci(hrp[100]);
        std::vector<double> hva(hrp[100]);
        CHECK(spal_synth_ragged_fill_f64(n, n, 64, 9, n - 100, n, hrp.data(), hci.data(), hva.data()) == 0);
        for (uint64_t i = 0; i < hrp[100]; ++i) CHECK(hci[i] == gci[grp[n - 100] + i] && hva[i] == gva[grp[n - 100] + i]);
        CHECK(spal_synth_ragged_fill_f64(n, n, 64, 10, 0, n, grp.data(), gci.data(), gva.data()) != 0);   // rowptr of another seed
        CHECK(spal_synth_ragged_fill_f64(n, n, 26, 9, 0, n, grp.data(), gci.data(), gva.data()) != 0);    // window < 27
        std::vector<uint64_t> r(100000), c(100000);
        std::vector<double> v(100000);
        CHECK(spal_synth_coo_f64(10, 3, 100000, 5, 10, 1, r.data(), c.data(), v.data()) == 0);
        for (size_t i = 0; i < r.size(); ++i) CHECK(r[i] < 10 && c[i] < 3);
        std::vector<float> x(12345);
        CHECK(spal_synth_vector_f32(x.size(), 3, x.data()) == 0);
        CHECK(spal_synth_vector_f64(0, 3, nullptr) == 0);
    }
    CHECK(spal_last_error() != nullptr && spal_version() != nullptr);
    printf("host sanitize: %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
