// C++ host-mirror test (include/spalinalg.hpp over the C ABI).
//   ./test_mirror host   -- no GPU needed: constructor panics, accessors
//   ./test_mirror gpu    -- the reference's known-answer vectors on the device
// Reads like the reference's own #[test]s (src/csr.rs:466-511,
// src/csr/conv/coo.rs:128-145, src/csc/ops/mul.rs:67-95).
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "spalinalg.hpp"

using namespace spalinalg;
static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

static bool panics(const std::function<void()> &f) {
    try { f(); } catch (const Panic &) { return true; }
    return false;
}

static void host_tests() {
    // should_panic cases of src/csr.rs:470-510
    CHECK(panics([] { CsrMatrix<double>(0, 1, {0, 1, 1}, {0}, {1.0}); }));       // new_invalid_nrows
    CHECK(panics([] { CsrMatrix<double>(2, 0, {0, 1, 1}, {0}, {1.0}); }));       // new_invalid_ncols
    CHECK(panics([] { CsrMatrix<double>(2, 1, {1, 1, 1}, {0}, {1.0}); }));       // first_not_zero
    CHECK(panics([] { CsrMatrix<double>(2, 1, {0, 1}, {0}, {1.0}); }));          // invalid_length
    CHECK(panics([] { CsrMatrix<double>(2, 1, {0, 1, 1}, {1}, {1.0}); }));       // invalid index
    CHECK(panics([] { CsrMatrix<double>(2, 1, {0, 2, 2}, {1, 0}, {1.0, 2.0}); })); // unsorted
    CHECK(panics([] { CsrMatrix<double>(2, 1, {0, 1, 1}, {0}, {1.0, 2.0}); }));  // values length
    // src/csc.rs:470-510
    CHECK(panics([] { CscMatrix<double>(0, 2, {0, 1, 1}, {0}, {1.0}); }));
    CHECK(panics([] { CscMatrix<double>(1, 0, {0, 1, 1}, {0}, {1.0}); }));
    CHECK(panics([] { CscMatrix<double>(1, 2, {1, 1, 1}, {0}, {1.0}); }));
    CHECK(panics([] { CscMatrix<double>(1, 2, {0, 1}, {0}, {1.0}); }));
    CHECK(panics([] { CscMatrix<double>(1, 2, {0, 1, 1}, {1}, {1.0}); }));
    CHECK(panics([] { CscMatrix<double>(1, 2, {0, 2, 2}, {1, 0}, {1.0, 2.0}); }));
    CHECK(panics([] { CscMatrix<double>(1, 2, {0, 1, 1}, {0}, {1.0, 2.0}); }));
    // doc-test src/csr.rs:116-122
    CsrMatrix<double> m(2, 3, {0, 1, 3}, {0, 1, 2}, {1.0, 2.0, 3.0});
    CHECK(m.nrows() == 2 && m.ncols() == 3 && m.nnz() == 3);
    CHECK(m.rowptr() == (std::vector<usize>{0, 1, 3}));
    // dimension mismatch panics before any device is touched (src/csr/ops/mul.rs:9)
    CHECK(panics([&] { (void)(m * std::vector<double>{1.0, 2.0}); }));
    // CooMatrix (src/coo.rs)
    CHECK(panics([] { CooMatrix<double>(0, 1); }));
    CooMatrix<float> c(2, 3);
    c.push(1, 2, 5.0f);
    CHECK(c.length() == 1);
    CHECK(panics([&] { c.push(2, 0, 1.0f); }));
    CHECK(panics([&] { c.push(0, 3, 1.0f); }));
    CHECK(panics([] { CooMatrix<double>::with_triplets(2, 2, {0}, {0, 1}, {1.0}); }));
}

static void gpu_tests() {
    // src/csr/conv/coo.rs:128-145
    CooMatrix<double> coo(2, 3);
    coo.push(1, 2, 5.0); coo.push(0, 2, 4.0); coo.push(0, 1, 3.0); coo.push(0, 0, 1.0);
    coo.push(0, 0, 2.0); coo.push(1, 0, 0.0); coo.push(1, 1, 1.00); coo.push(1, 1, -1.0);
    auto csr = CsrMatrix<double>::from(coo);
    CHECK(csr.rowptr() == (std::vector<usize>{0, 3, 4}));
    CHECK(csr.colind() == (std::vector<usize>{0, 1, 2, 2}));
    CHECK(csr.values() == (std::vector<double>{3.0, 3.0, 4.0, 5.0}));
    CHECK((csr * std::vector<double>{1.0, 1.0, 1.0}) == (std::vector<double>{10.0, 5.0}));
    // src/csc/conv/coo.rs:128-145 (same pushes, compressed by column)
    auto csc = CscMatrix<double>::from(coo);
    CHECK(csc.colptr() == (std::vector<usize>{0, 1, 2, 4}));
    CHECK(csc.rowind() == (std::vector<usize>{0, 0, 0, 1}));
    CHECK(csc.values() == (std::vector<double>{3.0, 3.0, 4.0, 5.0}));
    // src/csr/conv/csc.rs:65-78 and src/csc/conv/csr.rs:65-78
    auto csr2 = CsrMatrix<double>::from(csc);
    CHECK(csr2.rowptr() == csr.rowptr() && csr2.colind() == csr.colind() && csr2.values() == csr.values());
    auto csc2 = CscMatrix<double>::from(csr);
    CHECK(csc2.colptr() == csc.colptr() && csc2.rowind() == csc.rowind() && csc2.values() == csc.values());
    // src/csc/ops/mul.rs:67-95: every rhs column is an x, every result column its y
    CscMatrix<double> lhs(5, 3, {0, 3, 4, 6}, {0, 1, 4, 3, 1, 2}, {1.0, -5.0, 4.0, 3.0, 7.0, 2.0});
    CHECK((lhs * std::vector<double>{1.0, -5.0, 7.0}) == (std::vector<double>{1.0, 44.0, 14.0, -15.0, 4.0}));
    CHECK((lhs * std::vector<double>{0.0, 0.0, 3.0}) == (std::vector<double>{0.0, 21.0, 6.0, 0.0, 0.0}));
    CHECK((lhs * std::vector<double>{-2.0, 0.0, 0.0}) == (std::vector<double>{-2.0, 10.0, 0.0, 0.0, -8.0}));
    CHECK((lhs * std::vector<double>{0.0, 4.0, 0.0}) == (std::vector<double>{0.0, 0.0, 0.0, 12.0, 0.0}));
    // the same matrix in CSR (5x3)
    CsrMatrix<double> a(5, 3, {0, 1, 3, 4, 5, 6}, {0, 0, 2, 2, 1, 0}, {1.0, -5.0, 7.0, 2.0, 3.0, 4.0});
    CHECK((a * std::vector<double>{1.0, -5.0, 7.0}) == (std::vector<double>{1.0, 44.0, 14.0, -15.0, 4.0}));
    CsrMatrix<float> af(5, 3, {0, 1, 3, 4, 5, 6}, {0, 0, 2, 2, 1, 0}, {1.0f, -5.0f, 7.0f, 2.0f, 3.0f, 4.0f});
    CHECK((af * std::vector<float>{1.0f, -5.0f, 7.0f}) == (std::vector<float>{1.0f, 44.0f, 14.0f, -15.0f, 4.0f}));
}

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "host";
    host_tests();
    if (mode == "gpu") {
        try { gpu_tests(); } catch (const std::exception &e) { printf("FAIL exception: %s\n", e.what()); ++failures; }
    } else {
        // without a device the product must fail loudly, never compute on the host
        int n = 0;
        spal_device_count(&n);
        if (n == 0) {
            bool loud = false;
            try {
                CsrMatrix<double> m(2, 3, {0, 1, 3}, {0, 1, 2}, {1.0, 2.0, 3.0});
                (void)(m * std::vector<double>{1.0, 1.0, 1.0});
            } catch (const Error &e) { loud = (e.status == SPAL_ERR_NO_DEVICE); }
            CHECK(loud);
        }
    }
    printf("%s: %d failure(s)\n", mode.c_str(), failures);
    return failures ? 1 : 0;
}
