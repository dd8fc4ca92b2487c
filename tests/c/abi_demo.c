/* Plain C99 caller of the C ABI (include/spal.h): what a non-Rust, non-C++ host sees.
 * Without a GPU: prints the version, counts devices, runs the host-side validation.
 * With a GPU (argv[1] = "gpu"): y = A*x through spal_csr_create_f64 / spal_csr_spmv_f64 on the
 * reference's own 2x3 matrix (src/csr/conv/coo.rs:128-145, KAT G1) assembled from its COO
 * triplets by spal_coo_to_csr_f64, compared with the values worked out by hand. */
#include <stdio.h>
#include <string.h>

#include "spal.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int st_ = (call);                                                              \
        if (st_ != SPAL_OK) {                                                          \
            fprintf(stderr, "%s failed: status %d: %s\n", #call, st_, spal_last_error()); \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main(int argc, char **argv) {
    int count = -1, reason = 0;
    /* G1: pushes (1,2,5) (0,2,4) (0,1,3) (0,0,1) (0,0,2) (1,0,0) (1,1,1) (1,1,-1) into 2 x 3 */
    const uint64_t rows[8] = {1, 0, 0, 0, 0, 1, 1, 1}, cols[8] = {2, 2, 1, 0, 0, 0, 1, 1};
    const double vals[8] = {5, 4, 3, 1, 2, 0, 1, -1};
    const uint64_t rowptr[3] = {0, 3, 4}, colind[4] = {0, 1, 2, 2};
    const uint64_t bad_colind[4] = {0, 2, 1, 2}; /* not increasing inside row 0 */
    printf("%s\n", spal_version());
    CHECK(spal_device_count(&count));
    printf("devices: %d\n", count);
    CHECK(spal_csr_validate(2, 3, rowptr, 3, colind, 4, 4, &reason));
    if (spal_csr_validate(2, 3, rowptr, 3, bad_colind, 4, 4, &reason) != SPAL_ERR_INVARIANT || reason != 9) {
        fprintf(stderr, "validation did not reject unsorted columns (reason %d)\n", reason);
        return 1;
    }
    {   /* KAT G9 (src/coo.rs:876-886, 1000-1012): entries outside the matrix are refused by the ABI itself,
         * before any device is touched */
        const uint64_t r_bad[1] = {1}, c_ok[1] = {0}, r_ok[1] = {0}, c_bad[1] = {1};
        const double one[1] = {1.0};
        spal_coo_t coo = NULL;
        spal_csr_t csr = NULL;
        if (spal_coo_upload_f64(0, 1, 1, 1, r_bad, c_ok, one, &coo) != SPAL_ERR_INDEX_OUT_OF_BOUNDS || coo != NULL ||
            !strstr(spal_last_error(), "row < nrows")) {
            fprintf(stderr, "row outside the matrix accepted: %s\n", spal_last_error());
            return 1;
        }
        if (spal_coo_to_csr_f64(0, 1, 1, 1, r_ok, c_bad, one, &csr) != SPAL_ERR_INDEX_OUT_OF_BOUNDS || csr != NULL ||
            !strstr(spal_last_error(), "col < ncols")) {
            fprintf(stderr, "column outside the matrix accepted: %s\n", spal_last_error());
            return 1;
        }
        if (spal_coo_upload_f64(0, 0, 1, 0, NULL, NULL, NULL, &coo) != SPAL_ERR_INVARIANT) { /* src/coo.rs:819-823 */
            fprintf(stderr, "nrows == 0 accepted\n");
            return 1;
        }
        printf("coo rejections ok\n");
    }
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) {
        spal_csr_t a = NULL;
        uint64_t nr = 0, nc = 0, nnz = 0, rp[3], ci[4];
        int esz = 0;
        double va[4], y[2];
        const double x[3] = {1.0, 10.0, 100.0};
        char plan[1024];
        CHECK(spal_coo_to_csr_f64(0, 2, 3, 8, rows, cols, vals, &a));
        CHECK(spal_csr_shape(a, &nr, &nc, &nnz, &esz));
        if (nr != 2 || nc != 3 || nnz != 4 || esz != 8) { fprintf(stderr, "wrong shape\n"); return 1; }
        CHECK(spal_csr_download_f64(a, rp, ci, va));
        if (memcmp(rp, rowptr, sizeof rp) || memcmp(ci, colind, sizeof ci) || va[0] != 3 || va[1] != 3 ||
            va[2] != 4 || va[3] != 5) { fprintf(stderr, "assembly differs from KAT G1\n"); return 1; }
        CHECK(spal_csr_spmv_f64(a, x, 3, y, 2));
        if (y[0] != 3.0 + 30.0 + 400.0 || y[1] != 500.0) { fprintf(stderr, "wrong product: %g %g\n", y[0], y[1]); return 1; }
        if (spal_csr_spmv_f64(a, x, 2, y, 2) != SPAL_ERR_INVALID_ARGUMENT) { fprintf(stderr, "dimension mismatch accepted\n"); return 1; }
        CHECK(spal_csr_describe(a, plan, sizeof plan));
        printf("%s\n", plan);
        CHECK(spal_csr_destroy(a));
        printf("gpu ok\n");
    }
    return 0;
}
