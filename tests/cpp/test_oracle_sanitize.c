/* The CPU oracle (oracle/spal_oracle.c) under ASan + UBSan on the reference's
 * G1 / G5 vectors and random inputs (test infrastructure checking itself). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint64_t usize;
int orc_from_coo_f64(usize, usize, usize, const usize *, const usize *, const double *, usize *, usize *, double *, usize *);
int orc_spgemm_f64(usize, usize, const usize *, const usize *, const double *, usize, usize, const usize *, const usize *,
                   const double *, usize **, usize **, double **);
int orc_transpose_f64(usize, usize, const usize *, const usize *, const double *, usize *, usize *, double *);
void orc_csr_spmv_f64(usize, const usize *, const usize *, const double *, const double *, double *);
int orc_csc_spmv_f64(usize, usize, const usize *, const usize *, const double *, const double *, double *);
void orc_free(void *);
static int failures = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

int main(void) {
    { /* G1 src/csr/conv/coo.rs:128-145 */
        usize r[] = {1, 0, 0, 0, 0, 1, 1, 1}, c[] = {2, 2, 1, 0, 0, 0, 1, 1};
        double v[] = {5, 4, 3, 1, 2, 0, 1, -1};
        usize p[3], i[8], nnz = 0; double w[8];
        CHECK(orc_from_coo_f64(2, 3, 8, r, c, v, p, i, w, &nnz) == 0 && nnz == 4);
        CHECK(p[0] == 0 && p[1] == 3 && p[2] == 4 && i[3] == 2 && w[0] == 3 && w[3] == 5);
    }
    { /* G5 src/csc/ops/mul.rs:67-95 */
        usize lp[] = {0, 3, 4, 6}, li[] = {0, 1, 4, 3, 1, 2}; double lv[] = {1, -5, 4, 3, 7, 2};
        usize rp[] = {0, 3, 4, 5, 6}, ri[] = {0, 1, 2, 2, 0, 1}; double rv[] = {1, -5, 7, 3, -2, 4};
        usize *op, *oi; double *ov;
        CHECK(orc_spgemm_f64(5, 3, lp, li, lv, 3, 4, rp, ri, rv, &op, &oi, &ov) == 0);
        CHECK(op[4] == 11 && ov[1] == 44.0 && ov[10] == 12.0 && oi[10] == 3);
        orc_free(op); orc_free(oi); orc_free(ov);
        CHECK(orc_spgemm_f64(5, 3, lp, li, lv, 4, 4, rp, ri, rv, &op, &oi, &ov) == -2);
    }
    srand(3);
    for (int it = 0; it < 300; ++it) { /* random assembly -> spmv both ways */
        usize nr = 1 + rand() % 30, nc = 1 + rand() % 30, n = rand() % 400;
        usize *r = malloc((n + 1) * 8), *c = malloc((n + 1) * 8), *p = malloc((nr + 1) * 8), *i = malloc((n + 1) * 8);
        double *v = malloc((n + 1) * 8), *w = malloc((n + 1) * 8), *x = malloc(nc * 8), *y = malloc(nr * 8), *y2 = malloc(nr * 8);
        for (usize k = 0; k < n; ++k) { r[k] = rand() % nr; c[k] = rand() % nc; v[k] = (rand() % 7) - 3; }
        for (usize k = 0; k < nc; ++k) x[k] = (rand() % 5) - 2;
        usize nnz = 0;
        CHECK(orc_from_coo_f64(nr, nc, n, r, c, v, p, i, w, &nnz) == 0 && nnz <= n);
        orc_csr_spmv_f64(nr, p, i, w, x, y);
        usize *cp = malloc((nc + 1) * 8), *ci = malloc((nnz + 1) * 8); double *cv = malloc((nnz + 1) * 8);
        CHECK(orc_transpose_f64(nr, nc, p, i, w, cp, ci, cv) == 0);
        CHECK(orc_csc_spmv_f64(nr, nc, cp, ci, cv, x, y2) == 0);
        for (usize k = 0; k < nr; ++k) CHECK(y[k] == y2[k]);
        free(r); free(c); free(p); free(i); free(v); free(w); free(x); free(y); free(y2); free(cp); free(ci); free(cv);
    }
    printf("oracle sanitize: %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
