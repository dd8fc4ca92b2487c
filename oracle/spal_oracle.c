/*
 * spal_oracle.c -- CPU restatement of the lokyhark/spalinalg hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 *
 * Parity status: PINNED by the reference's own known-answer tests
 * (tests/golden/reference_kats.json, transcribed from the reference's
 * #[test] / doc-test vectors G1..G8).  The reference
 * is Rust and there is no rustc/cargo in the build image, so the reference
 * itself cannot be compiled here (oracle/_ref does not exist for this repo).
 *
 * Every function cites the reference file:line it restates (paths relative
 * to /root/reference).  Build with -O2 -ffp-contract=off: Rust never fuses a
 * multiply and an add, so neither may this file (see oracle/Makefile).
 *
 * Index type: the reference uses `usize`; here uint64_t (x86-64 Linux).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t usize;

/* ------------------------------------------------------------------------
 * Constructor validation.
 * Restates CsrMatrix::new  src/csr.rs:144-156  (CscMatrix::new
 * src/csc.rs:144-156 is the same code with rows/cols swapped: call with
 * (nmajor = ncols, nminor = nrows, ptr = colptr, ind = rowind)).
 * Returns 0 when the reference would construct the matrix, otherwise the
 * 1-based ordinal of the first assertion that would panic:
 *   1 nrows > 0            (:144)      2 ncols > 0              (:145)
 *   3 rowptr.len()==nrows+1 (:146)     4 rowptr[0] == 0         (:147)
 *   5 colind.len()==rowptr[nrows] (:148)
 *   6 values.len()==rowptr[nrows] (:149)
 *   7 rowptr non-decreasing (:150)     8 every colind < ncols   (:151)
 *   9 colind strictly increasing inside each row (:152-156)
 * Note on order: the reference names the dims (nrows, ncols) in both
 * formats, so for CSC "1" is still nrows and "2" still ncols; the caller
 * passes which is which through `major_is_rows`.
 * ---------------------------------------------------------------------- */
int orc_compressed_validate(usize nrows, usize ncols, int major_is_rows,
                            const usize *ptr, usize ptr_len,
                            const usize *ind, usize ind_len, usize val_len)
{
    usize nmajor = major_is_rows ? nrows : ncols;
    usize nminor = major_is_rows ? ncols : nrows;
    if (!(nrows > 0)) return 1;
    if (!(ncols > 0)) return 2;
    if (!(ptr_len == nmajor + 1)) return 3;
    if (!(ptr[0] == 0)) return 4;
    if (!(ind_len == ptr[nmajor])) return 5;
    if (!(val_len == ptr[nmajor])) return 6;
    for (usize i = 0; i + 1 < ptr_len; ++i)
        if (!(ptr[i] <= ptr[i + 1])) return 7;
    for (usize p = 0; p < ind_len; ++p)
        if (!(ind[p] < nminor)) return 8;
    for (usize m = 0; m < nmajor; ++m)
        for (usize p = ptr[m]; p + 1 < ptr[m + 1]; ++p)
            if (!(ind[p] < ind[p + 1])) return 9;
    return 0;
}

/* ------------------------------------------------------------------------
 * y = A * x for CSR, dense x.
 * Derived from `impl Mul for &CsrMatrix<T>` with an ncols x 1 right-hand
 * side: src/csr/ops/mul.rs:25-45.  Per output row the products are taken in
 * ascending stored order (= ascending column, guaranteed by
 * src/csr.rs:152-156); the first product is ASSIGNED (:34), later ones are
 * added with `+=` (:37); multiply and add are rounded separately.  A row
 * with no stored entry is structurally absent in the reference's result,
 * i.e. dense 0.0.
 * ---------------------------------------------------------------------- */
#define DEF_CSR_SPMV(NAME, T)                                                 \
void NAME(usize nrows, const usize *rowptr, const usize *colind,              \
          const T *values, const T *x, T *y)                                  \
{                                                                             \
    for (usize row = 0; row < nrows; ++row) {                                 \
        usize p = rowptr[row], e = rowptr[row + 1];                           \
        if (p == e) { y[row] = (T)0; continue; }                              \
        T acc = values[p] * x[colind[p]];                                     \
        for (++p; p < e; ++p) {                                               \
            T prod = values[p] * x[colind[p]];                                \
            acc += prod;                                                      \
        }                                                                     \
        y[row] = acc;                                                         \
    }                                                                         \
}
DEF_CSR_SPMV(orc_csr_spmv_f64, double)
DEF_CSR_SPMV(orc_csr_spmv_f32, float)

/* Same arithmetic on 32-bit indices: used only by bench.py's cpu_baseline
 * leg so the timed CPU loop reads the same bytes per entry as the GPU. */
#define DEF_CSR_SPMV32(NAME, T)                                               \
void NAME(usize nrows, const uint32_t *rowptr, const uint32_t *colind,        \
          const T *values, const T *x, T *y)                                  \
{                                                                             \
    for (usize row = 0; row < nrows; ++row) {                                 \
        uint32_t p = rowptr[row], e = rowptr[row + 1];                        \
        if (p == e) { y[row] = (T)0; continue; }                              \
        T acc = values[p] * x[colind[p]];                                     \
        for (++p; p < e; ++p) {                                               \
            T prod = values[p] * x[colind[p]];                                \
            acc += prod;                                                      \
        }                                                                     \
        y[row] = acc;                                                         \
    }                                                                         \
}
DEF_CSR_SPMV32(orc_csr_spmv_idx32_f64, double)
DEF_CSR_SPMV32(orc_csr_spmv_idx32_f32, float)

/* The same loop over the rows [row_begin, row_end) only: rows are independent,
 * so bench.py's informational all-cores figure runs one range per thread
 * (results identical to the sequential call). */
#define DEF_CSR_SPMV32_ROWS(NAME, T)                                          \
void NAME(usize row_begin, usize row_end, const uint32_t *rowptr,             \
          const uint32_t *colind, const T *values, const T *x, T *y)          \
{                                                                             \
    for (usize row = row_begin; row < row_end; ++row) {                       \
        uint32_t p = rowptr[row], e = rowptr[row + 1];                        \
        if (p == e) { y[row] = (T)0; continue; }                              \
        T acc = values[p] * x[colind[p]];                                     \
        for (++p; p < e; ++p) {                                               \
            T prod = values[p] * x[colind[p]];                                \
            acc += prod;                                                      \
        }                                                                     \
        y[row] = acc;                                                         \
    }                                                                         \
}
DEF_CSR_SPMV32_ROWS(orc_csr_spmv_idx32_rows_f64, double)
DEF_CSR_SPMV32_ROWS(orc_csr_spmv_idx32_rows_f32, float)

/* ------------------------------------------------------------------------
 * y = A * x for CSC, dense x.
 * Derived from `impl Mul for &CscMatrix<T>` src/csc/ops/mul.rs:26-46 with
 * an ncols x 1 right-hand side: for each y[i] the contributions arrive in
 * ascending k; first touch assigns (:34), later touches `+=` (:37).
 * `touched` plays the role of the reference's `set` workspace (:21,:32-33).
 * Rows never touched are structural zeros -> 0.0.
 * ---------------------------------------------------------------------- */
#define DEF_CSC_SPMV(NAME, T)                                                 \
int NAME(usize nrows, usize ncols, const usize *colptr, const usize *rowind,  \
         const T *values, const T *x, T *y)                                   \
{                                                                             \
    unsigned char *touched = (unsigned char *)calloc(nrows ? nrows : 1, 1);   \
    if (!touched) return -1;                                                  \
    for (usize i = 0; i < nrows; ++i) y[i] = (T)0;                            \
    for (usize k = 0; k < ncols; ++k) {                                       \
        for (usize p = colptr[k]; p < colptr[k + 1]; ++p) {                   \
            usize i = rowind[p];                                              \
            T prod = values[p] * x[k];                                        \
            if (!touched[i]) { touched[i] = 1; y[i] = prod; }                 \
            else             { y[i] += prod; }                                \
        }                                                                     \
    }                                                                         \
    free(touched);                                                            \
    return 0;                                                                 \
}
DEF_CSC_SPMV(orc_csc_spmv_f64, double)
DEF_CSC_SPMV(orc_csc_spmv_f32, float)

/* ------------------------------------------------------------------------
 * Transpose / CSR<->CSC conversion: counting sort by the minor index.
 * Restates CsrMatrix::transpose src/csr.rs:358-406 (count :367-373, prefix
 * :376-382, scatter :385-396); `From<&CsrMatrix> for CscMatrix`
 * src/csc/conv/csr.rs:4-52 and `From<&CscMatrix> for CsrMatrix`
 * src/csr/conv/csc.rs:4-52 are the same three loops.
 * in : nmajor x nminor compressed by major (ptr, ind, val)
 * out: optr[nminor+1], oind[nnz], oval[nnz]  (compressed by minor)
 * ---------------------------------------------------------------------- */
#define DEF_TRANSPOSE(NAME, T)                                                \
int NAME(usize nmajor, usize nminor, const usize *ptr, const usize *ind,      \
         const T *val, usize *optr, usize *oind, T *oval)                     \
{                                                                             \
    usize *vec = (usize *)calloc(nminor ? nminor : 1, sizeof(usize));         \
    if (!vec) return -1;                                                      \
    for (usize m = 0; m < nmajor; ++m)                                        \
        for (usize p = ptr[m]; p < ptr[m + 1]; ++p) vec[ind[p]] += 1;         \
    usize sum = 0;                                                            \
    optr[0] = 0;                                                              \
    for (usize j = 0; j < nminor; ++j) { sum += vec[j]; optr[j + 1] = sum; }  \
    for (usize j = 0; j < nminor; ++j) vec[j] = optr[j];                      \
    for (usize m = 0; m < nmajor; ++m)                                        \
        for (usize p = ptr[m]; p < ptr[m + 1]; ++p) {                         \
            usize j = ind[p];                                                 \
            usize q = vec[j];                                                 \
            oind[q] = m;                                                      \
            oval[q] = val[p];                                                 \
            vec[j] = q + 1;                                                   \
        }                                                                     \
    free(vec);                                                                \
    return 0;                                                                 \
}
DEF_TRANSPOSE(orc_transpose_f64, double)
DEF_TRANSPOSE(orc_transpose_f32, float)

/* ------------------------------------------------------------------------
 * COO -> CSR assembly.
 * Restates `impl From<&CooMatrix<T>> for CsrMatrix<T>`
 * src/csr/conv/coo.rs:4-115 phase by phase.
 * `From<&CooMatrix<T>> for CscMatrix<T>` src/csc/conv/coo.rs:4-115 is the
 * same code with the roles of row and column exchanged, so this function
 * is written on (major, minor): for CSR pass major = rows, minor = cols;
 * for CSC pass major = cols, minor = rows.
 *
 * in : len triplets (maj[i], min[i], val[i]) in insertion order
 * out: optr[nmajor+1]; oind/oval must have room for `len` entries;
 *      *onnz = number of stored entries after duplicate sum + zero drop.
 * ---------------------------------------------------------------------- */
#define NONE ((usize)-1)
#define DEF_FROM_COO(NAME, T)                                                 \
int NAME(usize nmajor, usize nminor, usize len, const usize *maj,             \
         const usize *min, const T *val, usize *optr, usize *oind, T *oval,   \
         usize *onnz)                                                         \
{                                                                             \
    int rc = -1;                                                              \
    usize *cnt = NULL, *mptr = NULL, *majind = NULL, *slot = NULL,            \
          *cur = NULL;                                                        \
    T *mval = NULL;                                                           \
    cnt = (usize *)calloc(nminor ? nminor : 1, sizeof(usize));                \
    mptr = (usize *)malloc((nminor + 1) * sizeof(usize));                     \
    majind = (usize *)malloc((len ? len : 1) * sizeof(usize));                \
    mval = (T *)malloc((len ? len : 1) * sizeof(T));                          \
    slot = (usize *)malloc((nmajor ? nmajor : 1) * sizeof(usize));            \
    cur = (usize *)malloc(((nmajor > nminor ? nmajor : nminor) + 1)           \
                          * sizeof(usize));                                   \
    if (!cnt || !mptr || !majind || !mval || !slot || !cur) goto done;        \
    /* P1 count entries per minor index            (coo.rs:10-13) */          \
    for (usize i = 0; i < len; ++i) cnt[min[i]] += 1;                         \
    /* P2 prefix sum -> minor pointers             (coo.rs:16-22) */          \
    {                                                                         \
        usize sum = 0;                                                        \
        mptr[0] = 0;                                                          \
        for (usize j = 0; j < nminor; ++j) { sum += cnt[j]; mptr[j+1] = sum; }\
    }                                                                         \
    /* P3 stable scatter into minor buckets        (coo.rs:25-33) */          \
    for (usize j = 0; j < nminor; ++j) cur[j] = mptr[j];                      \
    for (usize i = 0; i < len; ++i) {                                         \
        usize q = cur[min[i]];                                                \
        majind[q] = maj[i];                                                   \
        mval[q] = val[i];                                                     \
        cur[min[i]] = q + 1;                                                  \
    }                                                                         \
    /* P4 sum duplicates in insertion order        (coo.rs:36-57)             \
     * slot[] is the reference's Vec<Option<usize>>; a hit only counts when   \
     * it lies inside the current bucket (prev >= start).                  */ \
    for (usize m = 0; m < nmajor; ++m) slot[m] = NONE;                        \
    {                                                                         \
        usize nz = 0;                                                         \
        for (usize j = 0; j < nminor; ++j) {                                  \
            usize start = nz;                                                 \
            for (usize p = mptr[j]; p < mptr[j + 1]; ++p) {                   \
                usize m = majind[p];                                          \
                usize prev = slot[m];                                         \
                if (prev != NONE && prev >= start) {                          \
                    T v = mval[p];                                            \
                    mval[prev] += v;                                          \
                } else {                                                      \
                    slot[m] = nz;                                             \
                    majind[nz] = m;                                           \
                    mval[nz] = mval[p];                                       \
                    nz += 1;                                                  \
                }                                                             \
            }                                                                 \
            mptr[j] = start;                                                  \
        }                                                                     \
        mptr[nminor] = nz;                                                    \
    }                                                                         \
    /* P5 drop entries that compare equal to zero  (coo.rs:60-73)             \
     * (`!= T::zero()`: -0.0 is dropped, NaN is kept).  :63 reads the OLD     \
     * mptr[j+1] because only mptr[j] has been replaced so far.            */ \
    {                                                                         \
        usize nz = 0;                                                         \
        for (usize j = 0; j < nminor; ++j) {                                  \
            usize start = mptr[j];                                            \
            mptr[j] = nz;                                                     \
            for (usize p = start; p < mptr[j + 1]; ++p) {                     \
                if (mval[p] != (T)0) {                                        \
                    majind[nz] = majind[p];                                   \
                    mval[nz] = mval[p];                                       \
                    nz += 1;                                                  \
                }                                                             \
            }                                                                 \
        }                                                                     \
        mptr[nminor] = nz;                                                    \
        *onnz = nz;                                                           \
    }                                                                         \
    /* P6 count per major index, prefix sum        (coo.rs:76-91) */          \
    for (usize m = 0; m < nmajor; ++m) cur[m] = 0;                            \
    for (usize j = 0; j < nminor; ++j)                                        \
        for (usize p = mptr[j]; p < mptr[j + 1]; ++p) cur[majind[p]] += 1;    \
    {                                                                         \
        usize sum = 0;                                                        \
        optr[0] = 0;                                                          \
        for (usize m = 0; m < nmajor; ++m) { sum += cur[m]; optr[m+1] = sum; }\
    }                                                                         \
    /* P7 stable scatter minor-major -> major buckets (coo.rs:94-105) */      \
    for (usize m = 0; m < nmajor; ++m) cur[m] = optr[m];                      \
    for (usize j = 0; j < nminor; ++j)                                        \
        for (usize p = mptr[j]; p < mptr[j + 1]; ++p) {                       \
            usize m = majind[p];                                              \
            usize q = cur[m];                                                 \
            oind[q] = j;                                                      \
            oval[q] = mval[p];                                                \
            cur[m] = q + 1;                                                   \
        }                                                                     \
    rc = 0;                                                                   \
done:                                                                         \
    free(cnt); free(mptr); free(majind); free(mval); free(slot); free(cur);   \
    return rc;                                                                \
}
DEF_FROM_COO(orc_from_coo_f64, double)
DEF_FROM_COO(orc_from_coo_f32, float)

/* ------------------------------------------------------------------------
 * The reference's literal route to a product: sparse x sparse `Mul`.
 * Restates `impl Mul for &CscMatrix<T>` src/csc/ops/mul.rs:8-60: transpose
 * both inputs (:12), Gustavson accumulation with the `set`/`vec`
 * workspaces (:26-46: outer loop over output row i, then k ascending, then
 * the entries j of B's row k; first touch assigns :34, later ones `+=`
 * :37), transpose the result (:59).
 * `impl Mul for &CsrMatrix<T>` src/csr/ops/mul.rs:8-59 is the same loop
 * nest on the other axis: CSR `A * B` == this function called with
 * (lhs = B's arrays, rhs = A's arrays), the output arrays read as CSR
 * (because (A B)^T = B^T A^T and a CSR matrix's arrays are its transpose's
 * CSC arrays; the two scalar factors swap sides, which IEEE multiplication
 * does not notice).
 * Used (a) to pin the restatement against the reference's only product
 * known-answer test (G5, src/csc/ops/mul.rs:67-95) and (b) to show that
 * the direct SpMV above gives the same numbers as the reference's own route
 * when x is passed as an n x 1 sparse matrix.
 *
 * Operands are CSC: lhs = A (l_minor = nrows_A, l_major = ncols_A),
 * rhs = B (r_minor = nrows_B, r_major = ncols_B); needs l_major == r_minor
 * (mul.rs:9).  Output = CSC of A*B in malloc'ed arrays (free: orc_free).
 * Returns 0, -1 allocation failure, -2 dimension mismatch (reference
 * panics).  The j-indexed workspace is sized by its true extent (ncols_B);
 * the reference sizes it by nrows_A, which changes which
 * inputs panic, not the arithmetic.
 * ---------------------------------------------------------------------- */
#define DEF_SPGEMM(NAME, T, TRANSPOSE)                                        \
int NAME(usize l_minor, usize l_major, const usize *lptr, const usize *lind,  \
         const T *lval, usize r_minor, usize r_major, const usize *rptr,      \
         const usize *rind, const T *rval, usize **optr_out,                  \
         usize **oind_out, T **oval_out)                                      \
{                                                                             \
    if (l_major != r_minor) return -2;                                        \
    usize lnnz = lptr[l_major], rnnz = rptr[r_major];                         \
    int rc = -1;                                                              \
    usize cap = 16, nz = 0;                                                   \
    /* B by rows k  (= rhs.transpose() in the reference, its new `lhs`) */    \
    usize *bkp = (usize *)malloc((r_minor + 1) * sizeof(usize));              \
    usize *bki = (usize *)malloc((rnnz ? rnnz : 1) * sizeof(usize));          \
    T *bkv = (T *)malloc((rnnz ? rnnz : 1) * sizeof(T));                      \
    /* A by rows i  (= self.transpose(), its new `rhs`) */                    \
    usize *aip = (usize *)malloc((l_minor + 1) * sizeof(usize));              \
    usize *aii = (usize *)malloc((lnnz ? lnnz : 1) * sizeof(usize));          \
    T *aiv = (T *)malloc((lnnz ? lnnz : 1) * sizeof(T));                      \
    usize *set = (usize *)calloc(r_major ? r_major : 1, sizeof(usize));       \
    T *vec = (T *)calloc(r_major ? r_major : 1, sizeof(T));                   \
    usize *wptr = (usize *)malloc((l_minor + 1) * sizeof(usize));             \
    usize *wind = (usize *)malloc(cap * sizeof(usize));                       \
    T *wval = (T *)malloc(cap * sizeof(T));                                   \
    usize *optr = NULL, *oind = NULL; T *oval = NULL;                         \
    if (!bkp||!bki||!bkv||!aip||!aii||!aiv||!set||!vec||!wptr||!wind||!wval)  \
        goto done;                                                            \
    if (TRANSPOSE(r_major, r_minor, rptr, rind, rval, bkp, bki, bkv))         \
        goto done;                                                            \
    if (TRANSPOSE(l_major, l_minor, lptr, lind, lval, aip, aii, aiv))         \
        goto done;                                                            \
    for (usize i = 0; i < l_minor; ++i) {             /* mul.rs:26 */         \
        wptr[i] = nz;                                                         \
        for (usize p = aip[i]; p < aip[i + 1]; ++p) { /* :28, k ascending */  \
            usize k = aii[p];                                                 \
            for (usize q = bkp[k]; q < bkp[k + 1]; ++q) { /* :30 */           \
                usize j = bki[q];                                             \
                if (set[j] < i + 1) {                 /* :32-36 */            \
                    set[j] = i + 1;                                           \
                    if (nz == cap) {                                          \
                        usize ncap = cap * 2;                                 \
                        usize *ni = (usize *)realloc(wind, ncap * sizeof(usize)); \
                        if (!ni) goto done;                                   \
                        wind = ni;                                            \
                        T *nv = (T *)realloc(wval, ncap * sizeof(T));         \
                        if (!nv) goto done;                                   \
                        wval = nv;                                            \
                        cap = ncap;                                           \
                    }                                                         \
                    wind[nz] = j;                                             \
                    vec[j] = aiv[p] * bkv[q];                                 \
                    nz += 1;                                                  \
                } else {                              /* :37 */               \
                    vec[j] += aiv[p] * bkv[q];                                \
                }                                                             \
            }                                                                 \
        }                                                                     \
        for (usize p = wptr[i]; p < nz; ++p) wval[p] = vec[wind[p]]; /* :41-44 */ \
    }                                                                         \
    wptr[l_minor] = nz;                                                       \
    optr = (usize *)malloc((r_major + 1) * sizeof(usize));                    \
    oind = (usize *)malloc((nz ? nz : 1) * sizeof(usize));                    \
    oval = (T *)malloc((nz ? nz : 1) * sizeof(T));                            \
    if (!optr || !oind || !oval) goto done;                                   \
    if (TRANSPOSE(l_minor, r_major, wptr, wind, wval, optr, oind, oval))      \
        goto done;                                    /* :59 */               \
    *optr_out = optr; *oind_out = oind; *oval_out = oval;                     \
    optr = NULL; oind = NULL; oval = NULL;                                    \
    rc = 0;                                                                   \
done:                                                                         \
    free(bkp); free(bki); free(bkv); free(aip); free(aii); free(aiv);         \
    free(set); free(vec); free(wptr); free(wind); free(wval);                 \
    free(optr); free(oind); free(oval);                                       \
    return rc;                                                                \
}
DEF_SPGEMM(orc_spgemm_f64, double, orc_transpose_f64)
DEF_SPGEMM(orc_spgemm_f32, float, orc_transpose_f32)

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------------
 * Error bound helper for the parity criterion of SURVEY.md section 8d:
 * bound[i] = sum_k |A_ik| |x_k|  (componentwise scale of a row's sum).
 * ---------------------------------------------------------------------- */
void orc_csr_abs_bound_f64(usize nrows, const usize *rowptr,
                           const usize *colind, const double *values,
                           const double *x, double *bound)
{
    for (usize row = 0; row < nrows; ++row) {
        double acc = 0.0;
        for (usize p = rowptr[row]; p < rowptr[row + 1]; ++p) {
            double a = values[p] < 0 ? -values[p] : values[p];
            double b = x[colind[p]] < 0 ? -x[colind[p]] : x[colind[p]];
            acc += a * b;
        }
        bound[row] = acc;
    }
}
