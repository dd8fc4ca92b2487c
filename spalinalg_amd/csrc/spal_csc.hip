// spal_csc.hip -- CSC handle and y = A*x by atomic scatter.
//
// Contract (reference src/csc/ops/mul.rs:26-46): for
// every stored entry (i, k): y[i] += values[p] * x[k]; rows never touched
// give 0.0.  On the GPU the adds into one y[i] arrive in no fixed order
// (hardware f64/f32 atomics), so parity with the sequential CPU order is to
// rounding (<= 1e-10 relative, tested), not bitwise.
#include "csr_kernels.hpp"
#include "spal_internal.hpp"

namespace spal {

// L lanes share a column: x[k] is read once per lane group, rowind/values are
// streamed coalesced, every product goes out as one no-return atomic add.
template <typename T, int L>
__global__ __launch_bounds__(256) void csc_spmv_scatter(
    const uint32_t *__restrict__ colptr, const uint32_t *__restrict__ rowind,
    const T *__restrict__ vals, const T *__restrict__ x, T *__restrict__ y, uint32_t ncols) {
    constexpr uint32_t G = 256 / L;  // columns per workgroup pass
    const uint32_t g = threadIdx.x / L, s = threadIdx.x % L;
    for (uint32_t k = blockIdx.x * G + g; k < ncols; k += gridDim.x * G) {
        const uint32_t p0 = colptr[k], p1 = colptr[k + 1];
        if (p0 == p1) continue;
        const T xk = x[k];
        for (uint32_t p = p0 + s; p < p1; p += L) {
            const uint32_t i = load_stream(rowind + p);
            const T v = load_stream(vals + p);
            atomicAdd(&y[i], v * xk);  // -munsafe-fp-atomics: global_atomic_add_f64 / _f32
        }
    }
}

static int pick_lanes_csc(double mean) {
    int L = 2;
    while (L < 64 && (double)L < mean) L <<= 1;
    return L;
}

template <typename T>
static hipError_t csc_launch_t(const spal_csc *a, const void *x, void *y, hipStream_t st) {
    hipError_t e = hipMemsetAsync(y, 0, a->nrows * sizeof(T), st);
    if (e != hipSuccess || a->nnz == 0) return e;
    const int L = a->lanes_per_col;
    const uint32_t G = 256 / L;
    const uint64_t want = (a->ncols + G - 1) / G;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(want, 256ull * 8 * 4);
#define SPAL_CSC_CASE(LL)                                                                         \
    case LL:                                                                                      \
        hipLaunchKernelGGL((csc_spmv_scatter<T, LL>), dim3(grid), dim3(256), 0, st, a->d_colptr, \
                           a->d_rowind, (const T *)a->d_values, (const T *)x, (T *)y,            \
                           (uint32_t)a->ncols);                                                   \
        break;
    switch (L) {
        SPAL_CSC_CASE(2) SPAL_CSC_CASE(4) SPAL_CSC_CASE(8) SPAL_CSC_CASE(16) SPAL_CSC_CASE(32)
        SPAL_CSC_CASE(64)
        default: return hipErrorInvalidValue;
    }
#undef SPAL_CSC_CASE
    return hipGetLastError();
}

static int csc_launch(spal_csc *a, const void *x, void *y, hipStream_t st) {
    hipError_t e = a->elem_size == 8 ? csc_launch_t<double>(a, x, y, st)
                                     : csc_launch_t<float>(a, x, y, st);
    if (e != hipSuccess) return fail(SPAL_ERR_HIP, "csc spmv launch failed: %s", hipGetErrorString(e));
    return SPAL_OK;
}

static void csc_free(spal_csc *a) {
    if (!a) return;
    (void)hipFree(a->d_colptr);
    (void)hipFree(a->d_rowind);
    (void)hipFree(a->d_values);
    (void)hipFree(a->d_x);
    (void)hipFree(a->d_y);
    if (a->stream) (void)hipStreamDestroy(a->stream);
    delete a;
}

template <typename T>
static int csc_create(int device, uint64_t nrows, uint64_t ncols, const uint64_t *colptr,
                      uint64_t colptr_len, const uint64_t *rowind, uint64_t rowind_len,
                      const T *values, uint64_t values_len, spal_csc_t *out) {
    if (!out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_create: out is NULL");
    *out = nullptr;
    if (!colptr || (!rowind && rowind_len) || (!values && values_len))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_create: null array");
    int reason = 0;
    SPAL_TRY(spal_csc_validate(nrows, ncols, colptr, colptr_len, rowind, rowind_len, values_len, &reason));
    const uint64_t nnz = colptr[ncols];
    if (ncols >= 0xffffffffull || nrows > 0xffffffffull || nnz > 0xffffffffull)
        return fail(SPAL_ERR_UNSUPPORTED,
                    "shape %llu x %llu with %llu entries does not fit 32-bit device indices",
                    (unsigned long long)nrows, (unsigned long long)ncols, (unsigned long long)nnz);
    DeviceGuard guard(device);
    if (guard.status != SPAL_OK) return guard.status;
    std::vector<uint32_t> cp32(ncols + 1), ri32(nnz);
    parallel_for(ncols + 1, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b; i < e; ++i) cp32[i] = (uint32_t)colptr[i];
    });
    parallel_for(nnz, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b; i < e; ++i) ri32[i] = (uint32_t)rowind[i];
    });
    spal_csc *a = new spal_csc;
    a->device = device;
    a->elem_size = (int)sizeof(T);
    a->nrows = nrows; a->ncols = ncols; a->nnz = nnz;
    a->kernel = 1;
    a->lanes_per_col = pick_lanes_csc(ncols ? (double)nnz / (double)ncols : 0.0);
    hipError_t e = hipMalloc(&a->d_colptr, (ncols + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&a->d_rowind, std::max<uint64_t>(nnz, 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&a->d_values, std::max<uint64_t>(nnz, 1) * sizeof(T));
    if (e == hipSuccess) e = hipMemcpy(a->d_colptr, cp32.data(), (ncols + 1) * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) e = hipMemcpy(a->d_rowind, ri32.data(), nnz * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) e = hipMemcpy(a->d_values, values, nnz * sizeof(T), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        csc_free(a);
        return fail(e == hipErrorOutOfMemory ? SPAL_ERR_OUT_OF_MEMORY : SPAL_ERR_HIP,
                    "spal_csc_create: upload failed: %s", hipGetErrorString(e));
    }
    *out = a;
    return SPAL_OK;
}

template <typename T>
static int csc_spmv_host(spal_csc_t a, const T *x, uint64_t x_len, T *y, uint64_t y_len) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_spmv: handle is NULL");
    if (a->elem_size != (int)sizeof(T))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_spmv: handle holds %s values",
                    a->elem_size == 8 ? "f64" : "f32");
    if (x_len != a->ncols)
        return fail(SPAL_ERR_INVALID_ARGUMENT,
                    "dimension mismatch: x.len() = %llu but ncols = %llu (assert_eq!, csc/ops/mul.rs:9)",
                    (unsigned long long)x_len, (unsigned long long)a->ncols);
    if (y_len != a->nrows)
        return fail(SPAL_ERR_INVALID_ARGUMENT, "y.len() = %llu but nrows = %llu",
                    (unsigned long long)y_len, (unsigned long long)a->nrows);
    if (!x || !y) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_spmv: null vector");
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    std::lock_guard<std::mutex> lock(a->mu);
    if (!a->d_x) SPAL_HIP_TRY(hipMalloc(&a->d_x, a->ncols * sizeof(T)));
    if (!a->d_y) SPAL_HIP_TRY(hipMalloc(&a->d_y, a->nrows * sizeof(T)));
    SPAL_HIP_TRY(hipMemcpyAsync(a->d_x, x, a->ncols * sizeof(T), hipMemcpyHostToDevice, a->stream));
    SPAL_TRY(csc_launch(a, a->d_x, a->d_y, a->stream));
    SPAL_HIP_TRY(hipMemcpyAsync(y, a->d_y, a->nrows * sizeof(T), hipMemcpyDeviceToHost, a->stream));
    SPAL_HIP_TRY(hipStreamSynchronize(a->stream));
    return SPAL_OK;
}

template <typename T>
static int csc_spmv_dev(spal_csc_t a, const T *x_dev, T *y_dev, void *stream) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_spmv_dev: handle is NULL");
    if (a->elem_size != (int)sizeof(T))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_spmv_dev: handle holds %s values",
                    a->elem_size == 8 ? "f64" : "f32");
    if (!x_dev || !y_dev) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_spmv_dev: null vector");
    DeviceGuard guard(a->device);
    if (guard.status != SPAL_OK) return guard.status;
    return csc_launch(a, x_dev, y_dev, (hipStream_t)stream);
}

}  // namespace spal

using namespace spal;

extern "C" {

int spal_csc_create_f64(int device, uint64_t nrows, uint64_t ncols, const uint64_t *colptr,
                        uint64_t colptr_len, const uint64_t *rowind, uint64_t rowind_len,
                        const double *values, uint64_t values_len, spal_csc_t *out) {
    return csc_create<double>(device, nrows, ncols, colptr, colptr_len, rowind, rowind_len, values,
                              values_len, out);
}
int spal_csc_create_f32(int device, uint64_t nrows, uint64_t ncols, const uint64_t *colptr,
                        uint64_t colptr_len, const uint64_t *rowind, uint64_t rowind_len,
                        const float *values, uint64_t values_len, spal_csc_t *out) {
    return csc_create<float>(device, nrows, ncols, colptr, colptr_len, rowind, rowind_len, values,
                             values_len, out);
}
int spal_csc_destroy(spal_csc_t a) {
    if (!a) return SPAL_OK;
    DeviceGuard guard(a->device);
    csc_free(a);
    return SPAL_OK;
}
int spal_csc_shape(spal_csc_t a, uint64_t *nrows, uint64_t *ncols, uint64_t *nnz, int *elem_size) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_shape: handle is NULL");
    if (nrows) *nrows = a->nrows;
    if (ncols) *ncols = a->ncols;
    if (nnz) *nnz = a->nnz;
    if (elem_size) *elem_size = a->elem_size;
    return SPAL_OK;
}
int spal_csc_spmv_f64(spal_csc_t a, const double *x, uint64_t x_len, double *y, uint64_t y_len) {
    return csc_spmv_host<double>(a, x, x_len, y, y_len);
}
int spal_csc_spmv_f32(spal_csc_t a, const float *x, uint64_t x_len, float *y, uint64_t y_len) {
    return csc_spmv_host<float>(a, x, x_len, y, y_len);
}
int spal_csc_spmv_dev_f64(spal_csc_t a, const double *x_dev, double *y_dev, void *stream) {
    return csc_spmv_dev<double>(a, x_dev, y_dev, stream);
}
int spal_csc_spmv_dev_f32(spal_csc_t a, const float *x_dev, float *y_dev, void *stream) {
    return csc_spmv_dev<float>(a, x_dev, y_dev, stream);
}
int spal_csc_set_option(spal_csc_t a, const char *key, int64_t value) {
    if (!a || !key) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_set_option: null argument");
    if (!strcmp(key, "lanes_per_col")) {
        if (value == 0) {
            a->lanes_per_col = pick_lanes_csc(a->ncols ? (double)a->nnz / (double)a->ncols : 0.0);
            return SPAL_OK;
        }
        if (value < 2 || value > 64 || (value & (value - 1)))
            return fail(SPAL_ERR_INVALID_ARGUMENT, "lanes_per_col must be one of 2,4,8,16,32,64");
        a->lanes_per_col = (int)value;
        return SPAL_OK;
    }
    return fail(SPAL_ERR_INVALID_ARGUMENT, "unknown option '%s'", key);
}
int spal_csc_describe(spal_csc_t a, char *buf, size_t buf_len) {
    if (!a || !buf || !buf_len) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_csc_describe: null argument");
    snprintf(buf, buf_len,
             "{\"format\": \"csc\", \"dtype\": \"%s\", \"nrows\": %llu, \"ncols\": %llu, \"nnz\": %llu, "
             "\"index_bits\": 32, \"kernel\": \"atomic_scatter\", \"lanes_per_col\": %d}",
             a->elem_size == 8 ? "f64" : "f32", (unsigned long long)a->nrows,
             (unsigned long long)a->ncols, (unsigned long long)a->nnz, a->lanes_per_col);
    return SPAL_OK;
}

}  // extern "C"
