// spal_mg.hip -- row-partitioned y = A*x over several GPUs of one node from ONE
// process (the `spal_mg_*` exports of section 8b): what a
// single-process host such as the Rust crate binds.  (bench.py uses the other
// arrangement -- one process per GPU over torch.distributed -- and the same
// kernels; see spalinalg_amd/dist.py.)
//
//   partition : contiguous row ranges with balanced stored entries
//               (spal_partition_rows); every GPU holds its rows' CSR arrays and
//               a full-length x.
//   exchange  : ncclBroadcast of x from GPU 0 (once per x), then per product the
//               local kernel on every GPU and an ncclAllGather of the y slices
//               (padded to the longest slice), so every GPU ends with all of y.
// RCCL (xGMI) is loaded lazily with dlopen: libspal_hip.so has no link-time
// dependency on it, and single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "spal_internal.hpp"

namespace spal {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static int rccl_load(Rccl **out) {
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!r.lib) {
        // an already loaded librccl.so.1 (e.g. torch's) is reused by the loader
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.lib) return fail(SPAL_ERR_HIP, "cannot load RCCL (librccl.so.1): %s", dlerror());
#define SPAL_RCCL_SYM(field, sym)                                                        \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, #sym));                   \
    if (!r.field) { r.lib = nullptr; return fail(SPAL_ERR_HIP, "RCCL lacks %s", #sym); }
        SPAL_RCCL_SYM(CommInitAll, ncclCommInitAll)
        SPAL_RCCL_SYM(CommDestroy, ncclCommDestroy)
        SPAL_RCCL_SYM(GroupStart, ncclGroupStart)
        SPAL_RCCL_SYM(GroupEnd, ncclGroupEnd)
        SPAL_RCCL_SYM(Broadcast, ncclBroadcast)
        SPAL_RCCL_SYM(AllGather, ncclAllGather)
        SPAL_RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef SPAL_RCCL_SYM
    }
    *out = &r;
    return SPAL_OK;
}

#define SPAL_NCCL_TRY(ctx, expr)                                                            \
    do {                                                                                    \
        ncclResult_t r_ = (expr);                                                           \
        if (r_ != ncclSuccess)                                                              \
            return ::spal::fail(SPAL_ERR_HIP, "%s failed: %s", #expr, (ctx)->GetErrorString(r_)); \
    } while (0)

}  // namespace spal

using namespace spal;

struct spal_mg {
    int ngpus = 0;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    Rccl *rccl = nullptr;
};

struct spal_mg_csr {
    spal_mg *ctx = nullptr;
    int elem_size = 8;
    uint64_t nrows = 0, ncols = 0, nnz = 0;
    std::vector<uint64_t> bounds;     // ngpus + 1 row boundaries
    uint64_t max_rows = 0;            // longest slice (all-gather padding)
    std::vector<spal_csr_t> shard;    // one CSR handle per GPU
    std::vector<void *> d_x;          // ncols per GPU
    std::vector<void *> d_yloc;       // max_rows per GPU (this GPU's slice, padded)
    std::vector<void *> d_yall;       // ngpus * max_rows per GPU (gathered, padded)
    std::mutex mu;
};

static void mg_csr_free(spal_mg_csr *a) {
    if (!a) return;
    for (size_t g = 0; g < a->shard.size(); ++g) {
        if (a->shard[g]) spal_csr_destroy(a->shard[g]);
        if (a->ctx && g < (size_t)a->ctx->ngpus) {
            (void)hipSetDevice(a->ctx->devices[g]);
            if (g < a->d_x.size()) (void)dev_free(a->d_x[g]);
            if (g < a->d_yloc.size()) (void)dev_free(a->d_yloc[g]);
            if (g < a->d_yall.size()) (void)dev_free(a->d_yall[g]);
        }
    }
    delete a;
}

template <typename T>
static int mg_csr_create(spal_mg *ctx, uint64_t nrows, uint64_t ncols, const uint64_t *rowptr,
                         uint64_t rowptr_len, const uint64_t *colind, uint64_t colind_len,
                         const T *values, uint64_t values_len, spal_mg_csr **out) {
    if (!ctx || !out) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_create: null argument");
    *out = nullptr;
    int reason = 0;
    SPAL_TRY(spal_csr_validate(nrows, ncols, rowptr, rowptr_len, colind, colind_len, values_len, &reason));
    spal_mg_csr *a = new spal_mg_csr;
    a->ctx = ctx;
    a->elem_size = (int)sizeof(T);
    a->nrows = nrows; a->ncols = ncols; a->nnz = rowptr[nrows];
    const int G = ctx->ngpus;
    a->bounds.resize(G + 1);
    int st = spal_partition_rows(rowptr, nrows, (uint32_t)G, a->bounds.data());
    if (st != SPAL_OK) { delete a; return st; }
    // a GPU must own at least one row (CsrMatrix::new requires nrows > 0): when
    // there are fewer rows than GPUs, or the balance left a range empty, fall back
    // to an even row split
    bool empty = false;
    for (int g = 0; g < G; ++g) empty |= a->bounds[g] == a->bounds[g + 1];
    if (empty) {
        if (nrows < (uint64_t)G) { delete a; return fail(SPAL_ERR_INVALID_ARGUMENT, "fewer rows (%llu) than GPUs (%d)", (unsigned long long)nrows, G); }
        for (int g = 0; g <= G; ++g) a->bounds[g] = nrows * (uint64_t)g / (uint64_t)G;
    }
    for (int g = 0; g < G; ++g) a->max_rows = std::max(a->max_rows, a->bounds[g + 1] - a->bounds[g]);
    a->shard.assign(G, nullptr);
    a->d_x.assign(G, nullptr); a->d_yloc.assign(G, nullptr); a->d_yall.assign(G, nullptr);
    std::vector<uint64_t> rp;
    for (int g = 0; g < G; ++g) {
        const uint64_t r0 = a->bounds[g], r1 = a->bounds[g + 1];
        const uint64_t e0 = rowptr[r0], e1 = rowptr[r1];
        rp.resize(r1 - r0 + 1);
        for (uint64_t r = r0; r <= r1; ++r) rp[r - r0] = rowptr[r] - e0;  // a row range is a CsrMatrix of its own
        st = a->elem_size == 8
                 ? spal_csr_create_f64(ctx->devices[g], r1 - r0, ncols, rp.data(), rp.size(), colind + e0,
                                       e1 - e0, (const double *)values + e0, e1 - e0, &a->shard[g])
                 : spal_csr_create_f32(ctx->devices[g], r1 - r0, ncols, rp.data(), rp.size(), colind + e0,
                                       e1 - e0, (const float *)values + e0, e1 - e0, &a->shard[g]);
        if (st != SPAL_OK) { mg_csr_free(a); return st; }
        hipError_t e = hipSetDevice(ctx->devices[g]);
        if (e == hipSuccess) e = dev_alloc((void **)&a->d_x[g], ncols * sizeof(T));
        if (e == hipSuccess) e = dev_alloc((void **)&a->d_yloc[g], a->max_rows * sizeof(T));
        if (e == hipSuccess) e = dev_alloc((void **)&a->d_yall[g], (size_t)G * a->max_rows * sizeof(T));
        if (e == hipSuccess) e = hipMemset(a->d_yloc[g], 0, a->max_rows * sizeof(T));
        if (e != hipSuccess) {
            mg_csr_free(a);
            return fail(e == hipErrorOutOfMemory ? SPAL_ERR_OUT_OF_MEMORY : SPAL_ERR_HIP,
                        "spal_mg_csr_create: %s", hipGetErrorString(e));
        }
    }
    *out = a;
    return SPAL_OK;
}

// x is on GPU 0's buffer: broadcast, multiply, all-gather; asynchronous on the
// context's per-GPU streams.
static int mg_broadcast_x(spal_mg_csr *a) {
    spal_mg *c = a->ctx;
    if (c->ngpus == 1) return SPAL_OK;
    const ncclDataType_t dt = a->elem_size == 8 ? ncclFloat64 : ncclFloat32;
    SPAL_NCCL_TRY(c->rccl, c->rccl->GroupStart());
    for (int g = 0; g < c->ngpus; ++g)
        SPAL_NCCL_TRY(c->rccl, c->rccl->Broadcast(a->d_x[g], a->d_x[g], a->ncols, dt, 0, c->comms[g], c->streams[g]));
    SPAL_NCCL_TRY(c->rccl, c->rccl->GroupEnd());
    return SPAL_OK;
}

static int mg_multiply_gather(spal_mg_csr *a) {
    spal_mg *c = a->ctx;
    for (int g = 0; g < c->ngpus; ++g) {
        SPAL_HIP_TRY(hipSetDevice(c->devices[g]));
        SPAL_TRY(a->elem_size == 8
                     ? spal_csr_spmv_dev_f64(a->shard[g], (const double *)a->d_x[g], (double *)a->d_yloc[g], c->streams[g])
                     : spal_csr_spmv_dev_f32(a->shard[g], (const float *)a->d_x[g], (float *)a->d_yloc[g], c->streams[g]));
    }
    const ncclDataType_t dt = a->elem_size == 8 ? ncclFloat64 : ncclFloat32;
    if (c->ngpus == 1) {
        SPAL_HIP_TRY(hipMemcpyAsync(a->d_yall[0], a->d_yloc[0], a->max_rows * (size_t)a->elem_size,
                                    hipMemcpyDeviceToDevice, c->streams[0]));
        return SPAL_OK;
    }
    SPAL_NCCL_TRY(c->rccl, c->rccl->GroupStart());
    for (int g = 0; g < c->ngpus; ++g)
        SPAL_NCCL_TRY(c->rccl, c->rccl->AllGather(a->d_yloc[g], a->d_yall[g], a->max_rows, dt, c->comms[g], c->streams[g]));
    SPAL_NCCL_TRY(c->rccl, c->rccl->GroupEnd());
    return SPAL_OK;
}

static int mg_sync(spal_mg *c) {
    for (int g = 0; g < c->ngpus; ++g) {
        SPAL_HIP_TRY(hipSetDevice(c->devices[g]));
        SPAL_HIP_TRY(hipStreamSynchronize(c->streams[g]));
    }
    return SPAL_OK;
}

template <typename T>
static int mg_csr_spmv(spal_mg_csr *a, const T *x, uint64_t x_len, T *y, uint64_t y_len) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_spmv: handle is NULL");
    if (a->elem_size != (int)sizeof(T))
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_spmv: handle holds %s values", a->elem_size == 8 ? "f64" : "f32");
    if (x_len != a->ncols)
        return fail(SPAL_ERR_INVALID_ARGUMENT,
                    "dimension mismatch: x.len() = %llu but ncols = %llu (assert_eq!, csr/ops/mul.rs:9)",
                    (unsigned long long)x_len, (unsigned long long)a->ncols);
    if (y_len != a->nrows) return fail(SPAL_ERR_INVALID_ARGUMENT, "y.len() != nrows");
    if (!x || !y) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_spmv: null vector");
    std::lock_guard<std::mutex> lock(a->mu);
    spal_mg *c = a->ctx;
    int prev = -1;
    (void)hipGetDevice(&prev);
    int st = SPAL_OK;
    do {
        hipError_t e = hipSetDevice(c->devices[0]);
        if (e == hipSuccess) e = hipMemcpyAsync(a->d_x[0], x, a->ncols * sizeof(T), hipMemcpyHostToDevice, c->streams[0]);
        if (e != hipSuccess) { st = fail(SPAL_ERR_HIP, "spal_mg_csr_spmv: %s", hipGetErrorString(e)); break; }
        if ((st = mg_broadcast_x(a)) != SPAL_OK) break;
        if ((st = mg_multiply_gather(a)) != SPAL_OK) break;
        // GPU 0 holds every slice (padded to max_rows): copy them out back to back
        e = hipSetDevice(c->devices[0]);
        for (int g = 0; g < c->ngpus && e == hipSuccess; ++g) {
            const uint64_t r0 = a->bounds[g], n = a->bounds[g + 1] - r0;
            e = hipMemcpyAsync(y + r0, (const T *)a->d_yall[0] + (size_t)g * a->max_rows, n * sizeof(T),
                               hipMemcpyDeviceToHost, c->streams[0]);
        }
        if (e != hipSuccess) { st = fail(SPAL_ERR_HIP, "spal_mg_csr_spmv: %s", hipGetErrorString(e)); break; }
        st = mg_sync(c);
    } while (0);
    if (prev >= 0) (void)hipSetDevice(prev);
    return st;
}

extern "C" {

int spal_mg_create(int ngpus, const int *devices, spal_mg_t *out) {
    if (!out || ngpus < 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_create: need ngpus >= 1 and out");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        (void)hipGetLastError();
        return fail(SPAL_ERR_NO_DEVICE, "no HIP device available");
    }
    if (ngpus > count) return fail(SPAL_ERR_INVALID_ARGUMENT, "%d GPUs requested, %d visible", ngpus, count);
    spal_mg *c = new spal_mg;
    c->ngpus = ngpus;
    for (int g = 0; g < ngpus; ++g) {
        const int d = devices ? devices[g] : g;
        if (d < 0 || d >= count) { delete c; return fail(SPAL_ERR_INVALID_ARGUMENT, "device %d out of range", d); }
        c->devices.push_back(d);
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    c->streams.assign(ngpus, nullptr);
    for (int g = 0; g < ngpus; ++g) {
        hipError_t e = hipSetDevice(c->devices[g]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->streams[g], hipStreamNonBlocking);
        if (e != hipSuccess) { spal_mg_destroy(c); return fail(SPAL_ERR_HIP, "spal_mg_create: %s", hipGetErrorString(e)); }
    }
    if (ngpus > 1) {  // one communicator per GPU, all owned by this process
        int st = rccl_load(&c->rccl);
        if (st != SPAL_OK) { spal_mg_destroy(c); return st; }
        c->comms.assign(ngpus, nullptr);
        ncclResult_t r = c->rccl->CommInitAll(c->comms.data(), ngpus, c->devices.data());
        if (r != ncclSuccess) {
            c->comms.clear();
            const char *msg = c->rccl->GetErrorString(r);
            spal_mg_destroy(c);
            return fail(SPAL_ERR_HIP, "ncclCommInitAll failed: %s", msg);
        }
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    *out = c;
    return SPAL_OK;
}

int spal_mg_destroy(spal_mg_t c) {
    if (!c) return SPAL_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (size_t g = 0; g < c->comms.size(); ++g)
        if (c->comms[g] && c->rccl) (void)c->rccl->CommDestroy(c->comms[g]);
    for (size_t g = 0; g < c->streams.size(); ++g)
        if (c->streams[g]) { (void)hipSetDevice(c->devices[g]); (void)hipStreamDestroy(c->streams[g]); }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete c;
    return SPAL_OK;
}

int spal_mg_device_count(spal_mg_t c, int *ngpus) {
    if (!c || !ngpus) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_device_count: null argument");
    *ngpus = c->ngpus;
    return SPAL_OK;
}

int spal_mg_csr_create_f64(spal_mg_t ctx, uint64_t nrows, uint64_t ncols, const uint64_t *rowptr,
                           uint64_t rowptr_len, const uint64_t *colind, uint64_t colind_len,
                           const double *values, uint64_t values_len, spal_mg_csr_t *out) {
    return mg_csr_create<double>(ctx, nrows, ncols, rowptr, rowptr_len, colind, colind_len, values, values_len, out);
}
int spal_mg_csr_create_f32(spal_mg_t ctx, uint64_t nrows, uint64_t ncols, const uint64_t *rowptr,
                           uint64_t rowptr_len, const uint64_t *colind, uint64_t colind_len,
                           const float *values, uint64_t values_len, spal_mg_csr_t *out) {
    return mg_csr_create<float>(ctx, nrows, ncols, rowptr, rowptr_len, colind, colind_len, values, values_len, out);
}
int spal_mg_csr_destroy(spal_mg_csr_t a) {
    if (!a) return SPAL_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    mg_csr_free(a);
    if (prev >= 0) (void)hipSetDevice(prev);
    return SPAL_OK;
}
int spal_mg_csr_partition(spal_mg_csr_t a, uint64_t *bounds) {
    if (!a || !bounds) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_partition: null argument");
    for (size_t g = 0; g < a->bounds.size(); ++g) bounds[g] = a->bounds[g];
    return SPAL_OK;
}
int spal_mg_csr_spmv_f64(spal_mg_csr_t a, const double *x, uint64_t x_len, double *y, uint64_t y_len) {
    return mg_csr_spmv<double>(a, x, x_len, y, y_len);
}
int spal_mg_csr_spmv_f32(spal_mg_csr_t a, const float *x, uint64_t x_len, float *y, uint64_t y_len) {
    return mg_csr_spmv<float>(a, x, x_len, y, y_len);
}

// resident path: x already lives in GPU 0's buffer (spal_mg_csr_x_root)
int spal_mg_csr_x_root(spal_mg_csr_t a, void **x_dev) {
    if (!a || !x_dev) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_x_root: null argument");
    *x_dev = a->d_x[0];
    return SPAL_OK;
}
int spal_mg_csr_broadcast_x(spal_mg_csr_t a) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_broadcast_x: handle is NULL");
    std::lock_guard<std::mutex> lock(a->mu);
    return mg_broadcast_x(a);
}
int spal_mg_csr_spmv_resident(spal_mg_csr_t a) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_spmv_resident: handle is NULL");
    std::lock_guard<std::mutex> lock(a->mu);
    int prev = -1;
    (void)hipGetDevice(&prev);
    int st = mg_multiply_gather(a);
    if (prev >= 0) (void)hipSetDevice(prev);
    return st;
}
int spal_mg_csr_y_root(spal_mg_csr_t a, void **y_dev, uint64_t *slice_stride) {
    if (!a || !y_dev) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_y_root: null argument");
    *y_dev = a->d_yall[0];
    if (slice_stride) *slice_stride = a->max_rows;
    return SPAL_OK;
}
int spal_mg_csr_synchronize(spal_mg_csr_t a) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_synchronize: handle is NULL");
    int prev = -1;
    (void)hipGetDevice(&prev);
    int st = mg_sync(a->ctx);
    if (prev >= 0) (void)hipSetDevice(prev);
    return st;
}

}  // extern "C"
