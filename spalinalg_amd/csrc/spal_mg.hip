// spal_mg.hip -- row-partitioned y = A*x over several GPUs of one node from ONE
// process (SURVEY.md sections 8e / 8f-4, the `spal_mg_*` exports of section 8b): what a
// single-process host such as the Rust crate binds.  (bench.py's default host is the
// other arrangement -- one process per GPU over torch.distributed -- with the same
// kernels; see spalinalg_amd/dist.py.  `bench.py --host mg` drives this file.)
//
//   partition : contiguous row ranges with balanced stored entries
//               (spal_partition_rows); every GPU holds its rows' CSR arrays and a
//               full-length x buffer of which it only ever reads its WINDOW
//               [need_lo, need_hi): the columns its rows store.
//   x         : scatter_x -- GPU 0 sends every GPU its window only (a banded shard:
//               its own slice +- W/2, i.e. 1/N of the vector per xGMI link instead
//               of all of it); broadcast_x -- the whole vector to everyone
//               (ncclBroadcast; the general path north_star names).
//   product   : the single-GPU kernel of every shard on that GPU's stream; no
//               collective inside.
//   y         : gather_y -- the slices land back to back in GPU 0's y (unequal
//               slices, no padding); spmv_resident -- all-gather (every GPU ends with
//               all of y).
//   halo      : spmv_halo (square matrices, iterative use: y is the next x) -- per
//               step every GPU receives only the entries of y that its rows read as
//               columns and it does not own (two neighbour messages for a band).
//
// Transport.  "rccl": grouped ncclSend / ncclRecv (+ ncclBroadcast / ncclAllGather) on
// one communicator per GPU (ncclCommInitAll), over xGMI.  "copy": peer-to-peer
// hipMemcpyPeerAsync on the receiver's stream, ordered by events -- no RCCL at all.
// The copy transport also accepts a device list with REPEATS (several shards on one
// physical GPU), which is how the window / halo planner and the stream ordering are
// tested on a 1-GPU box; RCCL refuses duplicate devices.  RCCL is loaded lazily with
// dlopen: libspal_hip.so has no link-time dependency on it.
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
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
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
        SPAL_RCCL_SYM(Send, ncclSend)
        SPAL_RCCL_SYM(Recv, ncclRecv)
        SPAL_RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef SPAL_RCCL_SYM
    }
    *out = &r;
    return SPAL_OK;
}

}  // namespace spal

using namespace spal;

enum { kTransportRccl = 0, kTransportCopy = 1 };
enum { kPhaseX = 0, kPhaseCompute = 1, kPhaseHalo = 2, kPhaseY = 3, kPhases = 4 };

struct spal_mg {
    int ngpus = 0;
    int transport = kTransportRccl;
    bool repeats = false;             // a physical device appears more than once (copy transport only)
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    // copy transport: `ready[g]` = the data GPU g is about to be read from is complete on its stream;
    // `pulled[g]` = the copies GPU g issued on its own stream are complete
    std::vector<hipEvent_t> ready, pulled;
    Rccl *rccl = nullptr;
};

// one message of an exchange: n elements from offset src_off of GPU src's buffer to offset dst_off of GPU dst's
struct MgMsg {
    int src, dst;
    uint64_t src_off, dst_off, n;
};

struct spal_mg_csr {
    spal_mg *ctx = nullptr;
    int elem_size = 8;
    uint64_t nrows = 0, ncols = 0, nnz = 0;
    std::vector<uint64_t> bounds;     // ngpus + 1 row boundaries
    std::vector<uint64_t> need_lo, need_hi;   // per GPU: the columns its rows store lie in [need_lo, need_hi) (lo == hi: none)
    uint64_t max_rows = 0;            // longest slice (all-gather padding)
    std::vector<spal_csr_t> shard;    // one CSR handle per GPU
    std::vector<void *> d_x;          // ncols per GPU: the current x (GPU g reads its window of it)
    std::vector<void *> d_x2;         // halo mode: the other vector of the pair (allocated on first use)
    std::vector<void *> d_yloc;       // this GPU's slice of y (GPU 0: the front of d_yroot)
    void *d_yroot = nullptr;          // GPU 0: y, nrows elements, slices back to back (gather_y)
    std::vector<void *> d_ypad;       // all-gather path: this GPU's slice padded to max_rows (allocated on first use)
    std::vector<void *> d_yall;       // all-gather path: ngpus * max_rows per GPU
    std::vector<MgMsg> x_msgs;        // scatter_x: GPU 0 -> g, window of g
    std::vector<MgMsg> y_msgs;        // gather_y:  g -> GPU 0, slice of g
    std::vector<MgMsg> halo_msgs;     // spmv_halo: owner -> reader, the part of the reader's window the owner computes
    bool last_was_halo = false;       // the last product's slices live in d_x (halo step), not in d_yloc
    // event-timed phases of the last calls (per GPU, on its stream)
    std::vector<hipEvent_t> ev;       // [g][phase][begin / end]
    bool timed[kPhases] = {false, false, false, false};
    std::mutex mu;
};

#define SPAL_NCCL_OK(c, expr, st)                                                                   \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess && (st) == SPAL_OK)                                                   \
            (st) = ::spal::fail(SPAL_ERR_HIP, "%s failed: %s", #expr, (c)->rccl->GetErrorString(r_)); \
    } while (0)

static hipEvent_t &mg_ev(spal_mg_csr *a, int g, int phase, int end) { return a->ev[((size_t)g * kPhases + phase) * 2 + end]; }

static int mg_phase_mark(spal_mg_csr *a, int phase, int end) {
    spal_mg *c = a->ctx;
    for (int g = 0; g < c->ngpus; ++g) {
        DeviceGuard guard(c->devices[g]);
        if (guard.status != SPAL_OK) return guard.status;
        SPAL_HIP_TRY(hipEventRecord(mg_ev(a, g, phase, end), c->streams[g]));
    }
    if (end) a->timed[phase] = true;
    return SPAL_OK;
}

// Moves the messages: vector buffers `buf[g]` (same element offsets on every GPU), asynchronously on the
// context's streams.  RCCL: one group of ncclSend / ncclRecv.  Copy: the receiver pulls on its own stream
// once the sender's stream has reached this point, and the sender's stream then waits for its readers.
static int mg_exchange(spal_mg_csr *a, const std::vector<MgMsg> &msgs, const std::vector<void *> &buf) {
    spal_mg *c = a->ctx;
    if (msgs.empty()) return SPAL_OK;
    const size_t es = (size_t)a->elem_size;
    if (c->transport == kTransportRccl) {
        const ncclDataType_t dt = a->elem_size == 8 ? ncclFloat64 : ncclFloat32;
        int st = SPAL_OK;
        SPAL_NCCL_OK(c, c->rccl->GroupStart(), st);
        if (st != SPAL_OK) return st;
        for (const MgMsg &m : msgs) {
            if (st != SPAL_OK) break;
            SPAL_NCCL_OK(c, c->rccl->Send((const char *)buf[m.src] + m.src_off * es, m.n, dt, m.dst, c->comms[m.src], c->streams[m.src]), st);
            SPAL_NCCL_OK(c, c->rccl->Recv((char *)buf[m.dst] + m.dst_off * es, m.n, dt, m.src, c->comms[m.dst], c->streams[m.dst]), st);
        }
        SPAL_NCCL_OK(c, c->rccl->GroupEnd(), st);   // always: a group left open would swallow every later call
        return st;
    }
    // copy transport
    std::vector<char> is_src(c->ngpus, 0), is_dst(c->ngpus, 0);
    for (const MgMsg &m : msgs) { is_src[m.src] = 1; is_dst[m.dst] = 1; }
    for (int g = 0; g < c->ngpus; ++g)
        if (is_src[g]) {
            DeviceGuard guard(c->devices[g]);
            if (guard.status != SPAL_OK) return guard.status;
            SPAL_HIP_TRY(hipEventRecord(c->ready[g], c->streams[g]));
        }
    for (int g = 0; g < c->ngpus; ++g) {
        if (!is_dst[g]) continue;
        DeviceGuard guard(c->devices[g]);
        if (guard.status != SPAL_OK) return guard.status;
        for (const MgMsg &m : msgs) {
            if (m.dst != g) continue;
            SPAL_HIP_TRY(hipStreamWaitEvent(c->streams[g], c->ready[m.src], 0));
            const char *src = (const char *)buf[m.src] + m.src_off * es;
            char *dst = (char *)buf[g] + m.dst_off * es;
            const size_t bytes = m.n * es;
            if (c->devices[m.src] == c->devices[g])
                SPAL_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->streams[g]));
            else
                SPAL_HIP_TRY(hipMemcpyPeerAsync(dst, c->devices[g], src, c->devices[m.src], bytes, c->streams[g]));
        }
        SPAL_HIP_TRY(hipEventRecord(c->pulled[g], c->streams[g]));
    }
    // a sender may overwrite what it sent only after its readers have it (what ncclSend on its stream gives for free)
    for (int g = 0; g < c->ngpus; ++g) {
        if (!is_src[g]) continue;
        DeviceGuard guard(c->devices[g]);
        if (guard.status != SPAL_OK) return guard.status;
        for (const MgMsg &m : msgs)
            if (m.src == g && m.dst != g) SPAL_HIP_TRY(hipStreamWaitEvent(c->streams[g], c->pulled[m.dst], 0));
    }
    return SPAL_OK;
}

static void mg_csr_free(spal_mg_csr *a) {
    if (!a) return;
    spal_mg *c = a->ctx;
    for (size_t g = 0; g < a->shard.size(); ++g) {
        if (a->shard[g]) spal_csr_destroy(a->shard[g]);
        if (!c || g >= (size_t)c->ngpus) continue;
        DeviceGuard guard(c->devices[g]);
        if (guard.status != SPAL_OK) continue;
        if (g < a->d_x.size()) (void)dev_free(a->d_x[g]);
        if (g < a->d_x2.size()) (void)dev_free(a->d_x2[g]);
        if (g < a->d_yloc.size() && g != 0) (void)dev_free(a->d_yloc[g]);   // (GPU 0's slice is the front of d_yroot)
        if (g < a->d_ypad.size()) (void)dev_free(a->d_ypad[g]);
        if (g < a->d_yall.size()) (void)dev_free(a->d_yall[g]);
        if (g == 0) (void)dev_free(a->d_yroot);
        for (int p = 0; p < kPhases * 2; ++p)
            if (((size_t)g * kPhases * 2 + p) < a->ev.size() && a->ev[(size_t)g * kPhases * 2 + p])
                (void)hipEventDestroy(a->ev[(size_t)g * kPhases * 2 + p]);
    }
    delete a;
}

static std::vector<MgMsg> cut_messages(const std::vector<uint64_t> &lo, const std::vector<uint64_t> &hi,
                                       const std::vector<uint64_t> &own, int G) {
    // reader g needs [lo[g], hi[g]); owner h holds [own[h], own[h + 1]): one message per non-empty intersection, h != g
    std::vector<MgMsg> out;
    for (int g = 0; g < G; ++g)
        for (int h = 0; h < G; ++h) {
            if (h == g) continue;
            const uint64_t a0 = std::max(lo[g], own[h]), a1 = std::min(hi[g], own[h + 1]);
            if (a0 < a1) out.push_back({h, g, a0, a0, a1 - a0});   // (the same position in the vector on both sides)
        }
    return out;
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
    a->d_x.assign(G, nullptr); a->d_x2.assign(G, nullptr); a->d_yloc.assign(G, nullptr);
    a->d_ypad.assign(G, nullptr); a->d_yall.assign(G, nullptr);
    a->need_lo.assign(G, 0); a->need_hi.assign(G, 0);
    a->ev.assign((size_t)G * kPhases * 2, nullptr);
    std::vector<uint64_t> rp;
    for (int g = 0; g < G; ++g) {
        const uint64_t r0 = a->bounds[g], r1 = a->bounds[g + 1];
        const uint64_t e0 = rowptr[r0], e1 = rowptr[r1];
        // the window of x this GPU reads: columns ascend inside a row (src/csr.rs:152-156), so a row's first and
        // last entry bound it
        uint64_t lo = ncols, hi = 0;
        for (uint64_t r = r0; r < r1; ++r)
            if (rowptr[r] < rowptr[r + 1]) {
                lo = std::min(lo, colind[rowptr[r]]);
                hi = std::max(hi, colind[rowptr[r + 1] - 1] + 1);
            }
        if (hi == 0) lo = 0;
        a->need_lo[g] = lo; a->need_hi[g] = hi;
        rp.resize(r1 - r0 + 1);
        for (uint64_t r = r0; r <= r1; ++r) rp[r - r0] = rowptr[r] - e0;  // a row range is a CsrMatrix of its own
        st = a->elem_size == 8
                 ? spal_csr_create_f64(ctx->devices[g], r1 - r0, ncols, rp.data(), rp.size(), colind + e0,
                                       e1 - e0, (const double *)values + e0, e1 - e0, &a->shard[g])
                 : spal_csr_create_f32(ctx->devices[g], r1 - r0, ncols, rp.data(), rp.size(), colind + e0,
                                       e1 - e0, (const float *)values + e0, e1 - e0, &a->shard[g]);
        if (st != SPAL_OK) { mg_csr_free(a); return st; }
        DeviceGuard guard(ctx->devices[g]);   // (restores the caller's device at the end of the iteration)
        if (guard.status != SPAL_OK) { mg_csr_free(a); return guard.status; }
        hipError_t e = dev_alloc((void **)&a->d_x[g], ncols * sizeof(T));
        if (e == hipSuccess && g == 0) e = dev_alloc(&a->d_yroot, nrows * sizeof(T));
        if (e == hipSuccess && g == 0) a->d_yloc[0] = a->d_yroot;   // bounds[0] == 0: GPU 0's kernel writes its slice in place
        if (e == hipSuccess && g != 0) e = dev_alloc((void **)&a->d_yloc[g], (r1 - r0) * sizeof(T));
        for (int p = 0; p < kPhases * 2 && e == hipSuccess; ++p) e = hipEventCreate(&a->ev[(size_t)g * kPhases * 2 + p]);
        if (e != hipSuccess) {
            mg_csr_free(a);
            return fail(e == hipErrorOutOfMemory ? SPAL_ERR_OUT_OF_MEMORY : SPAL_ERR_HIP,
                        "spal_mg_csr_create: %s", hipGetErrorString(e));
        }
    }
    // the exchanges' message lists
    {
        std::vector<uint64_t> root_own(G + 1, ncols);   // "GPU 0 owns all of x"
        root_own[0] = 0;
        a->x_msgs = cut_messages(a->need_lo, a->need_hi, root_own, G);
        // a GPU's slice buffer starts at its first row; in GPU 0's y the slice sits at that row
        for (int g = 1; g < G; ++g) a->y_msgs.push_back({g, 0, 0, a->bounds[g], a->bounds[g + 1] - a->bounds[g]});
        if (nrows == ncols) a->halo_msgs = cut_messages(a->need_lo, a->need_hi, a->bounds, G);
    }
    *out = a;
    return SPAL_OK;
}

static int mg_sync(spal_mg *c) {
    for (int g = 0; g < c->ngpus; ++g) {
        DeviceGuard guard(c->devices[g]);
        if (guard.status != SPAL_OK) return guard.status;
        SPAL_HIP_TRY(hipStreamSynchronize(c->streams[g]));
    }
    return SPAL_OK;
}

// the whole of x from GPU 0 to everyone (ncclBroadcast; copy transport: every GPU pulls all of it)
static int mg_broadcast_x(spal_mg_csr *a) {
    spal_mg *c = a->ctx;
    SPAL_TRY(mg_phase_mark(a, kPhaseX, 0));
    int st = SPAL_OK;
    if (c->ngpus > 1) {
        if (c->transport == kTransportRccl) {
            const ncclDataType_t dt = a->elem_size == 8 ? ncclFloat64 : ncclFloat32;
            SPAL_NCCL_OK(c, c->rccl->GroupStart(), st);
            if (st != SPAL_OK) return st;
            for (int g = 0; g < c->ngpus && st == SPAL_OK; ++g)
                SPAL_NCCL_OK(c, c->rccl->Broadcast(a->d_x[g], a->d_x[g], a->ncols, dt, 0, c->comms[g], c->streams[g]), st);
            SPAL_NCCL_OK(c, c->rccl->GroupEnd(), st);
        } else {
            std::vector<MgMsg> all;
            for (int g = 1; g < c->ngpus; ++g) all.push_back({0, g, 0, 0, a->ncols});
            st = mg_exchange(a, all, a->d_x);
        }
    }
    if (st != SPAL_OK) return st;
    a->last_was_halo = false;
    return mg_phase_mark(a, kPhaseX, 1);
}

// every GPU its window of x only
static int mg_scatter_x(spal_mg_csr *a) {
    SPAL_TRY(mg_phase_mark(a, kPhaseX, 0));
    SPAL_TRY(mg_exchange(a, a->x_msgs, a->d_x));
    a->last_was_halo = false;
    return mg_phase_mark(a, kPhaseX, 1);
}

static int mg_launch_shards(spal_mg_csr *a, const std::vector<void *> &xin, const std::vector<void *> &yout,
                            bool y_is_vector) {
    spal_mg *c = a->ctx;
    SPAL_TRY(mg_phase_mark(a, kPhaseCompute, 0));
    for (int g = 0; g < c->ngpus; ++g) {
        DeviceGuard guard(c->devices[g]);
        if (guard.status != SPAL_OK) return guard.status;
        // y_is_vector: the slice goes to its place inside a full-length vector (halo steps)
        char *y = (char *)yout[g] + (y_is_vector ? a->bounds[g] * (size_t)a->elem_size : 0);
        SPAL_TRY(a->elem_size == 8
                     ? spal_csr_spmv_dev_f64(a->shard[g], (const double *)xin[g], (double *)y, c->streams[g])
                     : spal_csr_spmv_dev_f32(a->shard[g], (const float *)xin[g], (float *)y, c->streams[g]));
    }
    return mg_phase_mark(a, kPhaseCompute, 1);
}

static int mg_spmv_local(spal_mg_csr *a) {
    SPAL_TRY(mg_launch_shards(a, a->d_x, a->d_yloc, false));
    a->last_was_halo = false;
    return SPAL_OK;
}

// slices -> GPU 0's y (GPU 0's own slice is already there)
static int mg_gather_y(spal_mg_csr *a) {
    spal_mg *c = a->ctx;
    SPAL_TRY(mg_phase_mark(a, kPhaseY, 0));
    if (a->last_was_halo) {
        // the slices of the last product sit inside the current x vectors: GPU 0 pulls them from there
        std::vector<void *> buf(a->d_x);
        {   // GPU 0's own slice: vector -> y
            DeviceGuard guard(c->devices[0]);
            if (guard.status != SPAL_OK) return guard.status;
            SPAL_HIP_TRY(hipMemcpyAsync(a->d_yroot, a->d_x[0], a->bounds[1] * (size_t)a->elem_size, hipMemcpyDeviceToDevice, c->streams[0]));
        }
        buf[0] = a->d_yroot;
        std::vector<MgMsg> msgs(a->y_msgs);
        for (MgMsg &m : msgs) m.src_off = m.dst_off;   // in a vector the slice sits at its rows' positions
        SPAL_TRY(mg_exchange(a, msgs, buf));
    } else {
        SPAL_TRY(mg_exchange(a, a->y_msgs, a->d_yloc));
    }
    return mg_phase_mark(a, kPhaseY, 1);
}

// one step of iterative use: y = A * x on every GPU's rows, written into the other vector of the pair at the
// rows' own positions, then the entries of y each GPU's rows read as columns are completed from their owners;
// the vectors swap roles
static int mg_spmv_halo(spal_mg_csr *a) {
    spal_mg *c = a->ctx;
    if (a->nrows != a->ncols)
        return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_spmv_halo needs a square matrix (y feeds back as x)");
    for (int g = 0; g < c->ngpus; ++g)
        if (!a->d_x2[g]) {
            DeviceGuard guard(c->devices[g]);
            if (guard.status != SPAL_OK) return guard.status;
            SPAL_HIP_TRY(dev_alloc(&a->d_x2[g], a->ncols * (size_t)a->elem_size));
        }
    SPAL_TRY(mg_launch_shards(a, a->d_x, a->d_x2, true));
    SPAL_TRY(mg_phase_mark(a, kPhaseHalo, 0));
    SPAL_TRY(mg_exchange(a, a->halo_msgs, a->d_x2));
    SPAL_TRY(mg_phase_mark(a, kPhaseHalo, 1));
    a->d_x.swap(a->d_x2);
    a->last_was_halo = true;
    return SPAL_OK;
}

// all-gather path: every GPU ends with all of y (slices padded to the longest)
static int mg_multiply_allgather(spal_mg_csr *a) {
    spal_mg *c = a->ctx;
    const size_t es = (size_t)a->elem_size;
    for (int g = 0; g < c->ngpus; ++g)
        if (!a->d_yall[g]) {
            DeviceGuard guard(c->devices[g]);
            if (guard.status != SPAL_OK) return guard.status;
            if (!a->d_ypad[g]) {   // (guarded on its own pointer: a failed d_yall below must not allocate it again next time)
                SPAL_HIP_TRY(dev_alloc(&a->d_ypad[g], a->max_rows * es));
                SPAL_HIP_TRY(hipMemsetAsync(a->d_ypad[g], 0, a->max_rows * es, c->streams[g]));
            }
            SPAL_HIP_TRY(dev_alloc(&a->d_yall[g], (size_t)c->ngpus * a->max_rows * es));
        }
    SPAL_TRY(mg_launch_shards(a, a->d_x, a->d_ypad, false));
    a->last_was_halo = false;
    SPAL_TRY(mg_phase_mark(a, kPhaseY, 0));
    int st = SPAL_OK;
    if (c->ngpus == 1 || c->transport == kTransportCopy) {
        // every GPU pulls every padded slice into its copy
        for (int g = 0; g < c->ngpus; ++g) {
            DeviceGuard guard(c->devices[g]);
            if (guard.status != SPAL_OK) return guard.status;
            SPAL_HIP_TRY(hipEventRecord(c->ready[g], c->streams[g]));
        }
        for (int g = 0; g < c->ngpus; ++g) {
            DeviceGuard guard(c->devices[g]);
            if (guard.status != SPAL_OK) return guard.status;
            for (int h = 0; h < c->ngpus; ++h) {
                SPAL_HIP_TRY(hipStreamWaitEvent(c->streams[g], c->ready[h], 0));
                char *dst = (char *)a->d_yall[g] + (size_t)h * a->max_rows * es;
                if (c->devices[h] == c->devices[g])
                    SPAL_HIP_TRY(hipMemcpyAsync(dst, a->d_ypad[h], a->max_rows * es, hipMemcpyDeviceToDevice, c->streams[g]));
                else
                    SPAL_HIP_TRY(hipMemcpyPeerAsync(dst, c->devices[g], a->d_ypad[h], c->devices[h], a->max_rows * es, c->streams[g]));
            }
            SPAL_HIP_TRY(hipEventRecord(c->pulled[g], c->streams[g]));
        }
        for (int g = 0; g < c->ngpus; ++g) {
            DeviceGuard guard(c->devices[g]);
            if (guard.status != SPAL_OK) return guard.status;
            for (int h = 0; h < c->ngpus; ++h)
                if (h != g) SPAL_HIP_TRY(hipStreamWaitEvent(c->streams[g], c->pulled[h], 0));
        }
    } else {
        const ncclDataType_t dt = a->elem_size == 8 ? ncclFloat64 : ncclFloat32;
        SPAL_NCCL_OK(c, c->rccl->GroupStart(), st);
        if (st != SPAL_OK) return st;
        for (int g = 0; g < c->ngpus && st == SPAL_OK; ++g)
            SPAL_NCCL_OK(c, c->rccl->AllGather(a->d_ypad[g], a->d_yall[g], a->max_rows, dt, c->comms[g], c->streams[g]), st);
        SPAL_NCCL_OK(c, c->rccl->GroupEnd(), st);
        if (st != SPAL_OK) return st;
    }
    return mg_phase_mark(a, kPhaseY, 1);
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
    {
        DeviceGuard guard(c->devices[0]);
        if (guard.status != SPAL_OK) return guard.status;
        SPAL_HIP_TRY(hipMemcpyAsync(a->d_x[0], x, a->ncols * sizeof(T), hipMemcpyHostToDevice, c->streams[0]));
    }
    // windows when they are a real saving, else the plain broadcast (a tree / ring beats N - 1 near-full copies from GPU 0)
    uint64_t sent = 0;
    for (const MgMsg &m : a->x_msgs) sent += m.n;
    if (c->ngpus > 1 && sent * 4 > (uint64_t)(c->ngpus - 1) * a->ncols * 3) SPAL_TRY(mg_broadcast_x(a));
    else SPAL_TRY(mg_scatter_x(a));
    SPAL_TRY(mg_spmv_local(a));
    SPAL_TRY(mg_gather_y(a));
    {
        DeviceGuard guard(c->devices[0]);
        if (guard.status != SPAL_OK) return guard.status;
        SPAL_HIP_TRY(hipMemcpyAsync(y, a->d_yroot, a->nrows * sizeof(T), hipMemcpyDeviceToHost, c->streams[0]));
    }
    return mg_sync(c);
}

extern "C" {

int spal_mg_create_transport(int ngpus, const int *devices, int transport, spal_mg_t *out) {
    if (!out || ngpus < 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_create: need ngpus >= 1 and out");
    *out = nullptr;
    if (transport < -1 || transport > 1) return fail(SPAL_ERR_INVALID_ARGUMENT, "transport must be -1 (auto), 0 (rccl) or 1 (copy)");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        (void)hipGetLastError();
        return fail(SPAL_ERR_NO_DEVICE, "no HIP device available");
    }
    spal_mg *c = new spal_mg;
    c->ngpus = ngpus;
    for (int g = 0; g < ngpus; ++g) {
        const int d = devices ? devices[g] : g;
        if (d < 0 || d >= count) {
            delete c;
            return devices ? fail(SPAL_ERR_INVALID_ARGUMENT, "device %d out of range", d)
                           : fail(SPAL_ERR_INVALID_ARGUMENT, "%d GPUs requested, %d visible", ngpus, count);
        }
        for (int h : c->devices) c->repeats |= (h == d);
        c->devices.push_back(d);
    }
    if (c->repeats && transport == kTransportRccl) {
        delete c;
        return fail(SPAL_ERR_INVALID_ARGUMENT, "a device list with repeats needs the copy transport (RCCL refuses duplicate devices)");
    }
    c->transport = transport >= 0 ? transport : (c->repeats ? kTransportCopy : kTransportRccl);
    if (ngpus == 1 && transport != kTransportRccl) c->transport = kTransportCopy;   // nothing to exchange: no RCCL needed
    // (ngpus == 1 with the RCCL transport asked for explicitly: a one-rank communicator -- loads RCCL and runs its
    //  calls, which is all of that path a 1-GPU box can exercise)
    c->streams.assign(ngpus, nullptr);
    c->ready.assign(ngpus, nullptr);
    c->pulled.assign(ngpus, nullptr);
    for (int g = 0; g < ngpus; ++g) {
        DeviceGuard guard(c->devices[g]);
        hipError_t e = guard.status == SPAL_OK ? hipSuccess : hipErrorInvalidDevice;
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->streams[g], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready[g], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->pulled[g], hipEventDisableTiming);
        if (e == hipSuccess && c->transport == kTransportCopy)
            for (int h = 0; h < ngpus; ++h) {   // direct xGMI copies between distinct devices
                int can = 0;
                if (c->devices[h] != c->devices[g] && hipDeviceCanAccessPeer(&can, c->devices[g], c->devices[h]) == hipSuccess && can)
                    if (hipDeviceEnablePeerAccess(c->devices[h], 0) != hipSuccess) (void)hipGetLastError();   // (already enabled)
            }
        if (e != hipSuccess) { spal_mg_destroy(c); return fail(SPAL_ERR_HIP, "spal_mg_create: %s", hipGetErrorString(e)); }
    }
    if (c->transport == kTransportRccl) {  // one communicator per GPU, all owned by this process
        int st = rccl_load(&c->rccl);
        if (st != SPAL_OK) { spal_mg_destroy(c); return st; }
        c->comms.assign(ngpus, nullptr);
        int prev = -1;
        (void)hipGetDevice(&prev);
        ncclResult_t r = c->rccl->CommInitAll(c->comms.data(), ngpus, c->devices.data());
        if (prev >= 0) (void)hipSetDevice(prev);
        if (r != ncclSuccess) {
            c->comms.clear();
            const char *msg = c->rccl->GetErrorString(r);
            spal_mg_destroy(c);
            return fail(SPAL_ERR_HIP, "ncclCommInitAll failed: %s", msg);
        }
    }
    *out = c;
    return SPAL_OK;
}

int spal_mg_create(int ngpus, const int *devices, spal_mg_t *out) {
    return spal_mg_create_transport(ngpus, devices, -1, out);
}

int spal_mg_destroy(spal_mg_t c) {
    if (!c) return SPAL_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (size_t g = 0; g < c->comms.size(); ++g)
        if (c->comms[g] && c->rccl) (void)c->rccl->CommDestroy(c->comms[g]);
    for (size_t g = 0; g < c->streams.size(); ++g) {
        (void)hipSetDevice(c->devices[g]);
        if (c->streams[g]) (void)hipStreamDestroy(c->streams[g]);
        if (g < c->ready.size() && c->ready[g]) (void)hipEventDestroy(c->ready[g]);
        if (g < c->pulled.size() && c->pulled[g]) (void)hipEventDestroy(c->pulled[g]);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete c;
    return SPAL_OK;
}

int spal_mg_device_count(spal_mg_t c, int *ngpus) {
    if (!c || !ngpus) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_device_count: null argument");
    *ngpus = c->ngpus;
    return SPAL_OK;
}

int spal_mg_transport(spal_mg_t c, int *transport) {
    if (!c || !transport) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_transport: null argument");
    *transport = c->transport;
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
    (void)mg_sync(a->ctx);
    mg_csr_free(a);
    if (prev >= 0) (void)hipSetDevice(prev);
    return SPAL_OK;
}
int spal_mg_csr_partition(spal_mg_csr_t a, uint64_t *bounds) {
    if (!a || !bounds) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_partition: null argument");
    for (size_t g = 0; g < a->bounds.size(); ++g) bounds[g] = a->bounds[g];
    return SPAL_OK;
}
int spal_mg_csr_windows(spal_mg_csr_t a, uint64_t *need_lo, uint64_t *need_hi) {
    if (!a || !need_lo || !need_hi) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_windows: null argument");
    for (size_t g = 0; g < a->need_lo.size(); ++g) { need_lo[g] = a->need_lo[g]; need_hi[g] = a->need_hi[g]; }
    return SPAL_OK;
}
int spal_mg_csr_exchange_bytes(spal_mg_csr_t a, uint64_t *x_scatter, uint64_t *y_gather, uint64_t *halo) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_exchange_bytes: handle is NULL");
    auto total = [&](const std::vector<MgMsg> &v) { uint64_t s = 0; for (const MgMsg &m : v) s += m.n * (uint64_t)a->elem_size; return s; };
    if (x_scatter) *x_scatter = total(a->x_msgs);
    if (y_gather) *y_gather = total(a->y_msgs);
    if (halo) *halo = total(a->halo_msgs);
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
    std::lock_guard<std::mutex> lock(a->mu);
    *x_dev = a->d_x[0];
    return SPAL_OK;
}
#define SPAL_MG_CALL(name, fn)                                                            \
    int name(spal_mg_csr_t a) {                                                           \
        if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, #name ": handle is NULL");         \
        std::lock_guard<std::mutex> lock(a->mu);                                          \
        return fn(a);                                                                     \
    }
SPAL_MG_CALL(spal_mg_csr_broadcast_x, mg_broadcast_x)
SPAL_MG_CALL(spal_mg_csr_scatter_x, mg_scatter_x)
SPAL_MG_CALL(spal_mg_csr_spmv_local, mg_spmv_local)
SPAL_MG_CALL(spal_mg_csr_gather_y, mg_gather_y)
SPAL_MG_CALL(spal_mg_csr_spmv_halo, mg_spmv_halo)
SPAL_MG_CALL(spal_mg_csr_spmv_resident, mg_multiply_allgather)
#undef SPAL_MG_CALL

int spal_mg_csr_y_root(spal_mg_csr_t a, void **y_dev, uint64_t *slice_stride) {
    if (!a || !y_dev) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_y_root: null argument");
    if (!a->d_yall[0]) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_y_root: no spal_mg_csr_spmv_resident has run yet");
    *y_dev = a->d_yall[0];
    if (slice_stride) *slice_stride = a->max_rows;
    return SPAL_OK;
}
int spal_mg_csr_y_gathered(spal_mg_csr_t a, void **y_dev) {
    if (!a || !y_dev) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_y_gathered: null argument");
    *y_dev = a->d_yroot;
    return SPAL_OK;
}
int spal_mg_csr_synchronize(spal_mg_csr_t a) {
    if (!a) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_synchronize: handle is NULL");
    return mg_sync(a->ctx);
}
int spal_mg_csr_timing(spal_mg_csr_t a, double *ms) {
    if (!a || !ms) return fail(SPAL_ERR_INVALID_ARGUMENT, "spal_mg_csr_timing: null argument");
    std::lock_guard<std::mutex> lock(a->mu);
    SPAL_TRY(mg_sync(a->ctx));
    for (int p = 0; p < kPhases; ++p) {
        ms[p] = -1.0;
        if (!a->timed[p]) continue;
        double worst = 0.0;
        for (int g = 0; g < a->ctx->ngpus; ++g) {
            DeviceGuard guard(a->ctx->devices[g]);
            if (guard.status != SPAL_OK) return guard.status;
            float t = 0.f;
            SPAL_HIP_TRY(hipEventElapsedTime(&t, mg_ev(a, g, p, 0), mg_ev(a, g, p, 1)));
            worst = std::max(worst, (double)t);
        }
        ms[p] = worst;
    }
    return SPAL_OK;
}

}  // extern "C"
