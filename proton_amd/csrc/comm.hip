// comm.hip -- the multi-GPU exchange of the path behind the C ABI: one process per
// GPU, RCCL over xGMI.  The reference has no communication of any kind (single process, single thread).
//
// What travels per assembly step is small by construction: the cells shard by rows, every slab assembles the CSR rows
// of the faces it owns (condensed.hip), and the only rows whose cells live on two ranks are a slab's bottom faces --
// so each rank sends the packed top-face rows of its top cell row (fbs (nf + 1) doubles per cell) one slab up:
// point-to-point, one xGMI link, Nx cells.  All-gather and all-reduce are here for the caller that wants the whole
// system on every rank (the north star's literal exchange) and for the dot products of a distributed solve.
//
// RCCL is bound at run time (dlopen): a process that already holds a copy (PyTorch ships its own librccl, built against
// the HIP runtime it also ships) must use THAT copy -- two HIP runtimes cannot both own the device -- and a build or a
// CPU-only load of this library must not need RCCL at all.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "../../include/proton_amd.h"

namespace {

// the slice of rccl.h this file uses (stable NCCL ABI)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclInt8 = 0, ncclUint8 = 1, ncclFloat64 = 8 };
enum { ncclSum = 0 };

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    std::string error;
};

Rccl *rccl()
{
    static Rccl r;
    if (r.handle || !r.error.empty()) return &r;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : names)                                  // a copy the process already holds (PyTorch's) first
        if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (const char *env = getenv("PA_RCCL_LIB")) { if (!r.handle) r.handle = dlopen(env, RTLD_NOW | RTLD_GLOBAL); }
    for (const char *n : names)
        if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) { r.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return &r; }
#define PA_SYM(field, name)                                                                     \
    do {                                                                                        \
        *(void **)(&r.field) = dlsym(r.handle, name);                                           \
        if (!r.field) { r.error = std::string("librccl lacks ") + name; return &r; }            \
    } while (0)
    PA_SYM(GetUniqueId, "ncclGetUniqueId"); PA_SYM(CommInitRank, "ncclCommInitRank"); PA_SYM(CommDestroy, "ncclCommDestroy");
    PA_SYM(GetErrorString, "ncclGetErrorString"); PA_SYM(Send, "ncclSend"); PA_SYM(Recv, "ncclRecv");
    PA_SYM(AllGather, "ncclAllGather"); PA_SYM(AllReduce, "ncclAllReduce"); PA_SYM(GroupStart, "ncclGroupStart");
    PA_SYM(GroupEnd, "ncclGroupEnd");
#undef PA_SYM
    return &r;
}

}  // namespace

struct pa_comm {
    int device = 0, nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t main = nullptr;        // the context's stream: producers and consumers of the exchanged buffers
    hipStream_t side = nullptr;        // the collectives' own stream, so that they overlap kernels enqueued after them
    hipEvent_t ready = nullptr, done = nullptr;
    bool pending = false;
    double *scratch = nullptr;         // 8 doubles on the device: the scalars of a distributed solve (pa_comm_cg_transport)
    std::string last_error;
};

extern "C" __attribute__((visibility("hidden"))) void *pa_context_stream_(pa_context *ctx);       // capi.hip
extern "C" __attribute__((visibility("hidden"))) int pa_context_device_(pa_context *ctx);

#define PA_CHIP(c, call)                                                                           \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) { (c)->last_error = std::string(#call) + ": " + hipGetErrorString(e_); return PA_ERR_HIP; } \
    } while (0)
#define PA_CNCCL(c, call)                                                                          \
    do {                                                                                           \
        ncclResult_t r_ = (call);                                                                  \
        if (r_ != 0) { (c)->last_error = std::string(#call) + ": " + rccl()->GetErrorString(r_); return PA_ERR_COMM; } \
    } while (0)

static const ncclResult_t ncclSuccess = 0;
static int comm_nccl_error(pa_comm *c, const char *what, ncclResult_t r)
{
    c->last_error = std::string(what ? what : "nccl") + ": " + rccl()->GetErrorString(r);
    return PA_ERR_COMM;
}

extern "C" {

int pa_comm_unique_id(void *id_out, size_t bytes)
{
    if (!id_out || bytes < PA_COMM_ID_BYTES) return PA_ERR_INVALID_ARG;
    Rccl *r = rccl();
    if (!r->error.empty()) { std::fprintf(stderr, "proton_amd: %s\n", r->error.c_str()); return PA_ERR_COMM; }
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != 0) return PA_ERR_COMM;
    std::memcpy(id_out, &id, sizeof(id));
    return PA_OK;
}

int pa_comm_create(pa_context *ctx, int nranks, int rank, const void *unique_id, pa_comm **out)
{
    if (!ctx || !out || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) return PA_ERR_INVALID_ARG;
    *out = nullptr;
    Rccl *r = rccl();
    if (!r->error.empty()) { std::fprintf(stderr, "proton_amd: %s\n", r->error.c_str()); return PA_ERR_COMM; }
    pa_comm *c = new (std::nothrow) pa_comm();
    if (!c) return PA_ERR_INVALID_ARG;
    c->device = pa_context_device_(ctx); c->nranks = nranks; c->rank = rank;
    c->main = (hipStream_t)pa_context_stream_(ctx);
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        std::fprintf(stderr, "proton_amd: pa_comm_create: %s\n", hipGetErrorString(e));
        delete c;
        return PA_ERR_HIP;
    }
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    const ncclResult_t st = r->CommInitRank(&c->comm, nranks, id, rank);
    if (st != 0) {
        std::fprintf(stderr, "proton_amd: ncclCommInitRank(rank %d of %d): %s\n", rank, nranks, r->GetErrorString(st));
        (void)hipEventDestroy(c->ready); (void)hipEventDestroy(c->done); (void)hipStreamDestroy(c->side);
        delete c;
        return PA_ERR_COMM;
    }
    *out = c;
    return PA_OK;
}

int pa_comm_destroy(pa_comm *c)
{
    if (!c) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->side);
    if (c->comm) (void)rccl()->CommDestroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
    (void)hipEventDestroy(c->ready); (void)hipEventDestroy(c->done); (void)hipStreamDestroy(c->side);
    delete c;
    return PA_OK;
}

const char *pa_comm_last_error(pa_comm *c) { return c ? c->last_error.c_str() : "null communicator"; }

int pa_comm_info(pa_comm *c, int *nranks, int *rank)
{
    if (!c) return PA_ERR_INVALID_ARG;
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    return PA_OK;
}

// everything enqueued so far on the context's stream is visible to the collective; the collective runs on the side stream
static int comm_begin(pa_comm *c)
{
    PA_CHIP(c, hipSetDevice(c->device));
    PA_CHIP(c, hipEventRecord(c->ready, c->main));
    PA_CHIP(c, hipStreamWaitEvent(c->side, c->ready, 0));
    return PA_OK;
}
static int comm_end(pa_comm *c)
{
    PA_CHIP(c, hipEventRecord(c->done, c->side));
    c->pending = true;
    return PA_OK;
}

int pa_comm_halo_exchange_start(pa_comm *c, const double *d_send_up, size_t send_count, double *d_recv_below, size_t recv_count)
{
    if (!c) return PA_ERR_INVALID_ARG;
    const bool send = d_send_up != nullptr && send_count > 0 && c->rank + 1 < c->nranks;
    const bool recv = d_recv_below != nullptr && recv_count > 0 && c->rank > 0;
    if (!send && !recv) return PA_OK;
    int st = comm_begin(c);
    if (st != PA_OK) return st;
    Rccl *r = rccl();
    PA_CNCCL(c, r->GroupStart());
    // a failing Send / Recv must not leave the group open: remember the first error, close the group, then report it
    ncclResult_t first = ncclSuccess;
    const char *what = nullptr;
    if (recv && first == ncclSuccess) { first = r->Recv(d_recv_below, recv_count, ncclFloat64, c->rank - 1, c->comm, c->side); what = "ncclRecv"; }
    if (send && first == ncclSuccess) { first = r->Send(d_send_up, send_count, ncclFloat64, c->rank + 1, c->comm, c->side); what = "ncclSend"; }
    const ncclResult_t ge = r->GroupEnd();
    if (first != ncclSuccess) return comm_nccl_error(c, what, first);
    if (ge != ncclSuccess) return comm_nccl_error(c, "ncclGroupEnd", ge);
    return comm_end(c);
}

int pa_comm_neighbour_exchange_start(pa_comm *c, const double *d_send_lo, size_t n_send_lo, const double *d_send_hi, size_t n_send_hi,
                                     double *d_recv_lo, size_t n_recv_lo, double *d_recv_hi, size_t n_recv_hi)
{
    if (!c) return PA_ERR_INVALID_ARG;
    const bool lo = c->rank > 0, hi = c->rank + 1 < c->nranks;
    const bool s_lo = lo && d_send_lo && n_send_lo, s_hi = hi && d_send_hi && n_send_hi;
    const bool r_lo = lo && d_recv_lo && n_recv_lo, r_hi = hi && d_recv_hi && n_recv_hi;
    if (!s_lo && !s_hi && !r_lo && !r_hi) return PA_OK;
    int st = comm_begin(c);
    if (st != PA_OK) return st;
    Rccl *r = rccl();
    PA_CNCCL(c, r->GroupStart());
    ncclResult_t first = ncclSuccess;              // (as above: the group is closed on every path)
    const char *what = nullptr;
    if (r_lo && first == ncclSuccess) { first = r->Recv(d_recv_lo, n_recv_lo, ncclFloat64, c->rank - 1, c->comm, c->side); what = "ncclRecv"; }
    if (r_hi && first == ncclSuccess) { first = r->Recv(d_recv_hi, n_recv_hi, ncclFloat64, c->rank + 1, c->comm, c->side); what = "ncclRecv"; }
    if (s_lo && first == ncclSuccess) { first = r->Send(d_send_lo, n_send_lo, ncclFloat64, c->rank - 1, c->comm, c->side); what = "ncclSend"; }
    if (s_hi && first == ncclSuccess) { first = r->Send(d_send_hi, n_send_hi, ncclFloat64, c->rank + 1, c->comm, c->side); what = "ncclSend"; }
    const ncclResult_t ge = r->GroupEnd();
    if (first != ncclSuccess) return comm_nccl_error(c, what, first);
    if (ge != ncclSuccess) return comm_nccl_error(c, "ncclGroupEnd", ge);
    return comm_end(c);
}

// ---- the transport of pa_conjugated_gradient_rows over this communicator (user = the communicator) ----
static int cgt_scratch(pa_comm *c, double **d)
{
    if (!c->scratch) PA_CHIP(c, hipMalloc((void **)&c->scratch, 8 * sizeof(double)));
    *d = c->scratch;
    return PA_OK;
}
static int cgt_allreduce(void *user, double *vals, int n)
{
    pa_comm *c = (pa_comm *)user;
    if (n > 8) return 1;
    if (c->nranks == 1) return 0;
    double *d = nullptr;
    if (cgt_scratch(c, &d) != PA_OK) return 1;
    if (hipMemcpyAsync(d, vals, n * sizeof(double), hipMemcpyHostToDevice, c->main) != hipSuccess) return 1;
    if (pa_comm_allreduce_sum_start(c, d, (size_t)n) != PA_OK || pa_comm_wait(c) != PA_OK) return 1;
    if (hipMemcpyAsync(vals, d, n * sizeof(double), hipMemcpyDeviceToHost, c->main) != hipSuccess) return 1;
    return hipStreamSynchronize(c->main) == hipSuccess ? 0 : 1;
}
static int cgt_halo(void *user, const double *send_lo, size_t n_send_lo, const double *send_hi, size_t n_send_hi, double *recv_lo,
                    size_t n_recv_lo, double *recv_hi, size_t n_recv_hi, void *)
{
    pa_comm *c = (pa_comm *)user;
    if (pa_comm_neighbour_exchange_start(c, send_lo, n_send_lo, send_hi, n_send_hi, recv_lo, n_recv_lo, recv_hi, n_recv_hi) != PA_OK) return 1;
    return pa_comm_wait(c) == PA_OK ? 0 : 1;
}
static int cgt_counts(void *user, int64_t need_lo, int64_t need_hi, int64_t *give_lo, int64_t *give_hi)
{
    // what I read below is what rank - 1 gives of its LAST entries; what I read above, rank + 1 of its FIRST: the counts
    // travel as doubles (exact below 2^53) through the neighbour exchange
    pa_comm *c = (pa_comm *)user;
    *give_lo = *give_hi = 0;
    if (c->nranks == 1) return 0;
    double *d = nullptr;
    if (cgt_scratch(c, &d) != PA_OK) return 1;
    double h[4] = {(double)need_lo, (double)need_hi, 0.0, 0.0};
    if (hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, c->main) != hipSuccess) return 1;
    if (pa_comm_neighbour_exchange_start(c, d, 1, d + 1, 1, d + 2, 1, d + 3, 1) != PA_OK || pa_comm_wait(c) != PA_OK) return 1;
    if (hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->main) != hipSuccess) return 1;
    if (hipStreamSynchronize(c->main) != hipSuccess) return 1;
    if (c->rank > 0) *give_lo = (int64_t)h[2];                   // rank - 1 sent its need_hi: that many of my first entries
    if (c->rank + 1 < c->nranks) *give_hi = (int64_t)h[3];       // rank + 1 sent its need_lo: that many of my last entries
    return 0;
}
int pa_comm_cg_transport(pa_comm *c, pa_cg_transport *out)
{
    if (!c || !out) return PA_ERR_INVALID_ARG;
    out->user = c;
    out->allreduce_sum = cgt_allreduce;
    out->halo = cgt_halo;
    out->neighbour_counts = cgt_counts;
    return PA_OK;
}

int pa_comm_allgather_start(pa_comm *c, const void *d_send, void *d_recv, size_t bytes_per_rank)
{
    if (!c || !d_send || !d_recv) return PA_ERR_INVALID_ARG;
    if (bytes_per_rank == 0) return PA_OK;
    int st = comm_begin(c);
    if (st != PA_OK) return st;
    if (bytes_per_rank % 8 == 0) PA_CNCCL(c, rccl()->AllGather(d_send, d_recv, bytes_per_rank / 8, ncclFloat64, c->comm, c->side));
    else PA_CNCCL(c, rccl()->AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, c->comm, c->side));
    return comm_end(c);
}

int pa_comm_allreduce_sum_start(pa_comm *c, double *d_buf, size_t count)
{
    if (!c || !d_buf) return PA_ERR_INVALID_ARG;
    if (count == 0) return PA_OK;
    int st = comm_begin(c);
    if (st != PA_OK) return st;
    PA_CNCCL(c, rccl()->AllReduce(d_buf, d_buf, count, ncclFloat64, ncclSum, c->comm, c->side));
    return comm_end(c);
}

int pa_comm_wait(pa_comm *c)
{
    if (!c) return PA_ERR_INVALID_ARG;
    if (!c->pending) return PA_OK;
    PA_CHIP(c, hipSetDevice(c->device));
    PA_CHIP(c, hipStreamWaitEvent(c->main, c->done, 0));      // later work on the context's stream sees the received data
    c->pending = false;
    return PA_OK;
}

}  // extern "C"
