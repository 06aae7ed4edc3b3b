// comm.hip -- the collective group of the C ABI (oea_comm_init, oea_allgather_rows, oea_allreduce_i64, ...).
//
// One process per GPU; the collectives are RCCL's (ring / tree over xGMI).  The reference has no distributed code at all
// (SURVEY F2), so nothing is replaced here: these calls are the multi-GPU exchange points of the partitioned
// translational step (oea_part_*: reduce-scatter of the packed gradients, all-gather of the updated rows), of the
// row-sharded evaluation (int64 / fp64 sums of the metrics, all-gather of argmax rows) and of the row-sharded graph
// aggregates (all-gather of a layer's output rows), for a host that is not Python.  The Python host uses
// torch.distributed (backend "nccl" = the same RCCL).
//
// RCCL is resolved with dlopen at oea_comm_init time: libopenea_hip.so has no link-time dependency on it, a single-GPU
// process never loads it, and a process that already holds an RCCL (PyTorch's) shares that copy.
//
// A communicator can also be built over HOST CALLBACKS (oea_comm_init_callbacks): the same entry points then hand every
// collective to the caller's function.  RCCL wants one GPU per rank; with the callbacks the one-call partitioned epoch
// (oea_triple_epoch_range_comm) runs with 2 and 4 ranks sharing ONE GPU over torch.distributed's gloo group, which is how
// the build pool (one GPU) tests the call path a multi-GPU node runs over RCCL.
#include <dlfcn.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace {

// the subset of rccl.h this file needs (ABI-stable NCCL 2 API)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclInt64 = 4, ncclFloat32 = 7, ncclFloat64 = 8 };       // ncclDataType_t
enum { ncclSum = 0 };                                            // ncclRedOp_t

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.handle ? &r : nullptr;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) return nullptr;
#define OEA_SYM(field, name)                                                        \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name));          \
    if (!r.field) { r.handle = nullptr; return nullptr; }
    OEA_SYM(GetUniqueId, "ncclGetUniqueId")
    OEA_SYM(CommInitRank, "ncclCommInitRank")
    OEA_SYM(CommDestroy, "ncclCommDestroy")
    OEA_SYM(AllReduce, "ncclAllReduce")
    OEA_SYM(AllGather, "ncclAllGather")
    OEA_SYM(ReduceScatter, "ncclReduceScatter")
    OEA_SYM(Send, "ncclSend")
    OEA_SYM(Recv, "ncclRecv")
    OEA_SYM(GroupStart, "ncclGroupStart")
    OEA_SYM(GroupEnd, "ncclGroupEnd")
    OEA_SYM(GetErrorString, "ncclGetErrorString")
#undef OEA_SYM
    return &r;
}

}  // namespace

struct oea_comm {
    ncclComm_t comm;
    int rank, nranks;
    oea_comm_callback fn;           // non-null: every collective goes to the host callback instead of RCCL
    void *user;
    oea_comm_alltoallv_callback a2a;       // the all-to-all of a callback communicator (oea_comm_set_alltoallv)
    // phase profile of oea_triple_epoch_range_comm (oea_comm_profile_begin / _end): events at the phase boundaries
    bool profiling;
    std::vector<hipEvent_t> events;        // OEA_COMM_PHASES + 1 per step
};

namespace {
int nccl_dtype(int32_t dtype) { return dtype == OEA_COMM_F32 ? ncclFloat32 : (dtype == OEA_COMM_F64 ? ncclFloat64 : ncclInt64); }
}

#define OEA_CHECK_RCCL(expr)                                                                       \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != 0) {                                                                             \
            oea::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, rccl()->GetErrorString(_r)); \
            return OEA_EHIP;                                                                       \
        }                                                                                          \
    } while (0)

extern "C" {

int oea_comm_unique_id(void *id_out_128) {
    OEA_REQUIRE(id_out_128, "null pointer");
    Rccl *r = rccl();
    if (!r) { oea::set_error("RCCL (librccl.so.1) cannot be loaded: %s", dlerror()); return OEA_EUNSUPPORTED; }
    OEA_CHECK_RCCL(r->GetUniqueId(static_cast<ncclUniqueId *>(id_out_128)));
    return OEA_OK;
}

int oea_comm_init(const void *unique_id_128, int32_t rank, int32_t nranks, oea_comm_t *out) {
    OEA_REQUIRE(unique_id_128 && out && nranks >= 1 && rank >= 0 && rank < nranks, "arguments");
    Rccl *r = rccl();
    if (!r) { oea::set_error("RCCL (librccl.so.1) cannot be loaded: %s", dlerror()); return OEA_EUNSUPPORTED; }
    ncclUniqueId id;
    memcpy(&id, unique_id_128, sizeof(id));
    ncclComm_t c = nullptr;
    OEA_CHECK_RCCL(r->CommInitRank(&c, nranks, id, rank));       // binds the calling thread's current HIP device
    *out = new oea_comm{c, rank, nranks, nullptr, nullptr, nullptr, false, {}};
    return OEA_OK;
}

int oea_comm_init_callbacks(int32_t rank, int32_t nranks, oea_comm_callback fn, void *user, oea_comm_t *out) {
    OEA_REQUIRE(fn && out && nranks >= 1 && rank >= 0 && rank < nranks, "arguments");
    *out = new oea_comm{nullptr, rank, nranks, fn, user, nullptr, false, {}};
    return OEA_OK;
}

int oea_comm_destroy(oea_comm_t c) {
    if (!c) return OEA_OK;
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    if (!c->fn) OEA_CHECK_RCCL(rccl()->CommDestroy(c->comm));
    delete c;
    return OEA_OK;
}

int32_t oea_comm_rank(oea_comm_t c) { return c ? c->rank : -1; }
int32_t oea_comm_size(oea_comm_t c) { return c ? c->nranks : 0; }

#define OEA_CALLBACK(op, send, recv, count, dtype)                                                                      \
    if (c->fn) {                                                                                                        \
        const int _rc = c->fn(c->user, op, send, recv, (int64_t)(count), dtype, stream);                                \
        if (_rc != 0) { oea::set_error("%s:%d: the collective callback returned %d", __FILE__, __LINE__, _rc); return OEA_EHIP; } \
        return OEA_OK;                                                                                                  \
    }

int oea_comm_allgather(oea_comm_t c, const void *send, void *recv, int64_t n_per_rank, int32_t dtype, void *stream) {
    OEA_REQUIRE(c && send && recv && n_per_rank >= 0 && dtype >= OEA_COMM_F32 && dtype <= OEA_COMM_I64, "arguments");
    OEA_CALLBACK(OEA_COMM_ALLGATHER, send, recv, n_per_rank, dtype)
    OEA_CHECK_RCCL(rccl()->AllGather(send, recv, (size_t)n_per_rank, nccl_dtype(dtype), c->comm, oea::as_stream(stream)));
    return OEA_OK;
}

int oea_comm_reduce_scatter(oea_comm_t c, const void *send, void *recv, int64_t n_per_rank, int32_t dtype, void *stream) {
    OEA_REQUIRE(c && send && recv && n_per_rank >= 0 && dtype >= OEA_COMM_F32 && dtype <= OEA_COMM_I64, "arguments");
    OEA_CALLBACK(OEA_COMM_REDUCE_SCATTER, send, recv, n_per_rank, dtype)
    OEA_CHECK_RCCL(rccl()->ReduceScatter(send, recv, (size_t)n_per_rank, nccl_dtype(dtype), ncclSum, c->comm, oea::as_stream(stream)));
    return OEA_OK;
}

int oea_comm_allreduce(oea_comm_t c, void *buf, int64_t n, int32_t dtype, void *stream) {
    OEA_REQUIRE(c && buf && n >= 0 && dtype >= OEA_COMM_F32 && dtype <= OEA_COMM_I64, "arguments");
    OEA_CALLBACK(OEA_COMM_ALLREDUCE, buf, buf, n, dtype)
    OEA_CHECK_RCCL(rccl()->AllReduce(buf, buf, (size_t)n, nccl_dtype(dtype), ncclSum, c->comm, oea::as_stream(stream)));
    return OEA_OK;
}
#undef OEA_CALLBACK

int oea_comm_set_alltoallv(oea_comm_t c, oea_comm_alltoallv_callback fn) {
    OEA_REQUIRE(c && c->fn, "a communicator made by oea_comm_init_callbacks");
    c->a2a = fn;
    return OEA_OK;
}

// peer p gets send[send_displs[p] .. + send_counts[p]) and recv[recv_displs[p] ..] takes recv_counts[p] elements from p (host arrays,
// elements of `dtype`).  RCCL: one group of ncclSend / ncclRecv pairs (the halo exchange of the partitioned step: ~24 MB per rank in
// 2 (G - 1) messages of ~1.5 MB each at G = 8 -- point-to-point xGMI links carry them concurrently)
int oea_comm_alltoallv(oea_comm_t c, const void *send, const int64_t *send_counts, const int64_t *send_displs, void *recv,
                       const int64_t *recv_counts, const int64_t *recv_displs, int32_t dtype, void *stream) {
    OEA_REQUIRE(c && send && recv && send_counts && send_displs && recv_counts && recv_displs, "null pointer");
    OEA_REQUIRE(dtype >= OEA_COMM_F32 && dtype <= OEA_COMM_I64, "dtype");
    if (c->fn) {
        OEA_REQUIRE(c->a2a, "callback communicator without an all-to-all (oea_comm_set_alltoallv)");
        const int rc = c->a2a(c->user, send, send_counts, send_displs, recv, recv_counts, recv_displs, dtype, stream);
        if (rc != 0) { oea::set_error("%s:%d: the all-to-all callback returned %d", __FILE__, __LINE__, rc); return OEA_EHIP; }
        return OEA_OK;
    }
    const size_t es = dtype == OEA_COMM_F32 ? 4 : 8;
    Rccl *r = rccl();
    OEA_CHECK_RCCL(r->GroupStart());
    for (int p = 0; p < c->nranks; ++p) {
        if (send_counts[p] > 0)
            OEA_CHECK_RCCL(r->Send(static_cast<const char *>(send) + (size_t)send_displs[p] * es, (size_t)send_counts[p], nccl_dtype(dtype), p, c->comm,
                                   oea::as_stream(stream)));
        if (recv_counts[p] > 0)
            OEA_CHECK_RCCL(r->Recv(static_cast<char *>(recv) + (size_t)recv_displs[p] * es, (size_t)recv_counts[p], nccl_dtype(dtype), p, c->comm,
                                   oea::as_stream(stream)));
    }
    OEA_CHECK_RCCL(r->GroupEnd());
    return OEA_OK;
}

int oea_allgather_rows(oea_comm_t c, const float *send, float *recv, int64_t rows_per_rank, int32_t ld, void *stream) {
    OEA_REQUIRE(ld > 0, "arguments");
    return oea_comm_allgather(c, send, recv, rows_per_rank * ld, OEA_COMM_F32, stream);
}
int oea_comm_reduce_scatter_f32(oea_comm_t c, const float *send, float *recv, int64_t n_per_rank, void *stream) {
    return oea_comm_reduce_scatter(c, send, recv, n_per_rank, OEA_COMM_F32, stream);
}
int oea_allreduce_f32(oea_comm_t c, float *buf, int64_t n, void *stream) { return oea_comm_allreduce(c, buf, n, OEA_COMM_F32, stream); }
int oea_allreduce_f64(oea_comm_t c, double *buf, int64_t n, void *stream) { return oea_comm_allreduce(c, buf, n, OEA_COMM_F64, stream); }
int oea_allreduce_i64(oea_comm_t c, int64_t *buf, int64_t n, void *stream) { return oea_comm_allreduce(c, buf, n, OEA_COMM_I64, stream); }

// ---- phase profile of the one-call partitioned epoch -----------------------------------------------------------------
int oea_comm_profile_begin(oea_comm_t c) {
    OEA_REQUIRE(c, "null communicator");
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    c->events.clear();
    c->profiling = true;
    return OEA_OK;
}

int oea_comm_profile_end(oea_comm_t c, double *phase_ms, int32_t *steps) {
    OEA_REQUIRE(c && phase_ms && steps, "null pointer");
    c->profiling = false;
    const size_t per = OEA_COMM_PHASES + 1, n = c->events.size() / per;
    for (int p = 0; p < OEA_COMM_PHASES; ++p) phase_ms[p] = 0.0;
    if (n) OEA_CHECK_HIP(hipEventSynchronize(c->events[n * per - 1]));
    for (size_t s_ = 0; s_ < n; ++s_)
        for (int p = 0; p < OEA_COMM_PHASES; ++p) {
            float ms = 0.f;
            OEA_CHECK_HIP(hipEventElapsedTime(&ms, c->events[s_ * per + p], c->events[s_ * per + p + 1]));
            phase_ms[p] += ms;
        }
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    c->events.clear();
    *steps = (int32_t)n;
    return OEA_OK;
}

}  // extern "C"

// boundary `i` (0 .. OEA_COMM_PHASES) of the current step of oea_triple_epoch_range_comm (triple_step.hip)
namespace oea {
int comm_phase_mark(oea_comm_t c, hipStream_t st) {
    if (!c || !c->profiling) return OEA_OK;
    hipEvent_t e;
    OEA_CHECK_HIP(hipEventCreate(&e));
    OEA_CHECK_HIP(hipEventRecord(e, st));
    c->events.push_back(e);
    return OEA_OK;
}
}  // namespace oea
