// comm.hip -- the ONE exchange of the path: the per-symbol summary rows [n_local][8] of every rank's shard to every
// rank, by RCCL over xGMI on the context's stream.  Symbols are split statically -- rank r of G owns [floor(N r / G),
// floor(N (r + 1) / G)) -- and nothing else of the hot path communicates.  RCCL is bound at run time (dlopen) on the first
// pq_comm_* call: the single-GPU library has no link-time dependency on it, and a host process that already carries an RCCL
// (PyTorch bundles one) shares that copy instead of loading a second one.
#include "pq_dev.h"
#include <dlfcn.h>
#include <rccl/rccl.h> // types and enums only

namespace {
struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
pq_status rccl_load() {
    if (g_rccl.so) return PQ_OK;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void *so = nullptr;
    for (const char *n : names) {
        so = dlopen(n, RTLD_NOW | RTLD_NOLOAD); // a copy the process already holds (PyTorch's) comes first
        if (so) break;
    }
    for (size_t k = 0; !so && k < sizeof names / sizeof *names; k++) so = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
    if (!so) { pq_set_error("RCCL is not loadable (librccl.so): %s", dlerror()); return PQ_ERR_UNSUPPORTED; }
    Rccl r;
    r.so = so;
#define PQ_SYM(field, name)                                                                     \
    *reinterpret_cast<void **>(&r.field) = dlsym(so, name);                                     \
    if (!r.field) { pq_set_error("RCCL symbol %s not found", name); dlclose(so); return PQ_ERR_UNSUPPORTED; }
    PQ_SYM(GetUniqueId, "ncclGetUniqueId")
    PQ_SYM(CommInitRank, "ncclCommInitRank")
    PQ_SYM(CommDestroy, "ncclCommDestroy")
    PQ_SYM(GroupStart, "ncclGroupStart")
    PQ_SYM(GroupEnd, "ncclGroupEnd")
    PQ_SYM(AllGather, "ncclAllGather")
    PQ_SYM(Broadcast, "ncclBroadcast")
    PQ_SYM(GetErrorString, "ncclGetErrorString")
#undef PQ_SYM
    g_rccl = r;
    return PQ_OK;
}
#define PQ_NCCL_TRY(expr)                                                                                        \
    do {                                                                                                         \
        ncclResult_t r__ = (expr);                                                                               \
        if (r__ != ncclSuccess) { pq_set_error("%s failed: %s", #expr, g_rccl.GetErrorString(r__)); return PQ_ERR_HIP; } \
    } while (0)
} // namespace

extern "C" {

pq_status pq_comm_unique_id(void *id128) {
    PQ_REQUIRE(id128, "pq_comm_unique_id: null pointer");
    PQ_TRY(rccl_load());
    static_assert(sizeof(ncclUniqueId) == PQ_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    PQ_NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return PQ_OK;
}

pq_status pq_comm_init(pq_ctx *ctx, int32_t rank, int32_t world, const void *id128) {
    PQ_REQUIRE(ctx && id128, "pq_comm_init: null pointer");
    PQ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "pq_comm_init: need 0 <= rank < world");
    PQ_REQUIRE(!ctx->comm, "pq_comm_init: the context already has a communicator");
    PQ_TRY(rccl_load());
    PQ_HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    PQ_NCCL_TRY(g_rccl.CommInitRank(&comm, world, id, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return PQ_OK;
}
// The side stream of the overlapped exchange is created on first use, not with the communicator: the runtime multiplexes streams onto
// four hardware queues in creation order, and a stream that exists before a suite's three side streams pushes two of the step's chains
// onto one queue (measured: 5.0 instead of 3.9 ms per step with an idle extra stream created first).  Highest priority, so that its
// (tiny) kernels take the first free wave slots of a chip that the next step's grids are filling.
static pq_status comm_stream_make(pq_ctx *ctx) {
    if (ctx->comm_stream) return PQ_OK;
    int prio_lo = 0, prio_hi = 0;
    PQ_HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    // The stream runs at the DEFAULT priority.  Round 5 created it at the highest one ("the exchange is latency-critical") and measured
    // 136 us per 625-symbol backtest step with the exchange overlapped against 52.5 in series -- which it booked as "a cross-stream
    // dependency costs ~40 us on this runtime".  It does not: scripts/ubench/xstream.hip, bare HIP, puts a one-way dependency between
    // two streams at 4.7 us per step (10 us per hop when the result gates the next step).  What cost 85 us was the PRIORITY: a kernel
    // arriving on a higher-priority queue while the step's kernel holds the chip makes the hardware save and restore running waves.
    // scripts/ubench/gather_cabi.cpp (C, no torch), 625 symbols: 139 us per step at the highest priority, 60.7 at the default one
    // (kernel only 50.9, in series 52.5).  PQ_COMM_PRIO=high restores the old stream for A/B runs.
    const char *pe = getenv("PQ_COMM_PRIO");
    if (pe && !strcmp(pe, "high")) PQ_HIP_TRY(hipStreamCreateWithPriority(&ctx->comm_stream, hipStreamNonBlocking, prio_hi));
    else PQ_HIP_TRY(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
        PQ_HIP_TRY(hipEventCreateWithFlags(&ctx->comm_ev_in[k], hipEventDisableTiming));
        PQ_HIP_TRY(hipEventCreateWithFlags(&ctx->comm_ev_done[k], hipEventDisableTiming));
        ctx->comm_pending[k] = false;
    }
    return PQ_OK;
}

pq_status pq_comm_destroy(pq_ctx *ctx) {
    PQ_REQUIRE(ctx, "pq_comm_destroy: null pointer");
    if (!ctx->comm) return PQ_OK;
    PQ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) PQ_HIP_TRY(hipStreamSynchronize(ctx->comm_stream));
    PQ_NCCL_TRY(g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(ctx->comm)));
    ctx->comm = nullptr;
    ctx->comm_world = 0;
    for (int k = 0; k < 2; k++) {
        if (ctx->comm_ev_in[k]) { (void)hipEventDestroy(ctx->comm_ev_in[k]); ctx->comm_ev_in[k] = nullptr; }
        if (ctx->comm_ev_done[k]) { (void)hipEventDestroy(ctx->comm_ev_done[k]); ctx->comm_ev_done[k] = nullptr; }
        ctx->comm_pending[k] = false;
    }
    if (ctx->comm_stream) { (void)hipStreamDestroy(ctx->comm_stream); ctx->comm_stream = nullptr; }
    return PQ_OK;
}

pq_status pq_shard_range(int64_t n_symbols, int32_t rank, int32_t world, int64_t *lo, int64_t *hi) {
    PQ_REQUIRE(lo && hi && n_symbols >= 0 && world >= 1 && rank >= 0 && rank < world, "pq_shard_range: bad argument");
    *lo = (int64_t)((__int128)n_symbols * rank / world);
    *hi = (int64_t)((__int128)n_symbols * (rank + 1) / world);
    return PQ_OK;
}

static pq_status gather_on(pq_ctx *ctx, const double *local, int64_t n_symbols, double *all, hipStream_t stream);
pq_status pq_gather_summaries(pq_ctx *ctx, const double *local, int64_t n_symbols, double *all) {
    PQ_REQUIRE(ctx && all, "pq_gather_summaries: null pointer");
    PQ_REQUIRE(n_symbols >= 0, "pq_gather_summaries: n_symbols < 0");
    PQ_REQUIRE(ctx->comm, "pq_gather_summaries: call pq_comm_init first");
    PQ_REQUIRE(!ctx->rec, "pq_gather_summaries cannot be recorded into a suite (call it after pq_suite_run)");
    return gather_on(ctx, local, n_symbols, all, ctx->stream);
}
// The same exchange OFF the step's critical path.  _begin: everything enqueued on the context's stream so far (the step that produced
// `local`) is marked with an event, the communicator's own stream waits for it and takes the collective; the call returns at once and
// the context's stream goes on with the next step.  _end: the context's stream waits (on the device: the host does not block) for that
// slot's collective, after which `all` may be read and `local` rewritten by work enqueued on the context's stream.  Two slots: a caller
// that double-buffers `local` / `all` keeps one exchange in flight while the next step computes.
pq_status pq_gather_summaries_begin(pq_ctx *ctx, const double *local, int64_t n_symbols, double *all, int32_t slot) {
    PQ_REQUIRE(ctx && all, "pq_gather_summaries_begin: null pointer");
    PQ_REQUIRE(n_symbols >= 0, "pq_gather_summaries_begin: n_symbols < 0");
    PQ_REQUIRE(slot == 0 || slot == 1, "pq_gather_summaries_begin: slot must be 0 or 1");
    PQ_REQUIRE(ctx->comm, "pq_gather_summaries_begin: call pq_comm_init first");
    PQ_REQUIRE(!ctx->rec, "pq_gather_summaries_begin cannot be recorded into a suite (call it after pq_suite_run)");
    PQ_REQUIRE(!ctx->comm_pending[slot], "pq_gather_summaries_begin: this slot has an exchange in flight (call pq_gather_summaries_end first)");
    PQ_HIP_TRY(hipSetDevice(ctx->device));
    PQ_TRY(comm_stream_make(ctx));
    PQ_HIP_TRY(hipEventRecord(ctx->comm_ev_in[slot], ctx->stream));
    PQ_HIP_TRY(hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ev_in[slot], 0));
    PQ_TRY(gather_on(ctx, local, n_symbols, all, ctx->comm_stream));
    PQ_HIP_TRY(hipEventRecord(ctx->comm_ev_done[slot], ctx->comm_stream));
    ctx->comm_pending[slot] = true;
    return PQ_OK;
}
pq_status pq_gather_summaries_end(pq_ctx *ctx, int32_t slot) {
    PQ_REQUIRE(ctx, "pq_gather_summaries_end: null pointer");
    PQ_REQUIRE(slot == 0 || slot == 1, "pq_gather_summaries_end: slot must be 0 or 1");
    if (!ctx->comm_pending[slot]) return PQ_OK;
    PQ_HIP_TRY(hipSetDevice(ctx->device));
    // steady state: the exchange of two steps ago completed long ago -- a host-side query then saves the cross-stream wait (a barrier
    // packet on the step's queue: ~10 us, scripts/ubench/xstream.hip)
    const hipError_t q = hipEventQuery(ctx->comm_ev_done[slot]);
    if (q != hipSuccess) {
        if (q != hipErrorNotReady) PQ_HIP_TRY(q);
        (void)hipGetLastError(); // hipErrorNotReady is sticky in hipGetLastError
        PQ_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->comm_ev_done[slot], 0));
    }
    ctx->comm_pending[slot] = false;
    return PQ_OK;
}
// host-side wait for everything on the communicator's own stream (a caller that wants to bound a first exchange from a helper thread:
// a collective that never returns then blocks that thread, not the context's stream)
pq_status pq_comm_sync(pq_ctx *ctx) {
    PQ_REQUIRE(ctx, "pq_comm_sync: null pointer");
    if (!ctx->comm_stream) return PQ_OK;
    PQ_HIP_TRY(hipSetDevice(ctx->device));
    PQ_HIP_TRY(hipStreamSynchronize(ctx->comm_stream));
    return PQ_OK;
}
static pq_status gather_on(pq_ctx *ctx, const double *local, int64_t n_symbols, double *all, hipStream_t stream) {
    PQ_HIP_TRY(hipSetDevice(ctx->device));
    const int G = ctx->comm_world;
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(ctx->comm);
    int64_t lo, hi;
    PQ_TRY(pq_shard_range(n_symbols, ctx->comm_rank, G, &lo, &hi));
    PQ_REQUIRE(local || hi == lo, "pq_gather_summaries: null pointer");
    if (n_symbols % G == 0) { // equal shards: one all-gather
        if (n_symbols > 0)
            PQ_NCCL_TRY(g_rccl.AllGather(local, all, (size_t)(hi - lo) * PQ_SUMMARY_COLS, ncclFloat64, comm, stream));
        return PQ_OK;
    }
    // ragged shards: every rank's rows are broadcast into their place, as one group (one launch per peer, all links busy at once)
    if (hi > lo) PQ_HIP_TRY(hipMemcpyAsync(all + lo * PQ_SUMMARY_COLS, local, (size_t)(hi - lo) * PQ_SUMMARY_COLS * sizeof(double), hipMemcpyDeviceToDevice, stream));
    PQ_NCCL_TRY(g_rccl.GroupStart());
    ncclResult_t bad = ncclSuccess;   // a failing call inside the group must still be followed by GroupEnd (else the communicator stays mid-group)
    for (int r = 0; r < G && bad == ncclSuccess; r++) {
        int64_t a, b;
        (void)pq_shard_range(n_symbols, r, G, &a, &b);   // (arguments validated above)
        if (b == a) continue;
        double *dst = all + a * PQ_SUMMARY_COLS;
        bad = g_rccl.Broadcast(dst, dst, (size_t)(b - a) * PQ_SUMMARY_COLS, ncclFloat64, r, comm, stream);
    }
    const ncclResult_t end = g_rccl.GroupEnd();
    PQ_NCCL_TRY(bad);
    PQ_NCCL_TRY(end);
    return PQ_OK;
}

} // extern "C"
