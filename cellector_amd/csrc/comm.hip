// Exchange layer of a sharded run: RCCL over xGMI, or device-side sums between logical shards that share a GPU (comm.h).
#include <dlfcn.h>

#include <algorithm>

#include <rccl/rccl.h>

#include "comm.h"
#include "ctx.h"

int comm_status_ok() { return (int)CELLECTOR_OK; }

// ---- librccl, opened on first use -------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

bool rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.so) return true;
    // (a process that already holds a copy — PyTorch-ROCm bundles one under the same soname — gets that copy)
    void *so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!so) {
        g_rccl.why = dlerror() ? dlerror() : "librccl.so.1 not found";
        return false;
    }
    bool ok = true;
#define SYM(field, name)                                                         \
    do {                                                                         \
        *(void **)(&g_rccl.field) = dlsym(so, name);                             \
        if (!g_rccl.field) { ok = false; g_rccl.why = "librccl lacks " name; }   \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(CommAbort, "ncclCommAbort");
    SYM(AllReduce, "ncclAllReduce");
    SYM(AllGather, "ncclAllGather");
    SYM(Broadcast, "ncclBroadcast");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    if (!ok) {
        dlclose(so);
        return false;
    }
    g_rccl.so = so;
    return true;
}

#define NCCLCHK(c, expr)                                                                                      \
    do {                                                                                                      \
        ncclResult_t r__ = (expr);                                                                            \
        if (r__ != ncclSuccess)                                                                               \
            return (int)ctx_fail((c), CELLECTOR_ECOMM, "%s failed: %s", #expr, g_rccl.GetErrorString(r__));   \
    } while (0)
}  // namespace

int comm_rccl_unique_id(void *out128, const char **err)
{
    static_assert(sizeof(ncclUniqueId) == 128, "the C ABI passes the id as 128 opaque bytes");
    if (!rccl_load()) {
        if (err) *err = g_rccl.why.c_str();
        return (int)CELLECTOR_ECOMM;
    }
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) {
        if (err) *err = g_rccl.GetErrorString(r);
        return (int)CELLECTOR_ECOMM;
    }
    memcpy(out128, &id, sizeof id);
    return (int)CELLECTOR_OK;
}

int comm_rccl_init_rank(cellector_ctx *c, const void *id128, int n, int rank)
{
    if (!rccl_load()) return (int)ctx_fail(c, CELLECTOR_ECOMM, "cannot load RCCL: %s", g_rccl.why.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    NCCLCHK(c, g_rccl.CommInitRank(&comm, n, id, rank));
    c->comm.n = n;
    c->comm.rank = rank;
    c->comm.nccl = comm;
    return (int)CELLECTOR_OK;
}

int comm_rccl_init_all(cellector_ctx **shards, int n, const int *devices)
{
    cellector_ctx *c0 = shards[0];
    if (!rccl_load()) return (int)ctx_fail(c0, CELLECTOR_ECOMM, "cannot load RCCL: %s", g_rccl.why.c_str());
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    NCCLCHK(c0, g_rccl.CommInitAll(comms.data(), n, devices));
    for (int r = 0; r < n; r++) {
        shards[r]->comm.n = n;
        shards[r]->comm.rank = r;
        shards[r]->comm.nccl = comms[(size_t)r];
    }
    return (int)CELLECTOR_OK;
}

// A shard of a multi-device ctx has failed: abort every shard's communicator, so that the peers' pending and future
// collectives return instead of waiting for the rank that will never join them.  ncclCommAbort frees the communicator: the
// caller drops the handles once every worker is back (multi.cpp, run_all).
void comm_rccl_abort(cellector_ctx *c)
{
    if (c->comm.nccl && g_rccl.so) (void)g_rccl.CommAbort((ncclComm_t)c->comm.nccl);
}

void comm_destroy(cellector_ctx *c)
{
    if (c->comm.nccl && g_rccl.so) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)g_rccl.CommDestroy((ncclComm_t)c->comm.nccl);
    }
    c->comm.nccl = nullptr;
    dev_free(c->comm.tmp);
    c->comm.tmp_n = 0;
    c->comm.local = nullptr;  // (owned by the root)
    c->comm.n = 1;
    c->comm.rank = 0;
}

// ---- same-process transport ----------------------------------------------------------------------------------------------
bool LocalGroup::barrier()
{
    std::unique_lock<std::mutex> lk(mu);
    if (failed) return false;
    const uint64_t gen = generation;
    if (++arrived == n) {
        arrived = 0;
        generation++;
        cv.notify_all();
    } else {
        cv.wait(lk, [&] { return generation != gen || failed; });
    }
    return !failed;
}
void LocalGroup::fail()
{
    std::lock_guard<std::mutex> lk(mu);
    failed = true;
    cv.notify_all();
}
#define BARRIER(c, g)                                                                                     \
    do {                                                                                                  \
        if (!(g)->barrier()) return (int)ctx_fail((c), CELLECTOR_ECOMM, "another shard of this ctx failed"); \
    } while (0)

struct peer_bufs_t { const void *p[CELLECTOR_MAX_SHARDS]; };
// out[i] = sum over the shards' buffers in rank order: every shard computes the same bits
template <typename T>
__global__ __launch_bounds__(256) void k_sum_peers(uint64_t n, int n_peers, peer_bufs_t peers, T *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    T s = static_cast<const T *>(peers.p[0])[i];
    for (int r = 1; r < n_peers; r++) s += static_cast<const T *>(peers.p[r])[i];
    out[i] = s;
}

template <typename T>
static int local_allreduce(cellector_ctx *c, T *buf, uint64_t count)
{
    LocalGroup *g = c->comm.local;
    const uint64_t need = (count * sizeof(T) + 7) / 8;  // (the scratch is kept in doubles)
    if (c->comm.tmp_n < need) {
        dev_free(c->comm.tmp);
        c->comm.tmp_n = 0;
        if (dev_alloc(c, &c->comm.tmp, need) != CELLECTOR_OK) return (int)CELLECTOR_ENOMEM;
        c->comm.tmp_n = need;
    }
    T *tmp = reinterpret_cast<T *>(c->comm.tmp);
    HIPCHK(c, hipStreamSynchronize(c->stream));  // this shard's contribution is complete
    g->bufs[c->comm.rank] = buf;
    BARRIER(c, g);
    peer_bufs_t peers;
    for (int r = 0; r < g->n; r++) peers.p[r] = g->bufs[r];
    if (count)
        hipLaunchKernelGGL(k_sum_peers<T>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, count, g->n, peers, tmp);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    BARRIER(c, g);  // everybody has read everybody's contribution
    if (count) HIPCHK(c, hipMemcpyAsync(buf, tmp, count * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    return (int)CELLECTOR_OK;
}

static int local_allgather(cellector_ctx *c, double *buf, uint64_t total_cells)
{
    LocalGroup *g = c->comm.local;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    g->bufs[c->comm.rank] = buf;
    BARRIER(c, g);
    for (int r = 0; r < g->n; r++) {
        uint64_t b, e;
        comm_range(c->comm, total_cells, r, &b, &e);
        if (r != c->comm.rank && e > b)
            HIPCHK(c, hipMemcpyAsync(buf + b, static_cast<const double *>(g->bufs[r]) + b, (e - b) * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    BARRIER(c, g);  // nobody overwrites a slice another shard still copies
    return (int)CELLECTOR_OK;
}

// ---- the interface ---------------------------------------------------------------------------------------------------------
int comm_allreduce_sum(cellector_ctx *c, double *buf, uint64_t count)
{
    if (!comm_active(c->comm)) return (int)CELLECTOR_OK;
    if (c->comm.local) return local_allreduce(c, buf, count);
    if (!c->comm.nccl) return (int)ctx_fail(c, CELLECTOR_ECOMM, "sharded ctx without a communicator");
    NCCLCHK(c, g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)c->comm.nccl, c->stream));
    return (int)CELLECTOR_OK;
}

int comm_allreduce_sum_u32(cellector_ctx *c, uint32_t *buf, uint64_t count)
{
    if (!comm_active(c->comm)) return (int)CELLECTOR_OK;
    if (c->comm.local) return local_allreduce(c, buf, count);
    if (!c->comm.nccl) return (int)ctx_fail(c, CELLECTOR_ECOMM, "sharded ctx without a communicator");
    NCCLCHK(c, g_rccl.AllReduce(buf, buf, (size_t)count, ncclUint32, ncclSum, (ncclComm_t)c->comm.nccl, c->stream));
    return (int)CELLECTOR_OK;
}

int comm_allgather_cells(cellector_ctx *c, double *buf, uint64_t total_cells)
{
    if (!comm_active(c->comm)) return (int)CELLECTOR_OK;
    if (c->comm.local) return local_allgather(c, buf, total_cells);
    if (!c->comm.nccl) return (int)ctx_fail(c, CELLECTOR_ECOMM, "sharded ctx without a communicator");
    if (!c->comm.has_bounds) {
        // equal slots: one in-place all-gather (the send buffer is this rank's slot of the receive buffer)
        const uint64_t per = comm_cells_per_rank(total_cells, c->comm.n);
        NCCLCHK(c, g_rccl.AllGather(buf + (uint64_t)c->comm.rank * per, buf, (size_t)per, ncclDouble, (ncclComm_t)c->comm.nccl, c->stream));
        return (int)CELLECTOR_OK;
    }
    // ranges of different length (nnz-balanced partition): every rank broadcasts its slice in place, as ONE group
    NCCLCHK(c, g_rccl.GroupStart());
    for (int r = 0; r < c->comm.n; r++) {
        uint64_t b, e;
        comm_range(c->comm, total_cells, r, &b, &e);
        if (e > b) NCCLCHK(c, g_rccl.Broadcast(buf + b, buf + b, (size_t)(e - b), ncclDouble, r, (ncclComm_t)c->comm.nccl, c->stream));
    }
    NCCLCHK(c, g_rccl.GroupEnd());
    return (int)CELLECTOR_OK;
}

void comm_balanced_bounds(const uint32_t *entries_per_cell, uint64_t total_cells, int n, uint64_t *bounds)
{
    uint64_t total = 0;
    for (uint64_t i = 0; i < total_cells; i++) total += entries_per_cell[i];
    // (big ranges: whole 1024-cell blocks per shard — the rounding moves at most 512 of >= 64k cells; small ones: any cell)
    const uint64_t align = total_cells / (uint64_t)n >= 65536 ? 1024 : 1;
    bounds[0] = 0;
    uint64_t run = 0, cell = 0;
    for (int k = 1; k < n; k++) {
        // first cell index at which the running sum reaches k / n of the entries
        const uint64_t want = (uint64_t)(((__uint128_t)total * (uint64_t)k) / (uint64_t)n);
        while (cell < total_cells && run + entries_per_cell[cell] <= want) run += entries_per_cell[cell++];
        uint64_t cut = cell;
        if (align > 1) cut = std::min(total_cells, (cut + align / 2) / align * align);
        bounds[k] = std::max(cut, bounds[k - 1]);
    }
    bounds[n] = total_cells;
}
