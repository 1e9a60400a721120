// Exchange layer of a sharded run (not part of the public ABI).  The scoring loop has three exchange points
// (include/cellector_ffi.h): PASS1 once at load (sum), NORM per iteration (all-gather of equal cell slices), LOCUS per
// iteration (sum).  Two transports behind one interface:
//   * RCCL over xGMI (one communicator rank per shard; ncclCommInitAll inside one process, ncclCommInitRank across
//     processes).  librccl is opened lazily: a single-GPU run never loads it.
//   * a same-process group for logical shards that SHARE a device (RCCL refuses duplicate devices): host barrier +
//     device-side sums in rank order.  Tests and one-GPU rehearsals of the sharded path use it.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <vector>

struct cellector_ctx;

#define CELLECTOR_MAX_SHARDS 16

// shards of one process that exchange through device memory they can all reach
struct LocalGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    const void *bufs[CELLECTOR_MAX_SHARDS] = {};
    bool failed = false;  // a shard gave up (error path): the others must not wait for it
    bool barrier();       // false: the group has failed
    void fail();
};

struct Comm {
    int n = 1, rank = 0;
    // the cell ranges of the ranks: rank r owns [bounds[r], bounds[r + 1]).  Without explicit bounds (has_bounds false) the
    // canonical split: n equal contiguous ranges of ceil(cells / n).  Explicit bounds come from cellector_set_partition or from
    // the nnz-balancing ingest of a multi-device ctx (multi.cpp): cells differ in coverage by orders of magnitude, and the
    // slowest shard sets the iteration.
    bool has_bounds = false;
    uint64_t bounds[CELLECTOR_MAX_SHARDS + 1] = {};
    void *nccl = nullptr;         // ncclComm_t of this shard (RCCL transport)
    LocalGroup *local = nullptr;  // same-process transport (shared by the group's shards; owned by the root ctx)
    double *tmp = nullptr;        // local transport: scratch for the sums, sized on first use
    uint64_t tmp_n = 0;
};

// a communicator is attached (n = 1 with one: the single-rank self-test of the RCCL plumbing)
static inline bool comm_active(const Comm &m) { return m.n > 1 || m.nccl != nullptr; }
// cells per rank of the canonical contiguous split (the NORM all-gather needs equal slots)
static inline uint64_t comm_cells_per_rank(uint64_t total_cells, int n) { return (total_cells + (uint64_t)n - 1) / (uint64_t)n; }

// rank r's cell range of a run over total_cells cells
static inline void comm_range(const Comm &m, uint64_t total_cells, int r, uint64_t *b, uint64_t *e)
{
    if (m.has_bounds) {
        *b = m.bounds[r] < total_cells ? m.bounds[r] : total_cells;
        *e = m.bounds[r + 1] < total_cells ? m.bounds[r + 1] : total_cells;
    } else {
        const uint64_t per = comm_cells_per_rank(total_cells, m.n);
        *b = (uint64_t)r * per < total_cells ? (uint64_t)r * per : total_cells;
        *e = *b + per < total_cells ? *b + per : total_cells;
    }
}
// n + 1 boundaries that give every rank about the same number of ENTRIES: cut where the running sum of entries_per_cell
// crosses k / n of the total, on a multiple of `align` cells when the ranges are large (a shard's tiles are 1024 cells tall)
void comm_balanced_bounds(const uint32_t *entries_per_cell, uint64_t total_cells, int n, uint64_t *bounds /*[n + 1]*/);

// Exchange point 2 by digit histograms or by gathering the keys?  Six levels (kernel + all-reduce of 48 KB) cost ~0.08 ms of
// kernels at 10^6 keys per shard (measured) plus six small-message latencies, whatever the size of the run; the gathering form
// moves 8 B per cell of the WHOLE run to every shard and selects over all of them there (0.054 ms per 10^6 keys, measured).
// BASELINE's 10^6 cells are 8 MB: one all-gather and one select are then the cheaper side; from 4*10^6 cells (32 MB) on the
// histograms are.  (An estimate: no multi-GPU box was available to measure the collectives.)
static inline bool comm_sharded_select(const Comm &m, int option, uint64_t total_cells)
{
    return option >= 0 ? option != 0 : (m.n > 2 && total_cells >= (4ull << 20)) || m.n == 1 /* self-test: run the new path */;
}

int comm_status_ok();  // CELLECTOR_OK as int (keeps this header free of the public one)
// sum over all shards, in place, on the ctx's stream; every shard ends with the same bits
int comm_allreduce_sum(cellector_ctx *c, double *buf, uint64_t count);
int comm_allreduce_sum_u32(cellector_ctx *c, uint32_t *buf, uint64_t count);  // (counts: the order statistics of a sharded run)
// every shard contributes its cells' slice of buf (global cell order, comm_range) and receives the others' (buf holds
// max(total_cells, n * ceil(total_cells / n)) doubles)
int comm_allgather_cells(cellector_ctx *c, double *buf, uint64_t total_cells);
// RCCL: unique id for ncclCommInitRank, communicator set-up / tear-down
int comm_rccl_unique_id(void *out128, const char **err);
int comm_rccl_init_rank(cellector_ctx *c, const void *id128, int n, int rank);
int comm_rccl_init_all(cellector_ctx **shards, int n, const int *devices);
void comm_rccl_abort(cellector_ctx *c);  // frees the communicator: drop c->comm.nccl afterwards
void comm_destroy(cellector_ctx *c);
