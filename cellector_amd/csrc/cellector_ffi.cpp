// C-ABI layer of libcellector_hip.so (include/cellector_ffi.h): argument checking, state machine,
// host<->device copies and the host-side scalar arithmetic of the scoring loop (threshold, priors).
// All matrix work is in the HIP kernels (kernels_*.hip); there is no CPU fallback.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <algorithm>
#include <chrono>
#include <vector>
#include <unordered_map>
#include <mutex>

#include "ctx.h"
#include "multi.h"

cellector_status ctx_fail(const cellector_ctx *c, cellector_status s, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return s;
}

// ---- timing -----------------------------------------------------------------------------------------
void timer_begin(cellector_ctx *c, int which)
{
    if (!c->timing) return;
    // an event pair costs a few microseconds of idle queue (2 % of an iteration at 10^6 cells x 200k loci): level 2 keeps only the
    // pair the roofline figure needs, level 3 records that pair around every fourth launch only
    KernelTimer &t = c->timers[which];
    t.open = false;
    if (c->timing >= 2 && which != (c->engine == 2 ? CELLECTOR_K_TILE_LL : CELLECTOR_K_CELL_LL)) return;
    if (c->timing == 3 && (t.calls++ & 3u) != 0) return;
    hipEvent_t a = nullptr, b = nullptr;
    if (c->ev_pool.size() >= 2) {
        a = c->ev_pool.back(); c->ev_pool.pop_back();
        b = c->ev_pool.back(); c->ev_pool.pop_back();
    } else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        return;
    }
    t.start.push_back(a);
    t.stop.push_back(b);
    t.open = true;
    (void)hipEventRecord(a, c->stream);
}
// the event pair of a launch whose start and stop ride on the kernel's own dispatch (hipExtLaunchKernelGGL): no barrier packets
// in the queue, i.e. none of the idle time a recorded pair costs.  False: this launch is not timed.
bool timer_take(cellector_ctx *c, int which, hipEvent_t *a_out, hipEvent_t *b_out)
{
    if (!c->timing) return false;
    KernelTimer &t = c->timers[which];
    if (c->timing >= 2 && which != (c->engine == 2 ? CELLECTOR_K_TILE_LL : CELLECTOR_K_CELL_LL)) return false;
    if (c->timing == 3 && (t.calls++ & 3u) != 0) return false;
    hipEvent_t a = nullptr, b = nullptr;
    if (c->ev_pool.size() >= 2) {
        a = c->ev_pool.back(); c->ev_pool.pop_back();
        b = c->ev_pool.back(); c->ev_pool.pop_back();
    } else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        return false;
    }
    // (kept apart from the recorded pairs' open / close bookkeeping: both events belong to one dispatch)
    t.start.insert(t.start.begin(), a);
    t.stop.insert(t.stop.begin(), b);
    *a_out = a; *b_out = b;
    return true;
}
void timer_end(cellector_ctx *c, int which)
{
    KernelTimer &t = c->timers[which];
    if (!t.open || t.stop.empty()) return;  // (timer_begin recorded nothing for this launch)
    t.open = false;
    (void)hipEventRecord(t.stop.back(), c->stream);
}
void timer_collect(cellector_ctx *c)
{
    for (int k = 0; k < CELLECTOR_K_COUNT; k++) {
        KernelTimer &t = c->timers[k];
        // (a pair whose stop event is still to be recorded — the first half of a cell pass queued ahead by em_finish — stays)
        const size_t n_done = t.start.size() - (t.open && !t.start.empty() ? 1 : 0);
        for (size_t i = 0; i < n_done; i++) {
            float ms = 0.f;
            if (hipEventSynchronize(t.stop[i]) == hipSuccess &&
                hipEventElapsedTime(&ms, t.start[i], t.stop[i]) == hipSuccess) {
                t.total_ms += ms;
                t.launches++;
            } else {
                (void)hipGetLastError();  // (an event that was never recorded: not this call's caller's error)
            }
            c->ev_pool.push_back(t.start[i]);
            c->ev_pool.push_back(t.stop[i]);
        }
        t.start.erase(t.start.begin(), t.start.begin() + (long)n_done);
        t.stop.erase(t.stop.begin(), t.stop.begin() + (long)n_done);
    }
}

static void free_matrix(cellector_ctx *c)
{
    dev_free(c->coo_locus); dev_free(c->coo_cell); dev_free(c->coo_alt); dev_free(c->coo_ref);
    dev_free(c->csr_ptr); dev_free(c->csr_ent); dev_free(c->csc_ptr); dev_free(c->csc_ent);
    dev_free(c->locus_ids); dev_free(c->s_alt); dev_free(c->s_ref); dev_free(c->n_ent); dev_free(c->to_used);
    dev_free(c->ab); dev_free(c->ab6); dev_free(c->mask); dev_free(c->mask_next);
    dev_free(c->flags); dev_free(c->flags_new); dev_free(c->ll); dev_free(c->ell); dev_free(c->nloci);
    dev_free(c->post);
    tiled_free(c);
    if (c->own_pass1) dev_free(c->x_pass1);
    if (c->own_norm) dev_free(c->x_norm);
    if (c->own_locus) dev_free(c->x_locus);
    c->x_pass1 = c->x_norm = c->x_locus = nullptr;
    c->own_pass1 = c->own_norm = c->own_locus = true;
    c->n_pass1 = c->n_norm = c->n_locus = 0;
    c->coo_n = 0; c->L = c->nnz = c->nloc = 0;
    c->state = cellector_ctx::ST_EMPTY;
    c->em_phase = 0; c->iteration = 0; c->have_iter = false; c->n_excluded_global = 0;
    // nothing built ahead for the previous matrix survives a reload: the next em_begin must form alpha/beta and the
    // tables itself (em_finish leaves tables_prebuilt set; the new matrix' table buffers are fresh allocations)
    c->tables_prebuilt = false; c->prebuilt_expected = false; c->work_zeroed = false; c->ovf_locus_pending = false;
    c->cell_join_pending = false;
}

// the side stream gets the lowest priority the device offers: its kernels should only fill slots the main stream's
// kernels leave idle
static bool create_side_stream(hipStream_t *out)
{
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, least) == hipSuccess;
}

#define REQUIRE(c, cond, msg)                                        \
    do {                                                             \
        if (!(cond)) return ctx_fail((c), CELLECTOR_EINVAL, "%s", msg); \
    } while (0)
#define SETDEV(c) HIPCHK((c), hipSetDevice((c)->device))

// ---- caching layer under dev_alloc / dev_free (see ctx.h) -------------------------------------------------------
namespace {
struct DevBlock { void *p; size_t bytes; int device; };
std::mutex g_cache_mu;
std::vector<DevBlock> g_cache_free;                 // freed, still mapped
std::unordered_map<void *, DevBlock> g_cache_live;  // handed out by dev_cache_malloc
const size_t CACHE_MIN = 64ull << 20;               // smaller blocks go straight to the driver
}  // namespace

hipError_t dev_cache_malloc(void **p, size_t bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (bytes >= CACHE_MIN) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < g_cache_free.size(); i++) {
            const DevBlock &b = g_cache_free[i];
            // fits, is at most twice the request, and is the tightest such block
            if (b.device == dev && b.bytes >= bytes && b.bytes - bytes <= bytes &&
                (best == (size_t)-1 || b.bytes < g_cache_free[best].bytes))
                best = i;
        }
        if (best != (size_t)-1) {
            DevBlock b = g_cache_free[best];
            g_cache_free.erase(g_cache_free.begin() + (long)best);
            g_cache_live[b.p] = b;
            *p = b.p;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {  // out of memory: give the cached blocks back and try once more
        (void)hipGetLastError();
        dev_cache_trim();
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess && bytes >= CACHE_MIN) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        g_cache_live[*p] = DevBlock{*p, bytes, dev};
    }
    return e;
}

void dev_cache_free(void *p)
{
    if (!p) return;
    DevBlock blk{};
    bool cached = false;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_cache_live.find(p);
        if (it != g_cache_live.end()) {
            blk = it->second;
            g_cache_live.erase(it);
            cached = true;
        }
    }
    if (!cached) {
        (void)hipFree(p);
        return;
    }
    // hipFree would have waited for the block's device; a block handed out again must not still be in use either.  The wait is
    // for the BLOCK's device (a shard thread may free another shard's block) and happens outside the lock: other shards'
    // allocations do not queue behind it.
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != blk.device) (void)hipSetDevice(blk.device);
    (void)hipDeviceSynchronize();
    if (cur != blk.device) (void)hipSetDevice(cur);
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_cache_free.push_back(blk);
}

// a block nobody has used yet goes to the free list as it is (no device synchronisation needed)
void dev_cache_park(void *p, size_t bytes, int device)
{
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_cache_free.push_back(DevBlock{p, bytes, device});
}

void dev_cache_trim(int device)
{
    std::vector<DevBlock> blocks;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if (device < 0) blocks.swap(g_cache_free);
        else {  // only this device's blocks: the other shards of a multi-device ctx may still be building from theirs
            std::vector<DevBlock> keep;
            for (const DevBlock &b : g_cache_free) (b.device == device ? blocks : keep).push_back(b);
            g_cache_free.swap(keep);
        }
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const DevBlock &b : blocks) {
        (void)hipSetDevice(b.device);
        (void)hipFree(b.p);
    }
    (void)hipSetDevice(cur);
}

extern "C" {

const char *cellector_version(void) { return "cellector_amd 0.1 (gfx950)"; }

cellector_status cellector_create(cellector_ctx **out, int device_id)
{
    if (!out) return CELLECTOR_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return CELLECTOR_EDEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return CELLECTOR_EDEVICE;
    cellector_ctx *c = new (std::nothrow) cellector_ctx();
    if (!c) return CELLECTOR_ENOMEM;
    c->device = device_id;
    // ln(FCACHE[x]), x = 0..170: statrs' factorial cache, logs taken with the host libm like the reference
    double lf[LF_TABLE_N], f = 1.0;
    lf[0] = std::log(1.0);
    for (int i = 1; i < LF_TABLE_N; i++) {
        f *= (double)i;
        lf[i] = std::log(f);
    }
    bool ok = hipMalloc((void **)&c->lf, sizeof lf) == hipSuccess &&
              hipMemcpy(c->lf, lf, sizeof lf, hipMemcpyHostToDevice) == hipSuccess &&
              hipMalloc((void **)&c->d_counters, 8 * sizeof(uint32_t)) == hipSuccess &&
              hipMalloc((void **)&c->sel_hist, CELLECTOR_SEL_HIST_WORDS * sizeof(uint32_t)) == hipSuccess &&
              hipMemset(c->sel_hist, 0, CELLECTOR_SEL_HIST_WORDS * sizeof(uint32_t)) == hipSuccess &&  // k_sel_finish re-zeroes
              hipMalloc((void **)&c->sel_state, 4 * SEL_T * sizeof(uint64_t)) == hipSuccess &&
              hipMalloc((void **)&c->sel_out, 16 * sizeof(double)) == hipSuccess &&
              hipHostMalloc((void **)&c->h_sel, 32 * sizeof(double)) == hipSuccess &&
              (memset(c->h_sel, 0, 32 * sizeof(double)), true) &&
              hipHostGetDevicePointer((void **)&c->h_sum_dev, c->h_sel, 0) == hipSuccess &&
              create_side_stream(&c->side) &&
              hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_sum, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_tab, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        cellector_destroy(c);
        return CELLECTOR_EDEVICE;
    }
    *out = c;
    return CELLECTOR_OK;
}

cellector_status cellector_device_count(int *out)
{
    if (!out) return CELLECTOR_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *out = n;
    return CELLECTOR_OK;
}

cellector_status cellector_create_multi(cellector_ctx **out, const int *device_ids, int n_devices)
{
    if (!out || !device_ids || n_devices < 1 || n_devices > CELLECTOR_MAX_SHARDS) return CELLECTOR_EINVAL;
    *out = nullptr;
    if (n_devices == 1) return cellector_create(out, device_ids[0]);  // a plain single-shard ctx: nothing to exchange
    return multi_create(out, device_ids, n_devices);
}

cellector_status cellector_comm_unique_id(void *out_128_bytes)
{
    if (!out_128_bytes) return CELLECTOR_EINVAL;
    const char *why = nullptr;
    const int st = comm_rccl_unique_id(out_128_bytes, &why);
    if (st != CELLECTOR_OK) fprintf(stderr, "cellector_comm_unique_id: %s\n", why ? why : "RCCL error");
    return (cellector_status)st;
}

cellector_status cellector_comm_init_rank(cellector_ctx *c, const void *unique_id_128, int n_ranks, int rank)
{
    if (!c || !unique_id_128) return CELLECTOR_EINVAL;
    REQUIRE(c, !c->multi, "a multi-device ctx has its communicator already");
    REQUIRE(c, c->state == cellector_ctx::ST_EMPTY, "attach the communicator before the ingest");
    REQUIRE(c, !comm_active(c->comm), "ctx already has a communicator");
    REQUIRE(c, n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bad rank / rank count");
    // (a one-rank communicator is made only on request: CELLECTOR_COMM_SELFTEST runs the RCCL calls of the sharded path on one GPU)
    if (n_ranks == 1 && !getenv("CELLECTOR_COMM_SELFTEST")) return CELLECTOR_OK;
    c->norm_zero = false;  // the NORM slices are all-gathered inside the library
    return (cellector_status)comm_rccl_init_rank(c, unique_id_128, n_ranks, rank);
}

void cellector_destroy(cellector_ctx *c)
{
    if (!c) return;
    if (c->multi) { multi_destroy(c); return; }
    (void)hipSetDevice(c->device);
    if (c->stream && c->owns_stream) (void)hipStreamSynchronize(c->stream);
    comm_destroy(c);
    if (c->side) (void)hipStreamSynchronize(c->side);
    timer_collect(c);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    c->ev_pool.clear();
    free_matrix(c);
    dev_free(c->lf); dev_free(c->d_counters); dev_free(c->sel_hist); dev_free(c->sel_state); dev_free(c->sel_out);
    dev_free(c->sel_list); dev_free(c->seld_hist); dev_free(c->seld_state);
    if (c->h_sel) (void)hipHostFree(c->h_sel);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
    if (c->ev_sum) (void)hipEventDestroy(c->ev_sum);
    if (c->ev_tab) (void)hipEventDestroy(c->ev_tab);
    delete c;
    dev_cache_trim();
}

const char *cellector_last_error(const cellector_ctx *c) { return c ? c->err.c_str() : "null ctx"; }

cellector_status cellector_set_stream(cellector_ctx *c, void *s)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return ctx_fail(c, CELLECTOR_EINVAL, "a multi-device ctx runs every shard on a stream of its own");
    if (c->owns_stream) return ctx_fail(c, CELLECTOR_EINVAL, "this shard's stream belongs to its multi-device ctx");
    c->stream = (hipStream_t)s;
    return CELLECTOR_OK;
}

cellector_status cellector_set_option(cellector_ctx *c, const char *key, int64_t v)
{
    if (!c || !key) return CELLECTOR_EINVAL;
    if (c->multi) return multi_set_option(c, key, v);
    if (!strcmp(key, "compute_expected")) c->compute_expected = v != 0;
    else if (!strcmp(key, "ref_arith")) {
        if (v && c->engine != 1) return ctx_fail(c, CELLECTOR_EINVAL, "ref_arith evaluates every entry with the reference's ln_gamma arithmetic: an engine 1 option (set engine 1 first)");
        c->ref_arith = v != 0;
    }
    else if (!strcmp(key, "timing")) {
        c->timing = v < 0 ? 0 : (v > 3 ? 3 : (int)v);
        for (KernelTimer &t : c->timers) t.calls = 0;
        if (c->timing && hipSetDevice(c->device) == hipSuccess)  // events ready before the timed loop starts
            while (c->ev_pool.size() < 64) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) break;
                c->ev_pool.push_back(e);
            }
    }
    else if (!strcmp(key, "keep_coo")) c->keep_coo = v != 0;
    else if (!strcmp(key, "bank_order")) c->bank_order = v != 0;
    else if (!strcmp(key, "fuse_filter")) c->fuse_filter = v != 0;
    else if (!strcmp(key, "synth_continue_pct")) {
        if (v < 0 || v > 90) return ctx_fail(c, CELLECTOR_EINVAL, "synth_continue_pct must be within 0..90");
        c->synth_continue_pct = (int)v;
    }
    else if (!strcmp(key, "overlap")) {
        if (v < 0 || v > 2) return ctx_fail(c, CELLECTOR_EINVAL, "overlap must be 0, 1 or 2");
        c->overlap = (int)v;
    }
    else if (!strcmp(key, "side_lds")) c->side_lds = (int)v;
    else if (!strcmp(key, "ovf_deep_wide")) c->ovf_deep_wide = v != 0;
    else if (!strcmp(key, "ovf_deep")) {
        if (v < -1 || v > 1) return ctx_fail(c, CELLECTOR_EINVAL, "ovf_deep must be -1 (automatic), 0 or 1");
        if (c->tiled_ready && v >= 0 && c->ovf_deep != (v != 0))
            return ctx_fail(c, CELLECTOR_EINVAL, "ovf_deep decides which overflow layouts the ingest builds: set it before the ingest");
        c->ovf_deep_opt = (int)v;
    }
    else if (!strcmp(key, "t2_waves")) {
        if (v < 1 || v > (1 << 20)) return ctx_fail(c, CELLECTOR_EINVAL, "t2_waves must be within 1..2^20");
        c->t2_waves = (int)v;
    }
    else if (!strcmp(key, "t2")) {
        if (v < -1 || v > 1) return ctx_fail(c, CELLECTOR_EINVAL, "t2 must be -1 (automatic), 0 or 1");
        if (c->tiled_ready && v >= 0 && c->t2 != (v != 0))
            return ctx_fail(c, CELLECTOR_EINVAL, "t2 decides which overflow layouts the ingest builds: set it before the ingest");
        c->t2_opt = (int)v;
    }
    else if (!strcmp(key, "t2_tiles")) {
        if (!(v == -1 || v == 0 || v == 6 || v == 8)) return ctx_fail(c, CELLECTOR_EINVAL, "t2_tiles must be -1 (automatic), 0, 6 or 8");
        if (c->tiled_ready && v >= 0 && c->t2_tiles != (int)v)
            return ctx_fail(c, CELLECTOR_EINVAL, "t2_tiles decides which layouts the ingest builds: set it before the ingest");
        c->t2_tiles_opt = (int)v;
    }
    else if (!strcmp(key, "norm_zero")) c->norm_zero = v != 0;
    else if (!strcmp(key, "sharded_select")) c->sharded_select = v < 0 ? -1 : (v != 0);
    else if (!strcmp(key, "parse_window")) c->parse_window_opt = v < 0 ? 0 : v;
    else if (!strcmp(key, "tile_groups")) {
        if (v < 0 || v > 64) return ctx_fail(c, CELLECTOR_EINVAL, "tile_groups must be 0 (automatic) or 1..64");
        c->tile_groups_opt = (int)v;
    }
    else if (!strcmp(key, "locus_mode")) {
        if (v < 0 || v > 2) return ctx_fail(c, CELLECTOR_EINVAL, "locus_mode must be 0 (automatic), 1 (stream) or 2 (minority-driven)");
        c->locus_mode = (int)v;
    }
    else if (!strcmp(key, "compact_bits")) {
        if (v != 0 && v != 32) return ctx_fail(c, CELLECTOR_EINVAL, "compact_bits must be 0 (automatic) or 32");
        c->c4_bits_opt = (int)v;
    } else if (!strcmp(key, "engine")) {
        if (v != 1 && v != 2) return ctx_fail(c, CELLECTOR_EINVAL, "engine must be 1 (CSR/CSC kernels) or 2 (tiled)");
        if (c->em_phase != 0) return ctx_fail(c, CELLECTOR_EINVAL, "cannot switch engine inside an iteration");
        if (v == 1 && c->state == cellector_ctx::ST_READY && !c->csc_ent && c->nnz)
            return ctx_fail(c, CELLECTOR_EINVAL, "engine 1 needs the by-locus CSC, which an engine-2 ingest releases: set engine 1 before the ingest");
        if (v == 2 && c->state == cellector_ctx::ST_READY && !c->tiled_ready) {
            HIPCHK(c, hipSetDevice(c->device));
            if (c->n_masked_loci) return ctx_fail(c, CELLECTOR_EINVAL, "switch to engine 2 before any locus is masked");
            CHK(tiled_build(c));
        }
        if (v == 2 && c->ref_arith) return ctx_fail(c, CELLECTOR_EINVAL, "ref_arith is an engine 1 option: clear it before switching to engine 2");
        c->engine = (int)v;
    }
    else return ctx_fail(c, CELLECTOR_EINVAL, "unknown option '%s'", key);
    return CELLECTOR_OK;
}

cellector_status cellector_set_partition(cellector_ctx *c, const uint64_t *bounds, int n_bounds)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_set_partition(c, bounds, n_bounds);
    REQUIRE(c, comm_active(c->comm), "set_partition: the ctx has no communicator (a single shard takes cellector_set_shard)");
    REQUIRE(c, c->state == cellector_ctx::ST_EMPTY, "set_partition must precede the ingest");
    if (!bounds || n_bounds == 0) {  // back to the canonical equal ranges
        c->comm.has_bounds = false;
        return CELLECTOR_OK;
    }
    REQUIRE(c, n_bounds == c->comm.n + 1, "set_partition: one boundary more than there are ranks");
    REQUIRE(c, bounds[0] == 0, "set_partition: the first range starts at cell 0");
    for (int r = 0; r < c->comm.n; r++) REQUIRE(c, bounds[r] <= bounds[r + 1], "set_partition: boundaries must not decrease");
    for (int r = 0; r <= c->comm.n; r++) c->comm.bounds[r] = bounds[r];
    c->comm.has_bounds = true;
    return CELLECTOR_OK;
}

cellector_status cellector_partition(const cellector_ctx *c, uint64_t *bounds_out, int *n_ranks)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) c = multi_shard0(c);
    if (n_ranks) *n_ranks = c->comm.n;
    if (bounds_out) {
        REQUIRE(c, c->state != cellector_ctx::ST_EMPTY, "partition: nothing loaded yet");
        for (int r = 0; r < c->comm.n; r++) {
            uint64_t b, e;
            comm_range(c->comm, c->total_cells, r, &b, &e);
            bounds_out[r] = b;
            bounds_out[r + 1] = e;
        }
        if (c->comm.n == 1) { bounds_out[0] = c->cell_begin; bounds_out[1] = c->cell_end; }
    }
    return CELLECTOR_OK;
}

cellector_status cellector_set_shard(cellector_ctx *c, uint64_t b, uint64_t e)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi || comm_active(c->comm))
        return ctx_fail(c, CELLECTOR_EINVAL, "a ctx with a communicator shards the cells itself (contiguous ranges by rank: cellector_set_partition)");
    REQUIRE(c, c->state == cellector_ctx::ST_EMPTY, "set_shard must precede ingest");
    REQUIRE(c, b <= e, "empty or inverted shard range");
    c->cell_begin = b;
    c->cell_end = e;
    return CELLECTOR_OK;
}

// ---- ingest -------------------------------------------------------------------------------------------
static cellector_status begin_ingest(cellector_ctx *c, uint64_t total_loci, uint64_t total_cells)
{
    SETDEV(c);
    {
        // keep a caller-bound PASS1 buffer across the reset
        double *bound = c->own_pass1 ? nullptr : c->x_pass1;
        uint64_t nb = c->n_pass1;
        uint64_t cb = c->cell_begin, ce = c->cell_end;
        free_matrix(c);
        c->cell_begin = cb; c->cell_end = ce;
        if (bound) { c->x_pass1 = bound; c->n_pass1 = nb; c->own_pass1 = false; }
    }
    REQUIRE(c, total_loci <= 0xffffffffull && total_cells <= 0xffffffffull, "dims exceed 32-bit indices");
    c->total_loci = total_loci;
    c->total_cells = total_cells;
    if (c->ingest_all_cells) {
        c->cell_begin = 0;
        c->cell_end = total_cells;
    } else if (comm_active(c->comm)) {  // rank r owns the r-th contiguous range: equal ranges, or the partition it was given
        if (c->comm.has_bounds && c->comm.bounds[c->comm.n] != total_cells)
            return ctx_fail(c, CELLECTOR_EINVAL, "the partition covers %llu cells, the matrix has %llu",
                            (unsigned long long)c->comm.bounds[c->comm.n], (unsigned long long)total_cells);
        comm_range(c->comm, total_cells, c->comm.rank, &c->cell_begin, &c->cell_end);
    }
    if (c->cell_end > total_cells) c->cell_end = total_cells;
    if (c->cell_begin > c->cell_end) c->cell_begin = c->cell_end;
    c->nloc = c->cell_end - c->cell_begin;
    const uint64_t need = (uint64_t)P1_PLANES * total_loci;
    if (c->x_pass1) {
        REQUIRE(c, c->n_pass1 >= need, "bound PASS1 exchange buffer too small");
    } else {
        CHK(dev_alloc(c, &c->x_pass1, need));
        c->own_pass1 = true;
    }
    c->n_pass1 = need;
    return CELLECTOR_OK;
}

cellector_status cellector_ingest_coo(cellector_ctx *c, uint64_t total_loci, uint64_t total_cells, uint64_t nnz,
                                      const uint32_t *locus0, const uint32_t *cell0, const uint32_t *alt,
                                      const uint32_t *ref)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_ingest_coo(c, total_loci, total_cells, nnz, locus0, cell0, alt, ref);
    REQUIRE(c, nnz == 0 || (locus0 && cell0 && alt && ref), "null COO array");
    CHK(begin_ingest(c, total_loci, total_cells));
    CHK(ingest_stage_host_coo(c, nnz, locus0, cell0, alt, ref));
    CHK(ingest_pass1(c));
    c->state = cellector_ctx::ST_STAGED;
    return CELLECTOR_OK;
}

// CELLECTOR_TIMING=1: phase wall times of the ingest on stderr
static double lap_s(std::chrono::steady_clock::time_point *t)
{
    const auto now = std::chrono::steady_clock::now();
    const double s = std::chrono::duration<double>(now - *t).count();
    *t = now;
    return s;
}

cellector_status cellector_ingest_mtx(cellector_ctx *c, const char *alt_path, const char *ref_path)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_ingest_mtx(c, alt_path, ref_path);
    REQUIRE(c, alt_path && ref_path, "null path");
    const bool timing = getenv("CELLECTOR_TIMING") != nullptr && c->comm.rank == 0;
    auto t = std::chrono::steady_clock::now();
    MtxInput *in = nullptr;
    uint64_t tl = 0, tc = 0;
    CHK(mtx_input_open(c, alt_path, ref_path, &in, &tl, &tc));
    if (timing) fprintf(stderr, "[timing]   open / inflate          %8.3f s\n", lap_s(&t));
    cellector_status s = begin_ingest(c, tl, tc);
    if (s == CELLECTOR_OK) s = ingest_stage_mtx_device(c, in);  // tokenised and converted on the GPU
    mtx_input_close(in);
    CHK(s);
    if (timing) fprintf(stderr, "[timing]   upload + device parse   %8.3f s\n", lap_s(&t));
    CHK(ingest_pass1(c));
    c->state = cellector_ctx::ST_STAGED;
    return CELLECTOR_OK;
}

}  // extern "C"

// Multi-device text ingest, step 1: shard `c` tokenises the whole pair and stages the entries of ALL cells (global cell index).
cellector_status ffi_stage_mtx_all_cells(cellector_ctx *c, const char *alt_path, const char *ref_path, cellector_ctx *helper)
{
    REQUIRE(c, alt_path && ref_path, "null path");
    MtxInput *in = nullptr;
    uint64_t tl = 0, tc = 0;
    CHK(mtx_input_open(c, alt_path, ref_path, &in, &tl, &tc));
    c->ingest_all_cells = true;
    cellector_status s = begin_ingest(c, tl, tc);
    if (s == CELLECTOR_OK) s = ingest_stage_mtx_device(c, in, helper);
    c->ingest_all_cells = false;
    mtx_input_close(in);
    return s;
}
// ... step 2: a shard takes over its routed entries (arrays on its own device, cell index local, file order) as its staged COO.
cellector_status ffi_adopt_staged(cellector_ctx *c, uint64_t total_loci, uint64_t total_cells, uint32_t *locus, uint32_t *cell,
                                  uint16_t *alt, uint16_t *ref, uint64_t n, bool sorted)
{
    CHK(begin_ingest(c, total_loci, total_cells));
    c->coo_locus = locus; c->coo_cell = cell; c->coo_alt = alt; c->coo_ref = ref;
    c->coo_n = n;
    c->coo_sorted = sorted;
    CHK(ingest_pass1(c));
    c->state = cellector_ctx::ST_STAGED;
    return CELLECTOR_OK;
}

extern "C" {

cellector_status cellector_ingest_synthetic(cellector_ctx *c, uint64_t total_loci, uint64_t total_cells,
                                            double density, uint64_t seed, double minority_fraction,
                                            double doublet_fraction)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_ingest_synthetic(c, total_loci, total_cells, density, seed, minority_fraction, doublet_fraction);
    CHK(begin_ingest(c, total_loci, total_cells));
    CHK(synth_generate(c, density, seed, minority_fraction, doublet_fraction));
    CHK(ingest_pass1(c));
    c->state = cellector_ctx::ST_STAGED;
    return CELLECTOR_OK;
}

cellector_status cellector_write_staged_mtx(cellector_ctx *c, const char *alt_path, const char *ref_path)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return ctx_fail(c, CELLECTOR_EINVAL, "write_staged_mtx works on a single-device ctx (the staged entries of a multi-device ctx are sharded)");
    REQUIRE(c, alt_path && ref_path, "null path");
    REQUIRE(c, c->state != cellector_ctx::ST_EMPTY, "write_staged_mtx without a staged matrix");
    SETDEV(c);
    return synth_write_mtx(c, alt_path, ref_path);
}

cellector_status cellector_ingest_finish(cellector_ctx *c, uint64_t min_alt, uint64_t min_ref)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_ingest_finish(c, min_alt, min_ref);
    REQUIRE(c, c->state == cellector_ctx::ST_STAGED, "ingest_finish without a staged matrix");
    SETDEV(c);
    const bool timing = getenv("CELLECTOR_TIMING") != nullptr && c->comm.rank == 0;
    auto t = std::chrono::steady_clock::now();
    if (comm_active(c->comm)) {  // exchange point 1: global pass-1 counts and allele totals (every shard applies the same locus filter)
        CHK((cellector_status)comm_allreduce_sum(c, c->x_pass1, (uint64_t)P1_PLANES * c->total_loci));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    CHK(ingest_build(c, min_alt, min_ref));
    if (timing) fprintf(stderr, "[timing]   CSC / CSR build         %8.3f s\n", lap_s(&t));
    const uint64_t L = c->L, n = c->nloc;
    CHK(dev_alloc(c, &c->ab, L)); CHK(dev_alloc(c, &c->ab6, 8 * L));
    CHK(dev_alloc(c, &c->mask, L)); CHK(dev_alloc(c, &c->mask_next, L));
    CHK(dev_alloc(c, &c->flags, n)); CHK(dev_alloc(c, &c->flags_new, n));
    CHK(dev_alloc(c, &c->ll, n)); CHK(dev_alloc(c, &c->ell, n)); CHK(dev_alloc(c, &c->nloci, n));
    CHK(dev_alloc(c, &c->post, 4 * n));
    HIPCHK(c, hipMemsetAsync(c->mask, 1, L ? L : 1, c->stream));  // load_data.rs:176-179: all loci used
    HIPCHK(c, hipMemsetAsync(c->mask_next, 1, L ? L : 1, c->stream));
    HIPCHK(c, hipMemsetAsync(c->flags, 0, n ? n : 1, c->stream));
    HIPCHK(c, hipMemsetAsync(c->flags_new, 0, n ? n : 1, c->stream));
    HIPCHK(c, hipMemsetAsync(c->ll, 0, (n ? n : 1) * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->ell, 0, (n ? n : 1) * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->nloci, 0, (n ? n : 1) * 8, c->stream));
    // (with a communicator every rank owns an equal slot of NORM: the in-place all-gather's layout)
    const uint64_t need_norm = comm_active(c->comm) ? comm_cells_per_rank(c->total_cells, c->comm.n) * (uint64_t)c->comm.n : c->total_cells;
    const uint64_t need_locus = (uint64_t)LB_PLANES * L + LC_COUNTERS;
    if (c->x_norm) REQUIRE(c, c->n_norm >= need_norm, "bound NORM exchange buffer too small");
    else { CHK(dev_alloc(c, &c->x_norm, need_norm)); c->own_norm = true; }
    if (c->x_locus) REQUIRE(c, c->n_locus >= need_locus, "bound LOCUS exchange buffer too small");
    else { CHK(dev_alloc(c, &c->x_locus, need_locus)); c->own_locus = true; }
    c->n_norm = need_norm;
    c->n_locus = need_locus;
    HIPCHK(c, hipMemsetAsync(c->x_norm, 0, (need_norm ? need_norm : 1) * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->x_locus, 0, need_locus * 8, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {
        // The near-tie band of cellector_iter_summary.n_near_threshold follows the matrix' depth: the reference's ln_gamma
        // differences (stats.rs:41-53) carry ~eps * lnGamma(alpha + beta) of cancellation error per term, which the device's
        // product form does not reproduce — 1e-11 on a normalised LL at alpha + beta ~ 1e4 (vartrix-like depth), 1e-8 at 1e6.
        // band = max(1e-9, 8 eps lnGamma(max over the used loci of S_alt + S_ref + 2)), relative to max(1, |threshold|);
        // the global totals are the same on every shard.
        std::vector<double> sa(L), sr(L);
        if (L) {
            HIPCHK(c, hipMemcpy(sa.data(), c->s_alt, L * 8, hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(sr.data(), c->s_ref, L * 8, hipMemcpyDeviceToHost));
        }
        double max_ab = 2.0;
        for (uint64_t l = 0; l < L; l++) max_ab = std::max(max_ab, sa[l] + sr[l] + 2.0);
        c->near_rel = std::max(CELLECTOR_NEAR_TIE_REL, 8.0 * 2.220446049250313e-16 * lgamma(max_ab));
    }
    if (c->engine == 2) {
        CHK(tiled_build(c));
        // the packed by-locus CSC (8 B per entry: 16 GB at 2e9 entries) is only streamed by engine 1; engine 2 has built
        // its compact CSC and overflow CSC from it.  Engine 1 must therefore be chosen BEFORE the ingest.
        dev_free(c->csc_ent);
    }
    if (timing) fprintf(stderr, "[timing]   tiled layouts           %8.3f s\n", lap_s(&t));
    dev_cache_trim(c->device);  // the ingest's big temporaries are done: hand this device's cached blocks back
    c->state = cellector_ctx::ST_READY;
    c->em_phase = 0; c->iteration = 0; c->have_iter = false; c->n_excluded_global = 0;
    return CELLECTOR_OK;
}

cellector_status cellector_load_mtx(cellector_ctx *c, const char *a, const char *r, uint64_t min_alt, uint64_t min_ref)
{
    CHK(cellector_ingest_mtx(c, a, r));
    return cellector_ingest_finish(c, min_alt, min_ref);
}

cellector_status cellector_load_coo(cellector_ctx *c, uint64_t tl, uint64_t tc, uint64_t nnz, const uint32_t *l,
                                    const uint32_t *ce, const uint32_t *a, const uint32_t *r, uint64_t min_alt,
                                    uint64_t min_ref)
{
    CHK(cellector_ingest_coo(c, tl, tc, nnz, l, ce, a, r));
    return cellector_ingest_finish(c, min_alt, min_ref);
}

// ---- accessors ----------------------------------------------------------------------------------------
cellector_status cellector_dims(const cellector_ctx *c, cellector_dims_t *o)
{
    if (!c || !o) return CELLECTOR_EINVAL;
    if (c->multi) return multi_dims(c, o);
    o->total_cells = c->total_cells; o->total_loci = c->total_loci; o->loci_used = c->L;
    o->cell_begin = c->cell_begin; o->cell_end = c->cell_end; o->nnz_used = c->nnz;
    return CELLECTOR_OK;
}

static cellector_status d2h(const cellector_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return CELLECTOR_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CELLECTOR_OK;
}
#define READY(c) REQUIRE(c, (c) && (c)->state == cellector_ctx::ST_READY, "no matrix loaded")
// per-locus state is replicated on every shard: shard 0 of a multi-device ctx answers
#define SHARD0(c, call)                                                      \
    do {                                                                     \
        if ((c)->multi) {                                                    \
            const cellector_ctx *s0__ = multi_shard0(c);                     \
            const cellector_status st__ = (call);                            \
            if (st__ != CELLECTOR_OK) (c)->err = s0__->err;                  \
            return st__;                                                     \
        }                                                                    \
    } while (0)

cellector_status cellector_locus_ids(const cellector_ctx *c, uint64_t *out)
{
    if (!c) return CELLECTOR_EINVAL;
    SHARD0(c, cellector_locus_ids(s0__, out));
    READY(c);
    return d2h(c, out, c->locus_ids, c->L * 8);
}

cellector_status cellector_locus_counts(const cellector_ctx *c, double *out)
{
    if (!c) return CELLECTOR_EINVAL;
    SHARD0(c, cellector_locus_counts(s0__, out));
    READY(c);
    std::vector<double> a(c->L), r(c->L);
    CHK(d2h(c, a.data(), c->s_alt, c->L * 8));
    CHK(d2h(c, r.data(), c->s_ref, c->L * 8));
    for (uint64_t l = 0; l < c->L; l++) { out[2 * l] = r[l]; out[2 * l + 1] = a[l]; }
    return CELLECTOR_OK;
}

cellector_status cellector_entries_per_cell(const cellector_ctx *c, uint32_t *out)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_entries_per_cell(c, out);
    READY(c);
    std::vector<uint64_t> p(c->nloc + 1);
    CHK(d2h(c, p.data(), c->csr_ptr, (c->nloc + 1) * 8));
    for (uint64_t i = 0; i < c->nloc; i++) out[i] = (uint32_t)(p[i + 1] - p[i]);
    return CELLECTOR_OK;
}

cellector_status cellector_csr_rows(const cellector_ctx *c, uint64_t rb, uint64_t re, uint64_t *row_ptr,
                                    uint64_t *entries, uint64_t capacity)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_csr_rows(c, rb, re, row_ptr, entries, capacity);
    READY(c);
    REQUIRE(c, rb <= re && re <= c->nloc && row_ptr, "bad row range");
    CHK(d2h(c, row_ptr, c->csr_ptr + rb, (re - rb + 1) * 8));
    const uint64_t base = row_ptr[0], cnt = row_ptr[re - rb] - base;
    for (uint64_t i = 0; i <= re - rb; i++) row_ptr[i] -= base;
    if (entries) {
        REQUIRE(c, capacity >= cnt, "entries capacity too small");
        CHK(d2h(c, entries, c->csr_ent + base, cnt * 8));
    }
    return CELLECTOR_OK;
}

// ---- exchange buffers ---------------------------------------------------------------------------------
cellector_status cellector_exchange_buffer(cellector_ctx *c, cellector_xchg which, void **dev_ptr, uint64_t *n)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return ctx_fail(c, CELLECTOR_EINVAL, "the exchange buffers of a multi-device ctx are internal");
    double *p = nullptr;
    uint64_t cnt = 0;
    switch (which) {
    case CELLECTOR_XCHG_PASS1: p = c->x_pass1; cnt = c->n_pass1; break;
    case CELLECTOR_XCHG_NORM:
        p = c->x_norm; cnt = c->state == cellector_ctx::ST_READY ? c->total_cells : c->n_norm; break;
    case CELLECTOR_XCHG_LOCUS:
        p = c->x_locus;
        cnt = c->state == cellector_ctx::ST_READY ? (uint64_t)LB_PLANES * c->L + LC_COUNTERS : c->n_locus; break;
    default: return ctx_fail(c, CELLECTOR_EINVAL, "unknown exchange buffer");
    }
    if (dev_ptr) *dev_ptr = p;
    if (n) *n = cnt;
    return CELLECTOR_OK;
}

cellector_status cellector_bind_exchange_buffer(cellector_ctx *c, cellector_xchg which, void *dev_ptr, uint64_t n)
{
    if (!c || !dev_ptr) return CELLECTOR_EINVAL;
    if (c->multi || comm_active(c->comm)) return ctx_fail(c, CELLECTOR_EINVAL, "a ctx with a communicator owns its exchange buffers");
    double *p = (double *)dev_ptr;
    c->tables_prebuilt = false;  // (tables built ahead read the old buffers)
    switch (which) {
    case CELLECTOR_XCHG_PASS1:
        REQUIRE(c, c->state == cellector_ctx::ST_EMPTY, "bind PASS1 before ingest");
        if (c->own_pass1) dev_free(c->x_pass1);
        c->x_pass1 = p; c->n_pass1 = n; c->own_pass1 = false;
        break;
    case CELLECTOR_XCHG_NORM:
        REQUIRE(c, c->state != cellector_ctx::ST_READY || n >= c->total_cells, "NORM buffer too small");
        if (c->own_norm) dev_free(c->x_norm);
        c->x_norm = p; c->n_norm = n; c->own_norm = false;
        break;
    case CELLECTOR_XCHG_LOCUS:
        REQUIRE(c, c->state != cellector_ctx::ST_READY || n >= (uint64_t)LB_PLANES * c->L + LC_COUNTERS,
                "LOCUS buffer too small");
        if (c->state == cellector_ctx::ST_READY) {
            // carry the current tallies over (they seed the next alpha/beta update)
            HIPCHK(c, hipMemcpyAsync(p, c->x_locus, ((uint64_t)LB_PLANES * c->L + LC_COUNTERS) * 8,
                                     hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (c->own_locus) dev_free(c->x_locus);
        c->x_locus = p; c->n_locus = n; c->own_locus = false;
        break;
    default: return ctx_fail(c, CELLECTOR_EINVAL, "unknown exchange buffer");
    }
    return CELLECTOR_OK;
}

// ---- EM iteration -------------------------------------------------------------------------------------
cellector_status cellector_em_begin(cellector_ctx *c)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return ctx_fail(c, CELLECTOR_EINVAL, "a multi-device ctx runs whole iterations: cellector_em_iteration");
    READY(c);
    REQUIRE(c, c->em_phase == 0, "em_begin: previous iteration not finished");
    SETDEV(c);
    // engine 2 forms alpha/beta inside its first kernel (k_build_tables); an empty shard has no cell pass at all
    if (c->engine != 2 || c->prebuilt_expected != c->compute_expected) c->tables_prebuilt = false;
    if (c->engine != 2 || c->nloc == 0) CHK(launch_alpha_beta(c));
    if (c->nloc != c->total_cells && c->norm_zero && !comm_active(c->comm))  // other shards' slices must be zero before a SUM exchange
        HIPCHK(c, hipMemsetAsync(c->x_norm, 0, c->total_cells * 8, c->stream));
    cellector_status st = c->engine == 2 ? tiled_cell_pass(c, c->ab, c->x_norm + c->cell_begin, true)
                                         : launch_cell_ll(c, c->ab, c->x_norm + c->cell_begin);
    c->work_zeroed = false;  // (only this iteration's first tile pass may rely on k_alpha_beta's reset)
    CHK(st);
    c->em_phase = 1;
    return CELLECTOR_OK;
}

cellector_status cellector_em_threshold(cellector_ctx *c, double iqr_multiple)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return ctx_fail(c, CELLECTOR_EINVAL, "a multi-device ctx runs whole iterations: cellector_em_iteration");
    READY(c);
    REQUIRE(c, c->em_phase == 1, "em_threshold without em_begin");
    SETDEV(c);
    const uint64_t n = c->total_cells;
    REQUIRE(c, n > 0, "no cells");
    // exact median / R-8 quartiles / threshold, all on the device (no host round trip in this phase).
    // Exchange point 2: a ctx with a communicator runs the radix select over the shards' keys where they are and exchanges
    // digit histograms (six all-reduces of 48 KB); option sharded_select = 0 gathers every shard's slice of the normalised
    // LLs instead and selects over all of them on every shard.  (A host that runs the exchanges itself gathers NORM before
    // this call.)
    if (comm_active(c->comm) && comm_sharded_select(c->comm, c->sharded_select, n)) {
        CHK(select_threshold_sharded(c, c->x_norm + c->cell_begin, c->nloc, n, iqr_multiple));
    } else {
        if (comm_active(c->comm)) CHK((cellector_status)comm_allgather_cells(c, c->x_norm, n));
        CHK(select_threshold(c, c->x_norm, n, iqr_multiple));
    }
    // (the counters k_flag adds to were reset by this iteration's k_alpha_beta)
    CHK(launch_flag(c, c->sel_out + 10));
    if (c->engine == 2) CHK(tiled_locus_pass(c));
    else CHK(launch_locus_stats(c));
    c->em_phase = 2;
    return CELLECTOR_OK;
}

cellector_status cellector_em_finish(cellector_ctx *c, cellector_iter_summary *out)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return ctx_fail(c, CELLECTOR_EINVAL, "a multi-device ctx runs whole iterations: cellector_em_iteration");
    READY(c);
    REQUIRE(c, c->em_phase == 2, "em_finish without em_threshold");
    SETDEV(c);
    // exchange point 3: per-locus minority tallies, contribution sums and the change counters
    if (comm_active(c->comm)) CHK((cellector_status)comm_allreduce_sum(c, c->x_locus, (uint64_t)LB_PLANES * c->L + LC_COUNTERS));
    if (c->filter_fused) c->filter_fused = false;  // (engine 2, unsharded: k_locus_finalize applied the filter)
    else CHK(launch_locus_filter(c));
    CHK(launch_iter_summary(c));
    // the next iteration's first kernel is queued behind the summary: it runs while the host waits for the summary, wakes
    // up and decides (should the loop end here, the tables it built are simply never used)
    if (c->engine == 2 && c->tiled_ready) CHK(tiled_prebuild_tables(c));
    // the fallback of the poll below: the event that rides on the table kernel's dispatch, else one recorded behind the summary
    hipEvent_t ev_wait = c->tab_event_valid ? c->ev_tab : c->ev_sum;
    if (!c->tab_event_valid) HIPCHK(c, hipEventRecord(c->ev_sum, c->stream));

    // The iteration's only host synchronisation.  The summary kernel stores its sequence number behind the values in
    // pinned memory: polling that wakes the host a few tens of microseconds before hipEventSynchronize returns, and the
    // next iteration's tile kernel is the next thing the GPU waits for.  (The event stays the fallback: it is polled too,
    // and waited for once the summary is 20 ms late.)
    {
        const double want = (double)c->sum_seq;
        const uint64_t *seq = reinterpret_cast<const uint64_t *>(c->h_sel + CELLECTOR_SUM_SEQ);
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spin = 1;; spin++) {
            const uint64_t bits = __atomic_load_n(seq, __ATOMIC_ACQUIRE);
            double have;
            memcpy(&have, &bits, sizeof have);
            if (have == want) break;
            if ((spin & 0xfffu) == 0) {
                if (hipEventQuery(ev_wait) == hipSuccess) break;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
                    HIPCHK(c, hipEventSynchronize(ev_wait));
                    break;
                }
            }
            __builtin_ia32_pause();
        }
    }
    double cnt[LC_COUNTERS];
    uint32_t dc[8] = {0};
    for (int i = 0; i < LC_COUNTERS; i++) cnt[i] = c->h_sel[i];
    dc[0] = (uint32_t)c->h_sel[LC_COUNTERS];
    c->last_median = c->h_sel[LC_COUNTERS + 1]; c->last_iqr = c->h_sel[LC_COUNTERS + 2]; c->last_thr = c->h_sel[LC_COUNTERS + 3];
    if (dc[0]) {
        c->n_masked_loci += dc[0];
        if (c->tiled_ready) CHK(tiled_masked_update(c));
    }
    std::swap(c->flags, c->flags_new);   // excluded_cells <- new_excluded (main.rs:43)
    std::swap(c->mask, c->mask_next);    // loci_used for the next iteration; mask_next keeps this iteration's
    c->n_excluded_global = (uint64_t)cnt[LC_N_EXCLUDED];
    c->iteration++;
    c->have_iter = true;
    c->em_phase = 0;
    // (timers are read out when asked for — cellector_kernel_time — not here: waiting for the last event pair and destroying
    //  the events is host time on the path to the next iteration's first launch; a long run is drained now and then)
    if (c->timing && c->timers[CELLECTOR_K_TILE_LL].start.size() + c->timers[CELLECTOR_K_CELL_LL].start.size() > 512) timer_collect(c);
    if (out) {
        out->n_new_excluded = (uint64_t)cnt[LC_N_NEW];
        out->n_rescued = (uint64_t)cnt[LC_N_RESCUED];
        out->n_excluded = c->n_excluded_global;
        out->any_change = (out->n_new_excluded > 0 || out->n_rescued > 0) ? 1 : 0;  // main.rs:335
        out->n_loci_filtered = dc[0];
        out->n_near_threshold = (uint64_t)cnt[LC_N_NEAR];
        out->median = c->last_median; out->iqr = c->last_iqr; out->threshold = c->last_thr;
    }
    return CELLECTOR_OK;
}

cellector_status cellector_em_iteration(cellector_ctx *c, double iqr_multiple, cellector_iter_summary *out)
{
    if (c && c->multi) return multi_em_iteration(c, iqr_multiple, out);
    CHK(cellector_em_begin(c));
    CHK(cellector_em_threshold(c, iqr_multiple));
    return cellector_em_finish(c, out);
}

cellector_status cellector_iter_cell_outputs(const cellector_ctx *c, double *ll, double *ell, double *nl, double *norm)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_iter_cell_outputs(c, ll, ell, nl, norm);
    READY(c);
    const size_t b = c->nloc * 8;
    if (ll) CHK(d2h(c, ll, c->ll, b));
    if (ell) CHK(d2h(c, ell, c->ell, b));
    if (nl) CHK(d2h(c, nl, c->nloci, b));
    if (norm) CHK(d2h(c, norm, c->x_norm + c->cell_begin, b));
    return CELLECTOR_OK;
}

cellector_status cellector_iter_locus_outputs(const cellector_ctx *c, double *cmin, double *cmaj, uint64_t *nmin,
                                              uint64_t *nmaj, uint64_t *amin, uint64_t *rmin, uint64_t *amaj,
                                              uint64_t *rmaj)
{
    if (!c) return CELLECTOR_EINVAL;
    SHARD0(c, cellector_iter_locus_outputs(s0__, cmin, cmaj, nmin, nmaj, amin, rmin, amaj, rmaj));
    READY(c);
    REQUIRE(c, c->have_iter && c->em_phase == 0, "no finished iteration");
    const uint64_t L = c->L;
    std::vector<double> buf(LB_PLANES * L), sa(L), sr(L), ne(L);
    std::vector<uint8_t> m(L);
    CHK(d2h(c, buf.data(), c->x_locus, LB_PLANES * L * 8));
    CHK(d2h(c, sa.data(), c->s_alt, L * 8));
    CHK(d2h(c, sr.data(), c->s_ref, L * 8));
    CHK(d2h(c, ne.data(), c->n_ent, L * 8));
    CHK(d2h(c, m.data(), c->mask_next, L));  // the mask this iteration's passes ran under
    for (uint64_t l = 0; l < L; l++) {
        const bool live = m[l] != 0;  // a masked locus has no PMFData at all (main.rs:556)
        const double am = buf[LB_ALT_MIN * L + l], rm = buf[LB_REF_MIN * L + l], cm = buf[LB_CELLS_MIN * L + l];
        if (cmin) cmin[l] = buf[LB_CONTRIB_MIN * L + l];
        if (cmaj) cmaj[l] = buf[LB_CONTRIB_MAJ * L + l];
        if (nmin) nmin[l] = (uint64_t)cm;
        if (nmaj) nmaj[l] = live ? (uint64_t)(ne[l] - cm) : 0;
        if (amin) amin[l] = live ? (uint64_t)am : 0;
        if (rmin) rmin[l] = live ? (uint64_t)rm : 0;
        if (amaj) amaj[l] = live ? (uint64_t)(sa[l] - am) : 0;
        if (rmaj) rmaj[l] = live ? (uint64_t)(sr[l] - rm) : 0;
    }
    return CELLECTOR_OK;
}

cellector_status cellector_loci_mask(const cellector_ctx *c, uint8_t *out)
{
    if (!c) return CELLECTOR_EINVAL;
    SHARD0(c, cellector_loci_mask(s0__, out));
    READY(c);
    return d2h(c, out, c->mask, c->L);
}

cellector_status cellector_excluded(const cellector_ctx *c, uint8_t *out)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_excluded(c, out);
    READY(c);
    return d2h(c, out, c->flags, c->nloc);
}

cellector_status cellector_alpha_betas(const cellector_ctx *c, double *alpha, double *beta)
{
    if (!c) return CELLECTOR_EINVAL;
    SHARD0(c, cellector_alpha_betas(s0__, alpha, beta));
    READY(c);
    const uint64_t L = c->L;
    std::vector<double> buf(LB_PLANES * L), sa(L), sr(L);
    CHK(d2h(c, buf.data(), c->x_locus, LB_PLANES * L * 8));
    CHK(d2h(c, sa.data(), c->s_alt, L * 8));
    CHK(d2h(c, sr.data(), c->s_ref, L * 8));
    for (uint64_t l = 0; l < L; l++) {
        if (alpha) alpha[l] = (sa[l] + 1.0) - buf[LB_ALT_MIN * L + l];
        if (beta) beta[l] = (sr[l] + 1.0) - buf[LB_REF_MIN * L + l];
    }
    return CELLECTOR_OK;
}

cellector_status cellector_cell_log_likelihoods(cellector_ctx *c, const double *alpha, const double *beta,
                                                const uint8_t *mask, double *ll, double *ell, double *nl)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_cell_log_likelihoods(c, alpha, beta, mask, ll, ell, nl);
    READY(c);
    REQUIRE(c, alpha && beta, "null alpha/beta");
    REQUIRE(c, c->em_phase == 0, "iteration in flight");
    SETDEV(c);
    c->tables_prebuilt = false;  // this pass overwrites alpha/beta and the tables
    c->work_zeroed = false;
    CHK(launch_ab_from_host(c, alpha, beta, mask));
    // the tiled engine derives the used-locus count from the ctx's own mask; with a caller mask use the CSR kernel
    if (c->engine == 2 && !mask && c->n_masked_loci == 0) CHK(tiled_cell_pass(c, c->ab, nullptr, false));
    else CHK(launch_cell_ll(c, c->ab, nullptr));
    const size_t b = c->nloc * 8;
    if (ll) CHK(d2h(c, ll, c->ll, b));
    if (ell) CHK(d2h(c, ell, c->ell, b));
    if (nl) CHK(d2h(c, nl, c->nloci, b));
    if (c->timing) timer_collect(c);
    return CELLECTOR_OK;
}

// ---- posteriors ---------------------------------------------------------------------------------------
cellector_status cellector_posteriors(cellector_ctx *c, double *posterior, double *doublet, double *ll_maj,
                                      double *ll_min)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_posteriors(c, posterior, doublet, ll_maj, ll_min);
    READY(c);
    REQUIRE(c, c->em_phase == 0, "iteration in flight");
    SETDEV(c);
    const double N = (double)c->total_cells;
    const double mf0 = ((double)c->n_excluded_global + 1.0) / (N + 1.0);  // main.rs:240
    const double mf = std::fmax(mf0, 0.01);                               // main.rs:250
    const double lp_dbl = std::log(N / 1000.0 / 100.0 * std::fmax(mf, 0.1));  // main.rs:259
    const double lp_min = std::log(mf), lp_maj = std::log(1.0 - mf);      // main.rs:264-265
    if (c->engine == 2) CHK(tiled_posteriors(c, mf0, lp_min, lp_maj, lp_dbl));
    else CHK(launch_posteriors(c, mf0, lp_min, lp_maj, lp_dbl));
    const size_t b = c->nloc * 8;
    if (posterior) CHK(d2h(c, posterior, c->post, b));
    if (doublet) CHK(d2h(c, doublet, c->post + c->nloc, b));
    if (ll_maj) CHK(d2h(c, ll_maj, c->post + 2 * c->nloc, b));
    if (ll_min) CHK(d2h(c, ll_min, c->post + 3 * c->nloc, b));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->timing) timer_collect(c);
    return CELLECTOR_OK;
}

cellector_status cellector_final_allele_tallies(cellector_ctx *c, uint64_t *alt_min, uint64_t *ref_min,
                                                uint64_t *alt_maj, uint64_t *ref_maj)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_final_allele_tallies(c, alt_min, ref_min, alt_maj, ref_maj);
    READY(c);
    REQUIRE(c, c->keep_coo, "final tallies need the staged COO (option keep_coo=1)");
    SETDEV(c);
    const uint64_t TL = c->total_loci;
    uint64_t *d = nullptr;
    CHK(dev_alloc(c, &d, 4 * TL));
    cellector_status s = launch_final_tallies(c, d);
    std::vector<uint64_t> h(4 * TL);
    if (s == CELLECTOR_OK) s = d2h(c, h.data(), d, 4 * TL * 8);
    dev_free(d);
    CHK(s);
    if (alt_min) memcpy(alt_min, h.data(), TL * 8);
    if (ref_min) memcpy(ref_min, h.data() + TL, TL * 8);
    if (alt_maj) memcpy(alt_maj, h.data() + 2 * TL, TL * 8);
    if (ref_maj) memcpy(ref_maj, h.data() + 3 * TL, TL * 8);
    return CELLECTOR_OK;
}

cellector_status cellector_engine_info(const cellector_ctx *c, cellector_engine_info_t *o)
{
    if (!c || !o) return CELLECTOR_EINVAL;
    if (c->multi) return multi_engine_info(c, o);
    memset(o, 0, sizeof *o);
    o->engine = (uint64_t)c->engine;
    if (c->tiled_ready) {
        o->nnz_regular = c->nnz - c->ovf_n;
        o->nnz_overflow = c->ovf_n;
        o->cell_blocks = c->t_nb; o->locus_chunks = c->t_nj; o->chunk_groups = c->t_groups;
        const uint64_t elems = c->t_elems + (uint64_t)c->t_nb * c->t_nj * 128;  // slices + slice headers (u16 units)
        o->tile_bytes = elems * 2;
        o->tile_lookups = c->t_elems - (uint64_t)c->t_nb * c->t_nj * T_ROWS_PER_TILE;  // a row = its cell id + K entries
    }
    return CELLECTOR_OK;
}

// ---- timing -------------------------------------------------------------------------------------------
cellector_status cellector_kernel_time(cellector_ctx *c, cellector_kernel_id which, double *total_ms, uint64_t *launches)
{
    if (!c || which < 0 || which >= CELLECTOR_K_COUNT) return CELLECTOR_EINVAL;
    if (c->multi) return cellector_kernel_time(multi_shard0(c), which, total_ms, launches);
    (void)hipSetDevice(c->device);
    timer_collect(c);
    if (total_ms) *total_ms = c->timers[which].total_ms;
    if (launches) *launches = c->timers[which].launches;
    return CELLECTOR_OK;
}

}  // extern "C"
// (a shard of a multi-device ctx is handed its slice of the keys; n_total = all keys)
cellector_status ffi_order_statistics(cellector_ctx *c, const double *keys, uint64_t n_local, uint64_t n_total, double iqr_multiple, double *out3)
{
    REQUIRE(c, n_total > 0 && (keys || !n_local), "order statistics: no keys");
    SETDEV(c);
    double *d_keys = nullptr;
    CHK(dev_alloc(c, &d_keys, n_local ? n_local : 1));
    cellector_status st = CELLECTOR_OK;
    if (n_local && hipMemcpyAsync(d_keys, keys, n_local * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        st = ctx_fail(c, CELLECTOR_EDEVICE, "upload of the keys failed");
    if (st == CELLECTOR_OK)
        st = comm_active(c->comm) && comm_sharded_select(c->comm, c->sharded_select, n_total) ? select_threshold_sharded(c, d_keys, n_local, n_total, iqr_multiple)
                                                       : select_threshold(c, d_keys, n_local, iqr_multiple);
    if (st == CELLECTOR_OK && out3) st = d2h(c, out3, c->sel_out + 8, 3 * sizeof(double));
    else (void)hipStreamSynchronize(c->stream);
    dev_free(d_keys);
    return st;
}

extern "C" {

cellector_status cellector_order_statistics(cellector_ctx *c, const double *keys, uint64_t n, double iqr_multiple, double *out3)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_order_statistics(c, keys, n, iqr_multiple, out3);
    REQUIRE(c, !comm_active(c->comm) || c->comm.n == 1,
            "order statistics on one rank of a communicator: every rank would have to call with its slice");
    return ffi_order_statistics(c, keys, n, n, iqr_multiple, out3);
}

cellector_status cellector_reset_timing(cellector_ctx *c)
{
    if (!c) return CELLECTOR_EINVAL;
    if (c->multi) return multi_reset_timing(c);
    (void)hipSetDevice(c->device);
    timer_collect(c);
    for (int k = 0; k < CELLECTOR_K_COUNT; k++) { c->timers[k].total_ms = 0.0; c->timers[k].launches = 0; }
    return CELLECTOR_OK;
}

}  // extern "C"
