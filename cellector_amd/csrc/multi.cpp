// cellector_create_multi: ONE ctx over several GPUs of one process (multi-GPU is internal to the ctx, the host
// sees one logical matrix, every returned array is in global cell order).  The root ctx owns one shard ctx per listed device —
// contiguous, equal cell ranges, per-locus state replicated — one worker thread per shard, and the exchange transport
// (comm.h): RCCL over xGMI when the devices are distinct (ncclCommInitAll), device-side sums when a device is listed more than
// once (logical shards on one GPU: tests and one-GPU rehearsals; RCCL refuses duplicate devices).  Every public entry point
// called on the root fans out to the shards on their worker threads; the shards run the same code as a single-GPU ctx, with
// the three exchanges (PASS1 / NORM / LOCUS) done inside the library.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <functional>
#include <thread>

#include "ctx.h"
#include "multi.h"

struct MultiCtx {
    std::vector<cellector_ctx *> shards;
    LocalGroup local;
    bool use_local = false;
    // one persistent worker per shard: a job is the same function run on every shard
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::function<cellector_status(cellector_ctx *, int)> job;
    std::atomic<uint64_t> job_seq{0};
    int pending = 0;
    bool stop = false;
    std::vector<cellector_status> result;
    bool balance = true;         // option "balance": the text / COO ingest cuts the cells by entries (comm_balanced_bounds), not by count
    bool user_partition = false;  // cellector_set_partition was called: the ingest keeps those ranges
    bool comm_aborted = false;  // (under mu) a shard failed in RCCL mode: the communicators are gone, the ctx takes no more sharded work
};

static void worker_main(MultiCtx *m, int rank)
{
    cellector_ctx *c = m->shards[(size_t)rank];
    (void)hipSetDevice(c->device);
    uint64_t seen = 0;
    for (;;) {
        // Iterations of the scoring loop arrive back to back and take fractions of a millisecond: spin briefly for the
        // next job before sleeping on the condition variable (a futex wake-up is tens of microseconds).
        const auto t0 = std::chrono::steady_clock::now();
        while (m->job_seq.load(std::memory_order_acquire) == seen) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                std::unique_lock<std::mutex> lk(m->mu);
                m->cv_work.wait(lk, [&] { return m->job_seq.load(std::memory_order_acquire) != seen || m->stop; });
                break;
            }
            __builtin_ia32_pause();
        }
        std::function<cellector_status(cellector_ctx *, int)> fn;
        {
            std::lock_guard<std::mutex> lk(m->mu);
            if (m->stop) return;
            seen = m->job_seq.load(std::memory_order_acquire);
            fn = m->job;
        }
        const cellector_status st = fn(c, rank);
        if (st != CELLECTOR_OK && m->use_local) m->local.fail();  // (the other shards may be waiting for this one)
        if (st != CELLECTOR_OK && !m->use_local && m->shards.size() > 1) {
            // RCCL: the peers would sit in a collective (or in the stream wait behind it) forever — abort the communicators
            std::lock_guard<std::mutex> lk(m->mu);
            if (!m->comm_aborted) {
                m->comm_aborted = true;
                for (cellector_ctx *s : m->shards) comm_rccl_abort(s);
            }
        }
        {
            std::lock_guard<std::mutex> lk(m->mu);
            m->result[(size_t)rank] = st;
            if (--m->pending == 0) m->cv_done.notify_all();
        }
    }
}

// runs fn on every shard (on the shards' own threads) and returns the first failure, its message copied to the root
static cellector_status run_all(const cellector_ctx *root, std::function<cellector_status(cellector_ctx *, int)> fn)
{
    MultiCtx *m = root->multi;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->job = std::move(fn);
        m->pending = (int)m->shards.size();
        m->job_seq.fetch_add(1, std::memory_order_release);
    }
    m->cv_work.notify_all();
    {
        std::unique_lock<std::mutex> lk(m->mu);
        m->cv_done.wait(lk, [&] { return m->pending == 0; });
    }
    if (m->comm_aborted)  // (every worker is back: nobody reads the handles any more; ncclCommAbort has freed them)
        for (cellector_ctx *s : m->shards) s->comm.nccl = nullptr;
    {   // every worker is back: a failure that released the others from a barrier must not poison the next call
        std::lock_guard<std::mutex> lk(m->local.mu);
        m->local.failed = false;
        m->local.arrived = 0;
    }
    for (size_t r = 0; r < m->shards.size(); r++)
        if (m->result[r] != CELLECTOR_OK) {
            root->err = "shard " + std::to_string(r) + " (device " + std::to_string(m->shards[r]->device) + "): " + m->shards[r]->err;
            return m->result[r];
        }
    return CELLECTOR_OK;
}

static uint64_t shard_begin(const cellector_ctx *s) { return s->cell_begin; }

cellector_status multi_create(cellector_ctx **out, const int *device_ids, int n)
{
    cellector_ctx *root = new (std::nothrow) cellector_ctx();
    if (!root) return CELLECTOR_ENOMEM;
    MultiCtx *m = new (std::nothrow) MultiCtx();
    if (!m) { delete root; return CELLECTOR_ENOMEM; }
    root->multi = m;
    root->device = device_ids[0];
    cellector_status st = CELLECTOR_OK;
    bool dup = false;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++) dup = dup || device_ids[i] == device_ids[j];
    for (int i = 0; i < n && st == CELLECTOR_OK; i++) {
        cellector_ctx *s = nullptr;
        st = cellector_create(&s, device_ids[i]);
        if (st != CELLECTOR_OK) break;
        // every shard launches on a stream of its own (shards that share a device must not serialise on the null stream)
        if (hipSetDevice(s->device) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
            cellector_destroy(s);
            st = CELLECTOR_EDEVICE;
            break;
        }
        s->owns_stream = true;
        s->norm_zero = false;  // the NORM slices are all-gathered inside the library
        m->shards.push_back(s);
    }
    if (st == CELLECTOR_OK) {
        if (dup || getenv("CELLECTOR_COMM_LOCAL")) {  // logical shards on one device (or forced): device-side sums
            m->use_local = true;
            m->local.n = n;
            for (int r = 0; r < n; r++) {
                m->shards[(size_t)r]->comm.n = n;
                m->shards[(size_t)r]->comm.rank = r;
                m->shards[(size_t)r]->comm.local = &m->local;
            }
        } else {
            st = (cellector_status)comm_rccl_init_all(m->shards.data(), n, device_ids);
        }
    }
    if (st != CELLECTOR_OK) {
        fprintf(stderr, "cellector_create_multi: %s\n", m->shards.empty() ? "no usable device" : m->shards[0]->err.c_str());
        multi_destroy(root);
        return st;
    }
    m->result.assign((size_t)n, CELLECTOR_OK);
    for (int r = 0; r < n; r++) m->workers.emplace_back(worker_main, m, r);
    *out = root;
    return CELLECTOR_OK;
}

void multi_destroy(cellector_ctx *root)
{
    MultiCtx *m = root->multi;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->stop = true;
        m->job_seq.fetch_add(1, std::memory_order_release);
    }
    m->cv_work.notify_all();
    for (auto &t : m->workers) t.join();
    for (cellector_ctx *s : m->shards) cellector_destroy(s);  // (destroys the shard's communicator and stream first)
    delete m;
    root->multi = nullptr;
    delete root;
}

int multi_n_shards(const cellector_ctx *root) { return (int)root->multi->shards.size(); }

// ---- configuration ---------------------------------------------------------------------------------------------------------
cellector_status multi_set_option(cellector_ctx *root, const char *key, int64_t v)
{
    if (!strcmp(key, "norm_zero")) return CELLECTOR_OK;  // (the exchange is internal)
    if (!strcmp(key, "balance")) {
        root->multi->balance = v != 0;
        return CELLECTOR_OK;
    }
    for (cellector_ctx *s : root->multi->shards) {
        const cellector_status st = cellector_set_option(s, key, v);
        if (st != CELLECTOR_OK) {
            root->err = s->err;
            return st;
        }
    }
    return CELLECTOR_OK;
}

cellector_status multi_set_partition(cellector_ctx *root, const uint64_t *bounds, int n_bounds)
{
    MultiCtx *m = root->multi;
    for (cellector_ctx *s : m->shards) {
        const cellector_status st = cellector_set_partition(s, bounds, n_bounds);
        if (st != CELLECTOR_OK) {
            root->err = s->err;
            return st;
        }
    }
    m->user_partition = bounds != nullptr && n_bounds != 0;
    return CELLECTOR_OK;
}

// the ranges of the ingest that follows: the caller's (cellector_set_partition), or cut from the entries per cell so that every
// shard gets about the same number of entries ("balanced by nnz, not by cell count"), or — option balance = 0, and
// the synthetic generator, whose cells are alike — equal counts
static void set_ranges(MultiCtx *m, const uint32_t *entries_per_cell, uint64_t total_cells)
{
    if (m->user_partition) return;
    const int n = (int)m->shards.size();
    uint64_t bounds[CELLECTOR_MAX_SHARDS + 1];
    const bool bal = m->balance && entries_per_cell && n > 1;
    if (bal) comm_balanced_bounds(entries_per_cell, total_cells, n, bounds);
    for (cellector_ctx *s : m->shards) {
        s->comm.has_bounds = bal;
        if (bal) memcpy(s->comm.bounds, bounds, sizeof(uint64_t) * (size_t)(n + 1));
    }
}

// ---- ingest ------------------------------------------------------------------------------------------------------------------
// (every shard takes its cell range from its communicator in begin_ingest: a shard with one ignores cellector_set_shard)
// The text pair is read, uploaded and tokenised ONCE — the alt file on shard 0's device, the ref file on shard 1's at the same
// time (the two are independent byte streams until they are zipped line by line); its entries are then cut by owning cell range
// (file order kept) and every piece goes to its shard's device (peer copy over xGMI; a plain device copy between logical
// shards of one GPU).  n shards each parsing the whole pair would read the text n times on the host and push it over n PCIe
// links at once: slower than one GPU from a few shards on.  From three devices on every GPU parses 1/n of the bytes instead
// (the split ingest at the top of the function).
cellector_status multi_ingest_mtx(cellector_ctx *root, const char *alt_path, const char *ref_path)
{
    MultiCtx *m = root->multi;
    if (getenv("CELLECTOR_MULTI_PARSE_EACH"))  // (A/B: every shard parses the pair itself)
    {
        set_ranges(m, nullptr, 0);  // (no entry counts before the parse: equal ranges, or the caller's)
        return run_all(root, [=](cellector_ctx *s, int) { return cellector_ingest_mtx(s, alt_path, ref_path); });
    }
    cellector_ctx *s0 = m->shards[0];
    // ---- split ingest: every GPU tokenises 1/n of both files' bytes and the entries are routed to their owners
    // (kernels_parse.hip, ingest_stage_mtx_split).  Taken when the shards sit on at least three different devices (with two,
    // "alt here, ref there" below already uses both links) and both files are big enough for the windowed parser;
    // CELLECTOR_MULTI_SPLIT=1 forces it (tests on logical shards of one GPU), =0 switches it off.
    {
        int distinct = 0;
        for (size_t i = 0; i < m->shards.size(); i++) {
            bool seen = false;
            for (size_t j = 0; j < i; j++) seen = seen || m->shards[j]->device == m->shards[i]->device;
            distinct += seen ? 0 : 1;
        }
        const char *env = getenv("CELLECTOR_MULTI_SPLIT");
        const bool want = env ? atoi(env) != 0 : distinct >= 3;
        if (want) {
            MtxInput *in = nullptr;
            uint64_t tl = 0, tc = 0;
            if (hipSetDevice(s0->device) != hipSuccess) return ctx_fail(root, CELLECTOR_EDEVICE, "hipSetDevice failed");
            cellector_status st = mtx_input_open(s0, alt_path, ref_path, &in, &tl, &tc);
            if (st != CELLECTOR_OK) { root->err = s0->err; return st; }
            if (mtx_input_windowed(in, s0->parse_window_opt)) {
                const int n = (int)m->shards.size();
                LocalGroup bar;
                bar.n = n;
                if (!m->user_partition)
                    for (cellector_ctx *s : m->shards) s->comm.has_bounds = false;  // (the split ingest sets them, or equal ranges)
                MtxSplit *S = mtx_split_new(n, &bar, m->balance && !m->user_partition && n > 1);
                if (!S) { mtx_input_close(in); return ctx_fail(root, CELLECTOR_ENOMEM, "out of host memory"); }
                const uint64_t window = s0->parse_window_opt > 0 ? (uint64_t)s0->parse_window_opt : 0;
                st = run_all(root, [=](cellector_ctx *s, int rank) {
                    uint32_t *pl = nullptr, *pc = nullptr;
                    uint16_t *pa = nullptr, *pr = nullptr;
                    uint64_t cnt = 0;
                    bool sorted = false;
                    cellector_status r = ingest_stage_mtx_split(s, in, S, rank, window, &pl, &pc, &pa, &pr, &cnt, &sorted);
                    if (r != CELLECTOR_OK) return r;
                    return ffi_adopt_staged(s, tl, tc, pl, pc, pa, pr, cnt, sorted);
                });
                if (st == CELLECTOR_ECOMM)  // (report the shard that failed, not one of those it released)
                    for (size_t r = 0; r < m->shards.size(); r++)
                        if (m->result[r] != CELLECTOR_OK && m->result[r] != CELLECTOR_ECOMM) {
                            root->err = "shard " + std::to_string(r) + " (device " + std::to_string(m->shards[r]->device) + "): " + m->shards[r]->err;
                            st = m->result[r];
                            break;
                        }
                mtx_split_delete(S);
                mtx_input_close(in);
                return st;
            }
            mtx_input_close(in);  // (small files: one parser is enough)
        }
    }
    auto fail = [&](cellector_ctx *s, cellector_status st) {
        root->err = "shard (device " + std::to_string(s->device) + "): " + s->err;
        return st;
    };
    if (hipSetDevice(s0->device) != hipSuccess) return ctx_fail(root, CELLECTOR_EDEVICE, "hipSetDevice failed");
    // (a second shard tokenises the ref file meanwhile: on another GPU that is a second PCIe link and a second parser)
    // (logical shards of ONE GPU gain nothing from it — two parsers then share one link: 0.59 s against 0.26 s for 5.7 GB —
    //  so only a shard on another device helps; CELLECTOR_MULTI_REF_HELPER=1 forces it: tests of that path on a one-GPU box)
    cellector_ctx *helper = nullptr;
    for (size_t r = 1; r < m->shards.size() && !helper; r++)
        if (m->shards[r]->device != s0->device || getenv("CELLECTOR_MULTI_REF_HELPER")) helper = m->shards[r];
    cellector_status st = ffi_stage_mtx_all_cells(s0, alt_path, ref_path, helper);
    if (st != CELLECTOR_OK) return fail(s0, st);
    // detach the all-cells arrays from shard 0 (its own piece is cut from them like the others')
    cellector_ctx all;  // (a plain holder: no device state of its own is created or destroyed)
    all.device = s0->device; all.stream = s0->stream;
    all.coo_locus = s0->coo_locus; all.coo_cell = s0->coo_cell; all.coo_alt = s0->coo_alt; all.coo_ref = s0->coo_ref;
    all.coo_n = s0->coo_n;
    const bool sorted = s0->coo_sorted;
    const uint64_t TL = s0->total_loci, TC = s0->total_cells;
    s0->coo_locus = s0->coo_cell = nullptr; s0->coo_alt = s0->coo_ref = nullptr; s0->coo_n = 0;
    uint64_t *keep = nullptr;
    st = dev_alloc(&all, &keep, all.coo_n + 1);
    const int n = (int)m->shards.size();
    if (st == CELLECTOR_OK && m->balance && !m->user_partition && n > 1) {  // the ranges: cut by entries per cell
        std::vector<uint32_t> epc;
        st = ingest_cell_histogram(&all, all.coo_cell, all.coo_n, TC, &epc);
        if (st == CELLECTOR_OK) set_ranges(m, epc.data(), TC);
        else root->err = all.err;
    } else if (st == CELLECTOR_OK) {
        set_ranges(m, nullptr, TC);
    }
    for (int r = 0; r < n && st == CELLECTOR_OK; r++) {
        cellector_ctx *s = m->shards[(size_t)r];
        uint64_t cb, ce;
        comm_range(s->comm, TC, r, &cb, &ce);
        uint32_t *pl = nullptr, *pc = nullptr;
        uint16_t *pa = nullptr, *pr = nullptr;
        uint64_t cnt = 0;
        (void)hipSetDevice(all.device);
        st = ingest_split_coo(&all, cb, ce, keep, &pl, &pc, &pa, &pr, &cnt);
        if (st != CELLECTOR_OK) { root->err = all.err; break; }
        if (s->device != all.device) {  // move the piece to the shard's device
            uint32_t *ql = nullptr, *qc = nullptr;
            uint16_t *qa = nullptr, *qr = nullptr;
            (void)hipSetDevice(s->device);
            st = dev_alloc(s, &ql, cnt);
            if (st == CELLECTOR_OK) st = dev_alloc(s, &qc, cnt);
            if (st == CELLECTOR_OK) st = dev_alloc(s, &qa, cnt);
            if (st == CELLECTOR_OK) st = dev_alloc(s, &qr, cnt);
            if (st == CELLECTOR_OK && cnt &&
                (dev_copy_sync(s->stream, ql, s->device, pl, all.device, cnt * 4) != hipSuccess ||
                 dev_copy_sync(s->stream, qc, s->device, pc, all.device, cnt * 4) != hipSuccess ||
                 dev_copy_sync(s->stream, qa, s->device, pa, all.device, cnt * 2) != hipSuccess ||
                 dev_copy_sync(s->stream, qr, s->device, pr, all.device, cnt * 2) != hipSuccess))
                st = ctx_fail(s, CELLECTOR_EDEVICE, "peer copy of the shard's entries failed: %s", hipGetErrorString(hipGetLastError()));
            (void)hipSetDevice(all.device);
            dev_free(pl); dev_free(pc); dev_free(pa); dev_free(pr);
            if (st != CELLECTOR_OK) { dev_free(ql); dev_free(qc); dev_free(qa); dev_free(qr); fail(s, st); break; }
            pl = ql; pc = qc; pa = qa; pr = qr;
        }
        (void)hipSetDevice(s->device);
        st = ffi_adopt_staged(s, TL, TC, pl, pc, pa, pr, cnt, sorted);
        if (st != CELLECTOR_OK) fail(s, st);
    }
    (void)hipSetDevice(all.device);
    dev_free(keep);
    dev_free(all.coo_locus); dev_free(all.coo_cell); dev_free(all.coo_alt); dev_free(all.coo_ref);
    all.stream = nullptr;
    return st;
}
cellector_status multi_ingest_coo(cellector_ctx *root, uint64_t total_loci, uint64_t total_cells, uint64_t nnz, const uint32_t *locus0,
                                  const uint32_t *cell0, const uint32_t *alt, const uint32_t *ref)
{
    MultiCtx *m = root->multi;
    if (m->balance && !m->user_partition && m->shards.size() > 1 && cell0) {
        std::vector<uint32_t> epc(total_cells, 0u);
        for (uint64_t i = 0; i < nnz; i++)
            if (cell0[i] < total_cells) epc[cell0[i]]++;
        set_ranges(m, epc.data(), total_cells);
    } else {
        set_ranges(m, nullptr, total_cells);
    }
    return run_all(root, [=](cellector_ctx *s, int) { return cellector_ingest_coo(s, total_loci, total_cells, nnz, locus0, cell0, alt, ref); });
}
cellector_status multi_ingest_synthetic(cellector_ctx *root, uint64_t total_loci, uint64_t total_cells, double density, uint64_t seed,
                                        double minority_fraction, double doublet_fraction)
{
    set_ranges(root->multi, nullptr, total_cells);  // (the generator's cells are alike: equal ranges)
    return run_all(root, [=](cellector_ctx *s, int) {
        return cellector_ingest_synthetic(s, total_loci, total_cells, density, seed, minority_fraction, doublet_fraction);
    });
}
cellector_status multi_ingest_finish(cellector_ctx *root, uint64_t min_alt, uint64_t min_ref)
{
    return run_all(root, [=](cellector_ctx *s, int) { return cellector_ingest_finish(s, min_alt, min_ref); });
}

// ---- accessors ---------------------------------------------------------------------------------------------------------------
cellector_status multi_dims(const cellector_ctx *root, cellector_dims_t *o)
{
    const auto &sh = root->multi->shards;
    cellector_status st = cellector_dims(sh[0], o);
    o->cell_begin = 0;
    o->cell_end = o->total_cells;
    o->nnz_used = 0;
    for (const cellector_ctx *s : sh) o->nnz_used += s->nnz;
    return st;
}

// per-cell arrays: every shard writes its slice of the caller's buffers (global cell order)
template <typename F>
static cellector_status per_cell(const cellector_ctx *root, F fn)
{
    return run_all(root, [=](cellector_ctx *s, int) { return fn(s, shard_begin(s)); });
}

cellector_status multi_entries_per_cell(const cellector_ctx *root, uint32_t *out)
{
    return per_cell(root, [=](cellector_ctx *s, uint64_t b) { return cellector_entries_per_cell(s, out + b); });
}
cellector_status multi_excluded(const cellector_ctx *root, uint8_t *out)
{
    return per_cell(root, [=](cellector_ctx *s, uint64_t b) { return cellector_excluded(s, out + b); });
}
cellector_status multi_iter_cell_outputs(const cellector_ctx *root, double *ll, double *ell, double *nl, double *norm)
{
    return per_cell(root, [=](cellector_ctx *s, uint64_t b) {
        return cellector_iter_cell_outputs(s, ll ? ll + b : nullptr, ell ? ell + b : nullptr, nl ? nl + b : nullptr,
                                           norm ? norm + b : nullptr);
    });
}
cellector_status multi_cell_log_likelihoods(cellector_ctx *root, const double *alpha, const double *beta, const uint8_t *mask,
                                            double *ll, double *ell, double *nl)
{
    return per_cell(root, [=](cellector_ctx *s, uint64_t b) {
        return cellector_cell_log_likelihoods(s, alpha, beta, mask, ll ? ll + b : nullptr, ell ? ell + b : nullptr, nl ? nl + b : nullptr);
    });
}
cellector_status multi_posteriors(cellector_ctx *root, double *posterior, double *doublet, double *ll_maj, double *ll_min)
{
    return per_cell(root, [=](cellector_ctx *s, uint64_t b) {
        return cellector_posteriors(s, posterior ? posterior + b : nullptr, doublet ? doublet + b : nullptr,
                                    ll_maj ? ll_maj + b : nullptr, ll_min ? ll_min + b : nullptr);
    });
}

cellector_status multi_csr_rows(const cellector_ctx *root, uint64_t rb, uint64_t re, uint64_t *row_ptr, uint64_t *entries, uint64_t capacity)
{
    const auto &sh = root->multi->shards;
    if (rb > re || !row_ptr || re > sh[0]->total_cells) return ctx_fail(root, CELLECTOR_EINVAL, "bad row range");
    uint64_t base = 0;  // entries written so far
    row_ptr[0] = 0;
    for (const cellector_ctx *s : sh) {
        const uint64_t b = std::max(rb, s->cell_begin), e = std::min(re, s->cell_end);
        if (b >= e) continue;
        std::vector<uint64_t> rp(e - b + 1);
        cellector_status st = cellector_csr_rows(s, b - s->cell_begin, e - s->cell_begin, rp.data(), nullptr, 0);
        if (st == CELLECTOR_OK && entries) {
            if (capacity < base + rp.back()) return ctx_fail(root, CELLECTOR_EINVAL, "entries capacity too small");
            st = cellector_csr_rows(s, b - s->cell_begin, e - s->cell_begin, rp.data(), entries + base, capacity - base);
        }
        if (st != CELLECTOR_OK) {
            root->err = s->err;
            return st;
        }
        for (uint64_t i = 1; i < rp.size(); i++) row_ptr[b - rb + i] = base + rp[i];
        base += rp.back();
    }
    return CELLECTOR_OK;
}

// ---- the loop ------------------------------------------------------------------------------------------------------------------
cellector_status multi_em_iteration(cellector_ctx *root, double iqr_multiple, cellector_iter_summary *out)
{
    MultiCtx *m = root->multi;
    std::vector<cellector_iter_summary> sums(m->shards.size());
    cellector_iter_summary *sp = sums.data();
    const cellector_status st = run_all(root, [=](cellector_ctx *s, int r) { return cellector_em_iteration(s, iqr_multiple, sp + r); });
    if (st == CELLECTOR_OK && out) *out = sums[0];  // (identical on every shard: all of them see the exchanged buffers)
    return st;
}

cellector_status multi_final_allele_tallies(cellector_ctx *root, uint64_t *alt_min, uint64_t *ref_min, uint64_t *alt_maj, uint64_t *ref_maj)
{
    MultiCtx *m = root->multi;
    const uint64_t TL = m->shards[0]->total_loci;
    const size_t n = m->shards.size();
    std::vector<std::vector<uint64_t>> part(n, std::vector<uint64_t>(4 * TL));
    auto *pp = &part;
    const cellector_status st = run_all(root, [=](cellector_ctx *s, int r) {
        uint64_t *p = (*pp)[(size_t)r].data();
        return cellector_final_allele_tallies(s, p, p + TL, p + 2 * TL, p + 3 * TL);
    });
    if (st != CELLECTOR_OK) return st;
    uint64_t *outs[4] = {alt_min, ref_min, alt_maj, ref_maj};
    for (int k = 0; k < 4; k++) {
        if (!outs[k]) continue;
        for (uint64_t l = 0; l < TL; l++) {
            uint64_t a = 0;
            for (size_t r = 0; r < n; r++) a += part[r][(size_t)k * TL + l];
            outs[k][l] = a;
        }
    }
    return CELLECTOR_OK;
}

cellector_status multi_engine_info(const cellector_ctx *root, cellector_engine_info_t *o)
{
    memset(o, 0, sizeof *o);
    for (const cellector_ctx *s : root->multi->shards) {
        cellector_engine_info_t e;
        const cellector_status st = cellector_engine_info(s, &e);
        if (st != CELLECTOR_OK) return st;
        o->engine = e.engine;
        o->nnz_regular += e.nnz_regular; o->nnz_overflow += e.nnz_overflow; o->tile_bytes += e.tile_bytes;
        o->cell_blocks += e.cell_blocks; o->tile_lookups += e.tile_lookups;
        o->locus_chunks = e.locus_chunks; o->chunk_groups = e.chunk_groups;
    }
    return CELLECTOR_OK;
}

cellector_status multi_order_statistics(cellector_ctx *root, const double *keys, uint64_t n, double iqr_multiple, double *out3)
{
    if (!n || !keys) return ctx_fail(root, CELLECTOR_EINVAL, "order statistics: no keys");
    const int ns = (int)root->multi->shards.size();
    const uint64_t per = comm_cells_per_rank(n, ns);
    // (always the sharded select: the gathering form works on the loaded matrix's NORM buffers)
    return run_all(root, [=](cellector_ctx *s, int rank) {
        const uint64_t b = std::min(n, (uint64_t)rank * per), e = std::min(n, b + per);
        const int keep = s->sharded_select;
        s->sharded_select = 1;
        const cellector_status st = ffi_order_statistics(s, keys + b, e - b, n, iqr_multiple, rank == 0 ? out3 : nullptr);
        s->sharded_select = keep;
        return st;
    });
}

cellector_status multi_reset_timing(cellector_ctx *root)
{
    for (cellector_ctx *s : root->multi->shards) (void)cellector_reset_timing(s);
    return CELLECTOR_OK;
}

cellector_ctx *multi_shard0(const cellector_ctx *root) { return root->multi->shards[0]; }
