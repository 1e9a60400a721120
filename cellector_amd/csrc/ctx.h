// Internal state of a cellector_ctx (one shard on one GPU) and the launch wrappers that the
// C-ABI layer (cellector_ffi.cpp) calls.  Not part of the public ABI.
#pragma once
#include <cstring>

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cellector_ffi.h"
#include "comm.h"

// ---- packed matrix entry -------------------------------------------------------------------
// CSR (by cell):  bits 0..31 compact locus index, 32..47 alt count, 48..63 ref count
// CSC (by locus): bits 0..31 local cell index,    32..47 alt count, 48..63 ref count
#define ENT_IDX(e) ((uint32_t)((e) & 0xffffffffull))
#define ENT_ALT(e) ((uint32_t)(((e) >> 32) & 0xffffull))
#define ENT_REF(e) ((uint32_t)((e) >> 48))
#define CELLECTOR_MAX_COUNT 65535u

// LOCUS exchange buffer layout (f64): 5 planes of L then 8 counters
enum { LB_CONTRIB_MIN = 0, LB_CONTRIB_MAJ = 1, LB_CELLS_MIN = 2, LB_ALT_MIN = 3, LB_REF_MIN = 4, LB_PLANES = 5 };
enum { LC_N_NEW = 0, LC_N_RESCUED = 1, LC_N_EXCLUDED = 2, LC_N_NEAR = 3 /* cells within the near-tie band of the threshold */,
       LC_COUNTERS = 8 };
#define CELLECTOR_NEAR_TIE_REL 1e-9  // |norm - thr| <= band * max(1, |thr|) counts as a near-tie (cellector_iter_summary); the band's floor
// PASS1 exchange buffer layout (f64): 5 planes of total_loci
enum { P1_CELLS_REF = 0, P1_CELLS_ALT = 1, P1_SUM_REF = 2, P1_SUM_ALT = 3, P1_ENTRIES = 4, P1_PLANES = 5 };

// d_counters slots (u32)
enum { DC_N_FILTERED = 0, DC_N_MIN = 4 /* members of this shard's new exclusion set (k_flag) */ };

#define T_ROWS_PER_TILE 1024  // rows (cells) of a tile of the tiled layout (= T_BC in kernels_tiled.hip)
#define CELLECTOR_TILE_WORK_STRIDE 64  // column counters per table set of the persistent tile kernel (= T_GROUPS_MAX)

#define LF_TABLE_N 171  // ln(FCACHE[0..170]) — statrs ln_factorial cache, SURVEY Appendix B.2

struct KernelTimer {
    std::vector<hipEvent_t> start, stop;  // pending pairs
    double total_ms = 0.0;
    uint64_t launches = 0;
    uint64_t calls = 0;  // launches seen since the timing level was set (level 3 samples every fourth)
    bool open = false;   // timer_begin recorded a start event for the launch in progress
};

struct MultiCtx;  // multi.cpp: the shards and worker threads of a ctx made by cellector_create_multi

struct cellector_ctx {
    // a ROOT ctx (cellector_create_multi with more than one shard) owns no device state of its own: every entry point
    // fans out to its shards (multi.cpp) and returns arrays in global cell order
    MultiCtx *multi = nullptr;
    // the exchange transport of a shard that is part of a sharded run (n = 1: single shard, nothing is exchanged)
    Comm comm;
    bool owns_stream = false;  // the stream was created by the library (a shard of a root ctx)
    bool ingest_all_cells = false;  // begin_ingest: this shard stages ALL cells for now (multi-device text ingest parses once)
    int device = 0;
    hipStream_t stream = nullptr;
    // side stream for the small overflow kernels that run next to the tile kernel (fork/join with events)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_sum = nullptr;
    hipEvent_t ev_tab = nullptr;   // completion of the table kernel queued ahead by em_finish, attached to its dispatch (no barrier packet)
    bool tab_event_valid = false;  // ... and nothing the side stream depends on has been queued behind it since
    int overlap = 1;  // option "overlap": 1 = the overflow kernels run on the side stream next to the tile kernel, 0 = in front
    mutable std::string err;

    // options
    bool compute_expected = true;
    bool ref_arith = false;  // option ref_arith (engine 1): evaluate stats.rs:41-53 with ln_gamma differences, the reference's own rounding
    int64_t parse_window_opt = 0;  // option parse_window: 0 = whole file below 1 GB, 256 MB windows above; else the window in bytes
    bool fuse_filter = true;   // option "fuse_filter": an unsharded ctx applies the -80 locus filter inside k_locus_finalize (A/B)
    bool filter_fused = false;  // this iteration's locus pass did so (em_finish then launches no k_locus_filter)
    bool bank_order = true;  // option "bank_order": the tile builder orders every row's entries against LDS bank conflicts (tile_bank_order)
    int tile_groups_opt = 0;  // option tile_groups: 0 = chosen per matrix (tiled_setup), else forced (multiple of 8)
    // option sharded_select: a ctx with a communicator exchanges digit histograms (1) or all-gathers NORM (0); -1 = by the
    // number of ranks (use_sharded_select)
    int sharded_select = -1;
    bool norm_zero = true;  // option: clear the other shards' slices of NORM before the cell pass (needed by a sum exchange)
    int timing = 0;  // 0 off, 1 every timed region, 2 only the dominant kernel of the engine, 3 ... around every fourth launch
    bool keep_coo = true;
    double near_rel = CELLECTOR_NEAR_TIE_REL;  // near-tie band of this matrix, relative to max(1, |threshold|) (cellector_ingest_finish)
    int synth_continue_pct = 30;  // option synth_continue_pct: the synthetic generator's n = 1 + Geometric(1 - pct/100)

    // shard
    uint64_t cell_begin = 0, cell_end = UINT64_MAX;

    // dims
    uint64_t total_loci = 0, total_cells = 0, L = 0, nloc = 0, nnz = 0;
    enum { ST_EMPTY, ST_STAGED, ST_READY } state = ST_EMPTY;
    int em_phase = 0;  // 0 idle, 1 after begin, 2 after threshold

    // staged COO of this shard (all loci; cell index local)
    uint64_t coo_n = 0;
    uint32_t *coo_locus = nullptr, *coo_cell = nullptr;
    uint16_t *coo_alt = nullptr, *coo_ref = nullptr;
    bool coo_sorted = false;

    // matrix
    uint64_t *csr_ptr = nullptr, *csr_ent = nullptr;  // [nloc+1], [nnz]
    uint64_t *csc_ptr = nullptr, *csc_ent = nullptr;  // [L+1], [nnz]
    uint64_t *locus_ids = nullptr;                    // [L]
    double *s_alt = nullptr, *s_ref = nullptr, *n_ent = nullptr;  // [L] global totals
    uint64_t *to_used = nullptr;                      // [total_loci] compact index or ~0

    // per-locus loop state
    double2 *ab = nullptr;       // [L] alpha,beta for the running pass; alpha < 0 => locus masked
    double *ab6 = nullptr;       // [8L] posterior alpha/beta sets (min, maj, dbl, pad)
    uint8_t *mask = nullptr;     // [L] loci_used for the current iteration
    uint8_t *mask_next = nullptr;
    uint32_t *d_counters = nullptr;  // [8] device scratch counters
    // per-cell state
    uint8_t *flags = nullptr, *flags_new = nullptr;  // [nloc] exclusion set
    double *ll = nullptr, *ell = nullptr, *nloci = nullptr;  // [nloc]
    double *post = nullptr;  // [4*nloc] posterior, doublet, ll_maj, ll_min
    double *lf = nullptr;    // [LF_TABLE_N] ln factorial table

    // exchange buffers
    double *x_pass1 = nullptr, *x_norm = nullptr, *x_locus = nullptr;
    bool own_pass1 = true, own_norm = true, own_locus = true;
    uint64_t n_pass1 = 0, n_norm = 0, n_locus = 0;

    // ---- engine v2: table-driven tiled layout (kernels_tiled.hip) ----
    int engine = 2;                  // 1 = v1 CSR/CSC kernels, 2 = tiled (default)
    bool tiled_ready = false;
    uint32_t t_nb = 0, t_nj = 0, t_groups = 0, t_cpg = 0;  // cell blocks, locus chunks, chunk groups, chunks/group
    uint64_t t_npad = 0;             // nb * T_BC
    uint64_t *tile_ptr = nullptr;    // [nb*nj+1] offsets into tiles, in u16 elements (multiples of 128)
    uint16_t *tiles = nullptr;       // SELL-64-1024 slices: per tile 16 slices of 64 rows [cell, K entries code*384 + locus_in_chunk]
    uint16_t *thdr = nullptr;        // [nb*nj][T_HDR] tile headers: 16 x {first u16 of the slice, K}
    uint64_t t_elems = 0;            // entries (u16) in the stream, padding included
    double *tab_em = nullptr;        // table the last EM cell pass built (the locus pass reads its log-pmfs)
    int tab_em_stride = 1;           // 2 when that table holds (log-pmf, expected) pairs
    uint64_t *ovf_ptr = nullptr, *ovf_ent = nullptr;    // overflow CSR (alt+ref == 0 or > 4), packed like csr_ent
    uint64_t ovf_n = 0;
    double *ovf_tab = nullptr;       // [L][128] per-locus cumulative-log / expected tables for overflow entries
    double *ovf_etab = nullptr;      // [L][8] alpha, beta, E(n) for n = 5..8, pad: the cell side's 64-byte record per locus
    int side_lds = -1;               // option "side_lds": dynamic LDS bytes requested by the cell-side overflow kernel (residency
                                     // throttle; -1 = automatic)
    int ovf_deep_opt = -1;           // option "ovf_deep": -1 = decided per matrix (tiled_build), 0 / 1 = forced
    bool ovf_deep_wide = true;       // option "ovf_deep_wide": deep form with 16 lanes per row (0: a thread per row; A/B)
    bool ovf_deep = false;           // the overflow entries are a large share of the matrix (deep coverage): their cell side runs
                                     // the full form of the direct kernel (totals up to 17 in one kernel), never throttled
    bool ovf_locus_pending = false;  // the side stream still owes this iteration's locus-side overflow tables / values (event ev_join2)
    double *ovf_lp = nullptr;        // [ovf_n] shallow coverage: the EM pass' overflow log-pmfs, by-locus order (k_ovf_values -> k_locus_finalize)
    uint32_t *ovc_locus = nullptr;   // [ovf_n] compact locus index of every overflow entry, by-locus order (k_ovf_values)
    double *ovf_sum = nullptr;       // [3][2][nloc] per-cell sums of the overflow values (ll, expected) per table set
    uint64_t *ovf_ell_ptr = nullptr, *ovf_ell = nullptr;  // 64-row ELLPACK copy of the overflow CSR (cell side): [groups+1], slots
    uint32_t *ovf_tier_row[2] = {nullptr, nullptr};  // the overflow entries with alt+ref in 9..17 (tier 0) / above (tier 1):
    uint64_t *ovf_tier_ent[2] = {nullptr, nullptr};  //   their rows and packed entries, in row order
    uint64_t ovf_n_tier[2] = {0, 0};
    double *ovf_tier_val = nullptr;  // [2][ovf_n_tier[1]] per pass: log-pmf / expected term of the tier-1 entries (k_ovf_listed_values)
    uint32_t *ovf_nmask = nullptr;   // [L] which alt+ref totals (4..17) occur among the locus' overflow entries
    // tier 2 (kernels_tiled.hip, k_t2_tables): the overflow entries with totals 5..8 are table-driven as well
    int t2_opt = -1;                 // option "t2": -1 = decided per matrix (tiled_build: on unless the matrix is ovf_deep), 0 / 1 = forced
    bool t2 = false;
    int t2_waves = 512;              // option "t2_waves": one-wave blocks of k_t2_cell when it runs beside the tile kernel
    uint32_t *hist_all2 = nullptr;   // [L][32] tier-2 entries per (locus, pair), all cells of the shard (static)
    uint32_t *t2_plist = nullptr, *t2_slist = nullptr;  // the pairs (locus << 5 | pair) / table sectors (locus << 3 | sector) that occur, locus order (static)
    uint32_t t2_np = 0, t2_ns = 0;
    uint32_t *t2_pmask = nullptr;    // [L] bit c2: the pair occurs at the locus (static)
    uint32_t *cnt2 = nullptr;        // [L][32] ... of the cells of the new exclusion set (k_t2_minority; cleared by k_locus_finalize)
    double *tab2 = nullptr;          // [L][48] per pass: log-pmfs of the pairs that occur + expected terms, six 64-byte sectors per locus
    // tier-2 TILES (deep coverage; kernels_tiled.hip, geo_t2): the cell side of the totals 5..t2_tiles walks a second tile set with
    // its own chunk tables in LDS instead of evaluating those entries one by one (k_ovf_cell_wide keeps the other totals)
    int t2_tiles_opt = -1;           // option "t2_tiles": -1 = automatic (8 on an ovf_deep matrix), 0 = off, 6 / 8 = totals 5..6 / 5..8
    int t2_tiles = 0;                // 0, 6 or 8: in use (tiled_build)
    uint32_t t2_nj = 0, t2_groups = 0, t2_cpg = 0;  // chunks of geo_t2::BLU loci, chunk groups, chunks per group
    uint64_t *tile2_ptr = nullptr;   // [nb * t2_nj + 1]
    uint16_t *tiles2 = nullptr, *thdr2 = nullptr;
    double *tab2c = nullptr;         // [t2_nj][geo_t2::BL][geo_t2::LROW] per pass: the chunked tier-2 tables (+ tail pad)
    double *part2 = nullptr;         // [3][2][t2_groups][npad] per-group partial sums of the tier-2 tile passes
    uint32_t *tile_work2 = nullptr;  // [T_GROUPS_MAX] column counters of a tier-2 tile pass
    uint64_t *ovr_ptr = nullptr, *ovr_ent = nullptr;  // by-cell CSR of the overflow entries the tier-2 tiles leave out (totals 0, above t2_tiles)
    uint64_t ovr_n = 0;
    uint64_t *ovx_ptr = nullptr, *ovx_ent = nullptr;  // by-locus CSC of the overflow entries outside tier 2 (totals 0 and above 8)
    uint32_t *ovx_locus = nullptr;   // [ovx_n] their compact locus index
    double *ovx_lp = nullptr;        // [ovx_n] the EM pass' log-pmfs of those entries (k_ovx_values -> k_locus_finalize)
    uint64_t ovx_n = 0;
    uint64_t *c4_ptr = nullptr;      // [L+1] compact CSC of regular entries
    uint32_t *c4_ent = nullptr;      // 32-bit entries cell_local | code << 28, or 24-bit cell | code << 20 (c4_bits)
    int c4_bits = 32;
    int c4_bits_opt = 0;             // option "compact_bits": 0 = automatic, 32 = force the 32-bit entries
    uint64_t *ovc_ptr = nullptr, *ovc_ent = nullptr;    // overflow CSC, packed like csc_ent
    uint32_t *hist_all = nullptr;    // [L][14] regular entries per code
    double *tab = nullptr;           // [3 + 2][nj][15][384] log-pmf tables of the posterior sets, then the EM pass' (log-pmf, expected) pairs
    double *part = nullptr;          // [3][2][groups][npad] per-group partial sums (ll, ell)
    double2 *ab3 = nullptr;          // [3][L] posterior alpha/beta sets as double2
    uint32_t *masked_cnt = nullptr;  // [nloc] entries of the cell at masked loci
    uint32_t *flag_bits = nullptr;   // [ceil(nloc/32)] new exclusion set as a bitmask
    uint64_t n_masked_loci = 0;
    uint32_t *tile_work = nullptr;   // [3][T_GROUPS_MAX] column counters of the persistent tile kernel, one set per table set
    uint32_t *minlist = nullptr;     // [nloc] local ids of the cells of the new exclusion set (arbitrary order)
    uint32_t *hist_min = nullptr;    // [lr_sub][L][16] regular entries of minority cells per (locus, code), partial planes
    uint32_t *mroff = nullptr;       // [R+1][mroff_cap] the excluded cells' offset rows, transposed (per iteration)
    uint64_t *mbeg = nullptr;        // [mroff_cap] start of the excluded cells' rows in csr_ent
    uint64_t mroff_cap = 0;
    uint32_t lr_sub = 1;             // subsets of the exclusion set = partial planes of hist_min
    uint16_t *c4r = nullptr;         // [nnz] compact by-cell entries: locus inside its 4096-locus range | code << 12 (code 15: overflow entry)
    uint32_t *roff = nullptr;        // [nloc][R+1] offsets of the locus ranges inside each by-cell CSR row
    int locus_mode = 0;              // option "locus_mode": 0 = chosen on the device per iteration, 1 = stream the compact CSC,
                                     // 2 = minority-driven tally over the by-cell CSR

    // order-statistic workspace
    uint32_t *sel_hist = nullptr;   // [4096] top-bits histogram, [SEL_T][1024] next-bits histograms, [1] list length (+ pad)
    uint64_t *sel_state = nullptr;  // [2][SEL_T][2] prefix, remaining rank after the first / second step
    uint64_t *sel_list = nullptr;   // keys that carry a target's 22-bit prefix (capacity: all keys)
    uint64_t sel_list_cap = 0;
    double *sel_out = nullptr;      // [16] device: [0..5] order statistics, [8..10] median, iqr, threshold
    uint32_t *seld_hist = nullptr;  // sharded run: [6 levels][SEL_T][2048] digit histograms (select_threshold_sharded), made on first use
    uint64_t *seld_state = nullptr; // ... [7][SEL_T][2] prefix, remaining rank before / after every level
    double *h_sel = nullptr;        // pinned [32]: iteration summary written by k_iter_summary, read in em_finish
    double *h_sum_dev = nullptr;    // the device's address of h_sel
    uint64_t sum_seq = 0;           // number of summaries queued; h_sel[CELLECTOR_SUM_SEQ] = the last one that arrived
    bool tables_prebuilt = false;   // the next iteration's k_build_tables is already queued / done (em_finish)
    bool prebuilt_expected = false; // ... with this value of compute_expected
    bool work_zeroed = false;       // tile_work was reset by this iteration's k_alpha_beta
    bool cell_join_pending = false; // the main stream still has to wait for the side stream's cell-side overflow sums (ev_join)

    // iteration bookkeeping
    uint64_t iteration = 0;
    uint64_t n_excluded_global = 0;
    double last_median = 0, last_iqr = 0, last_thr = 0;
    bool have_iter = false;

    KernelTimer timers[CELLECTOR_K_COUNT];
    std::vector<hipEvent_t> ev_pool;  // collected timer events, reused (creating a pair costs microseconds before a launch)
};

#define SEL_T 6
#define CELLECTOR_SUM_SEQ 31
#define CELLECTOR_SEL_HIST_WORDS (4096 + SEL_T * 1024 + 64)

// ---- error plumbing ---------------------------------------------------------------------------
cellector_status ctx_fail(const cellector_ctx *c, cellector_status s, const char *fmt, ...);
#define HIPCHK(c, expr)                                                                          \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return ctx_fail((c), CELLECTOR_EDEVICE, "%s failed: %s (%s:%d)", #expr,              \
                            hipGetErrorString(e__), __FILE__, __LINE__);                         \
    } while (0)
#define CHK(expr)                                                                                \
    do {                                                                                         \
        cellector_status s__ = (expr);                                                           \
        if (s__ != CELLECTOR_OK) return s__;                                                     \
    } while (0)

// Device memory goes through a small caching layer (cellector_ffi.cpp): mapping fresh VRAM costs ~30-50 ms per GB on
// this platform once the footprint is large, and the ingest allocates and frees tens of GB of temporaries several times
// over (measured at 2e9 entries: 3.1 s of hipMalloc for 64 GB in the CSR build alone).  Freed blocks of >= 64 MB are kept
// and handed out again to requests of at least half their size; dev_cache_trim() returns them to the driver.
hipError_t dev_cache_malloc(void **p, size_t bytes);
void dev_cache_free(void *p);
void dev_cache_trim(int device = -1);  // -1: the cached blocks of every device
void dev_cache_park(void *p, size_t bytes, int device);  // a fresh, unused hipMalloc block for later requests of its size

template <typename T>
static inline cellector_status dev_alloc(cellector_ctx *c, T **p, uint64_t n)
{
    void *q = nullptr;
    hipError_t e = dev_cache_malloc(&q, (n ? n : 1) * sizeof(T));
    if (e != hipSuccess)
        return ctx_fail(c, CELLECTOR_ENOMEM, "hipMalloc(%llu bytes) failed: %s",
                        (unsigned long long)(n * sizeof(T)), hipGetErrorString(e));
    *p = (T *)q;
    return CELLECTOR_OK;
}
template <typename T>
static inline void dev_free(T *&p)
{
    if (p) dev_cache_free((void *)p);
    p = nullptr;
}

// Device-to-device copy between two shards (one device, or peers) on the RECEIVING shard's stream, complete on return.  (On the
// null stream such a copy may return before it is done, and the shards' non-blocking streams do not order against that stream:
// a kernel launched right behind it could read stale bytes.)
static inline hipError_t dev_copy_sync(hipStream_t st, void *dst, int dst_dev, const void *src, int src_dev, size_t bytes)
{
    if (!bytes) return hipSuccess;
    hipError_t e = dst_dev == src_dev ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st)
                                      : hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return e;
}

// ---- timing -----------------------------------------------------------------------------------
void timer_begin(cellector_ctx *c, int which);
void timer_end(cellector_ctx *c, int which);
bool timer_take(cellector_ctx *c, int which, hipEvent_t *start, hipEvent_t *stop);
void timer_collect(cellector_ctx *c);

// ---- launch wrappers (kernels_*.hip) ------------------------------------------------------------
// EM loop
cellector_status launch_alpha_beta(cellector_ctx *c);
cellector_status launch_cell_ll(cellector_ctx *c, const double2 *ab, double *norm_out /*may be null*/);
cellector_status launch_flag(cellector_ctx *c, const double *d_thr);
cellector_status launch_locus_stats(cellector_ctx *c);
cellector_status launch_locus_filter(cellector_ctx *c);
cellector_status launch_iter_summary(cellector_ctx *c);
cellector_status launch_ab_from_host(cellector_ctx *c, const double *alpha, const double *beta,
                                     const uint8_t *mask);
cellector_status launch_posteriors(cellector_ctx *c, double mf0, double lp_min, double lp_maj,
                                   double lp_dbl);
cellector_status launch_final_tallies(cellector_ctx *c, uint64_t *d_out /*[4*total_loci]*/);
// order statistics: exact values at SEL_T 0-based ranks of n keys
cellector_status select_threshold(cellector_ctx *c, const double *keys, uint64_t n, double iqr_multiple);
cellector_status ffi_order_statistics(cellector_ctx *c, const double *keys, uint64_t n_local, uint64_t n_total, double iqr_multiple,
                                      double *out3);  // (cellector_ffi.cpp; a shard's slice of cellector_order_statistics)
// ... over the keys of all shards of a sharded run (this shard holds n_local of the n_total), exchanged as digit histograms
cellector_status select_threshold_sharded(cellector_ctx *c, const double *keys, uint64_t n_local, uint64_t n_total, double iqr_multiple);
// ingest
cellector_status ingest_stage_host_coo(cellector_ctx *c, uint64_t nnz, const uint32_t *locus0,
                                       const uint32_t *cell0, const uint32_t *alt, const uint32_t *ref);
cellector_status ingest_pass1(cellector_ctx *c);
cellector_status ingest_split_coo(cellector_ctx *src, uint64_t cb, uint64_t ce, uint64_t *keep, uint32_t **o_locus, uint32_t **o_cell,
                                  uint16_t **o_alt, uint16_t **o_ref, uint64_t *n_out);
cellector_status ingest_cell_histogram(cellector_ctx *c, const uint32_t *d_cell, uint64_t n, uint64_t total_cells, std::vector<uint32_t> *out);
// multi-device text ingest (cellector_ffi.cpp): stage the whole pair on one shard / hand a shard its routed entries
cellector_status ffi_stage_mtx_all_cells(cellector_ctx *c, const char *alt_path, const char *ref_path, cellector_ctx *helper);
cellector_status ffi_adopt_staged(cellector_ctx *c, uint64_t total_loci, uint64_t total_cells, uint32_t *locus, uint32_t *cell,
                                  uint16_t *alt, uint16_t *ref, uint64_t n, bool sorted);
cellector_status ingest_build(cellector_ctx *c, uint64_t min_alt, uint64_t min_ref);
cellector_status synth_generate(cellector_ctx *c, double density, uint64_t seed, double minority_fraction,
                                double doublet_fraction);
cellector_status synth_write_mtx(cellector_ctx *c, const char *alt_path, const char *ref_path);
// device helpers
cellector_status dev_exclusive_scan_u64(cellector_ctx *c, uint64_t *data, uint64_t n, uint64_t *total_out_host);
cellector_status dev_sort_pairs_u32_u64(cellector_ctx *c, uint32_t *keys_in, uint32_t *keys_out,
                                        uint64_t *vals_in, uint64_t *vals_out, uint64_t n, int end_bit);
// engine v2 (kernels_tiled.hip)
cellector_status tiled_build(cellector_ctx *c);
void tiled_free(cellector_ctx *c);
cellector_status tiled_cell_pass(cellector_ctx *c, const double2 *ab, double *norm_out, bool for_em);
cellector_status tiled_locus_pass(cellector_ctx *c);
cellector_status tiled_masked_update(cellector_ctx *c);
cellector_status tiled_prebuild_tables(cellector_ctx *c);
cellector_status tiled_posteriors(cellector_ctx *c, double mf0, double lp_min, double lp_maj, double lp_dbl);
// device-side mtx text parse (kernels_parse.hip)
struct MtxInput;
cellector_status mtx_input_open(const cellector_ctx *c, const char *alt_path, const char *ref_path, MtxInput **out,
                                uint64_t *total_loci, uint64_t *total_cells);
void mtx_input_close(MtxInput *in);
// split ingest of a multi-device ctx (kernels_parse.hip): every shard tokenises a range of windows of both files
struct MtxSplit;
MtxSplit *mtx_split_new(int n_shards, LocalGroup *thread_barrier, bool balance /*cut the cells by entries, not by count*/);
void mtx_split_delete(MtxSplit *s);
bool mtx_input_windowed(const MtxInput *in, int64_t parse_window_opt);
cellector_status ingest_stage_mtx_split(cellector_ctx *c, MtxInput *in, MtxSplit *s, int rank, uint64_t parse_window, uint32_t **o_locus,
                                        uint32_t **o_cell, uint16_t **o_alt, uint16_t **o_ref, uint64_t *o_n, bool *o_sorted);
cellector_status ingest_stage_mtx_device(cellector_ctx *c, MtxInput *in, cellector_ctx *helper = nullptr /*parses the ref file*/);
