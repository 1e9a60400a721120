/*
 * cellector_ffi.h — C ABI of libcellector_hip.so: the MI355X (gfx950) implementation of
 * cellector's genotype-likelihood / EM scoring path.
 *
 * The reference (wheaton5/cellector, Rust, single-threaded CPU) has no FFI of its own; the seams
 * below are the reference's own function boundaries, so a Rust host replaces each call with one
 * `extern "C"` call (binding shown in INTEGRATION.md).  Citations are file:line under
 * /root/reference/cellector/src/.
 *
 * Conventions
 *  - every function returns a cellector_status and never throws or aborts across the boundary;
 *    cellector_last_error(ctx) gives the message (the reference panics -> stderr + exit 101);
 *  - plain pointers and sizes only; host output buffers are caller-allocated and caller-owned;
 *  - a ctx owns all device memory, is bound to one GPU, and is driven by one host thread at a time;
 *  - a ctx holds ONE SHARD of the matrix: the cells [cell_begin, cell_end) of the global cell
 *    range, all loci.  Per-locus state is replicated on every shard.  Multi-GPU: either ONE ctx over
 *    several devices (cellector_create_multi) or one process per GPU with a communicator attached to
 *    its ctx (cellector_comm_init_rank) — in both the library runs the three exchanges itself over
 *    RCCL/xGMI; or, without a communicator, the host sums the three exchange buffers below at the
 *    marked points (cellector_set_shard + bind_exchange_buffer: e.g. torch.distributed tensors).  With
 *    a single shard nothing is exchanged and cellector_em_iteration() runs the phases back to back;
 *  - every kernel is launched on the ctx's stream (default: the null stream).
 */
#ifndef CELLECTOR_FFI_H
#define CELLECTOR_FFI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cellector_ctx cellector_ctx;

typedef enum {
    CELLECTOR_OK = 0,
    CELLECTOR_EINVAL = 1,  /* bad argument / call out of order            */
    CELLECTOR_EIO = 2,     /* cannot open/read a file (reader, load_data.rs:240-251) */
    CELLECTOR_EPARSE = 3,  /* malformed mtx text (read_mtx_lines, load_data.rs:190-204) */
    CELLECTOR_ENOMEM = 4,
    CELLECTOR_EDEVICE = 5, /* HIP runtime error                            */
    CELLECTOR_ECOMM = 6
} cellector_status;

/* ---- lifecycle ----------------------------------------------------------------------------- */
cellector_status cellector_create(cellector_ctx **out, int device_id);
void cellector_destroy(cellector_ctx *ctx);
/* number of visible GPUs (0 without one; never fails the process) */
cellector_status cellector_device_count(int *out);

/* ---- multi-GPU (the reference is ONE binary, main.rs:25-50: the host sees one logical matrix) ---------------
 * (1) One process, several GPUs.  The ctx owns one shard per listed device (rank r = the r-th of n equal contiguous
 *     cell ranges; per-locus state replicated), one host thread and one stream per shard, and an RCCL communicator
 *     over the devices (ncclCommInitAll).  EVERY entry point below works on it unchanged — ingest, cellector_em_iteration,
 *     outputs, posteriors, final tallies — and returns arrays in GLOBAL cell order; the three exchanges (PASS1 at load,
 *     NORM all-gather and LOCUS all-reduce per iteration) run inside the library over xGMI.  Not available on it:
 *     cellector_set_stream / set_shard / bind_exchange_buffer / em_begin|threshold|finish (internal) and
 *     cellector_write_staged_mtx.  n_devices == 1 gives a plain ctx.  A device listed more than once gives logical
 *     shards on that GPU, exchanged by device-side sums instead of RCCL (which refuses duplicate devices): the same
 *     sharded code path on a one-GPU box (tests, rehearsals).  At most 16 shards. */
cellector_status cellector_create_multi(cellector_ctx **out, const int *device_ids, int n_devices);
/* (2) One process per GPU (torch.distributed / MPI launchers).  Rank 0 makes an id, the host broadcasts its 128 bytes by
 *     whatever means it has, every rank attaches a communicator to its own single-device ctx BEFORE the ingest.  The ctx
 *     then owns rank r's cell range and performs the exchanges itself: cellector_ingest_finish all-reduces PASS1,
 *     cellector_em_threshold first all-gathers NORM, cellector_em_finish first all-reduces LOCUS (so cellector_em_iteration
 *     is the whole distributed iteration); per-cell outputs are this rank's cells.  CELLECTOR_ECOMM on RCCL failures. */
cellector_status cellector_comm_unique_id(void *out_128_bytes);
cellector_status cellector_comm_init_rank(cellector_ctx *ctx, const void *unique_id_128_bytes, int n_ranks, int rank);
const char *cellector_last_error(const cellector_ctx *ctx); /* ctx-owned, valid until next call */
const char *cellector_version(void);
/* hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream); NULL = null stream. */
cellector_status cellector_set_stream(cellector_ctx *ctx, void *hip_stream);
/* Options: "compute_expected" (default 1: also accumulate expected_log_beta_binomial_pmf,
 * stats.rs:8-33, into expected_ll like the reference; 0 = skip that diagnostic column),
 * "timing" (default 0; 1: record HIP events around every timed group of kernels; 2: only around
 * the engine's dominant kernel — each event pair idles the queue for a few microseconds; 3: ... around every
 * fourth launch of it only),
 * "keep_coo" (default 1: keep the staged all-loci COO for cellector_final_allele_tallies),
 * "engine" (default 2: table-driven passes over the tiled 16-bit layout; 1: CSR/CSC kernels that
 * evaluate every entry's log-pmf — same results within rounding, kept for A/B checks; choose it before the
 * ingest: an engine-2 ingest releases the packed by-locus CSC that only engine 1 streams),
 * "compact_bits" (default 0: the locus pass stores 24-bit entries when the shard has <= 2^20 cells,
 * else 32-bit; 32 forces the wide form — set before ingest),
 * "locus_mode" (engine 2, default 0: per iteration the device picks how the per-locus minority counts
 * of get_locus_log_likelihoods, main.rs:368-420, are formed — 2 = walk only the excluded cells' rows,
 * 1 = stream the whole compact CSC past the exclusion bitmask; bit-identical results),
 * "overlap" (engine 2, default 1: the kernels of the few entries with alt+ref = 0 or > 4 run on a side
 * stream beside the table-lookup kernel; 2 = their locus-side part only after that kernel; 0 = everything
 * in one stream; same results to the bit),
 * "side_lds" (engine 2, default -1: automatic residency throttle of the side-stream kernels),
 * "ovf_deep" (engine 2, default -1: per matrix — when more than 3 % of the entries have alt+ref = 0 or > 4 (deep coverage)
 * their per-cell sums come from ONE unthrottled kernel that takes totals up to 17; 0 / 1 force either form; results agree
 * within rounding: the summation order inside a cell's overflow entries differs; "ovf_deep_wide", default 1: that kernel
 * gives 16 lanes to a row, 0 = a thread per row — A/B),
 * "tile_groups" (engine 2, default 0: the number of locus-chunk groups of the tile kernel is chosen per matrix;
 * 1..64 forces it — set before ingest; results may differ in the last bit),
 * "parse_window" (default 0: a text file of 1 GB or more is uploaded and tokenised in 256 MB windows, a smaller
 * one whole; a positive value forces windows of that many bytes — tests; lines of a windowed file may be 1 MB long),
 * "synth_continue_pct" (default 30: cellector_ingest_synthetic draws an entry's total as 1 + Geometric(0.7), vartrix-like
 * shallow coverage; a larger value gives deeper counts, e.g. 60 = 1 + Geometric(0.4) — benchmarks of the count distribution),
 * "norm_zero" (default 1: a shard clears the other shards' slices of CELLECTOR_XCHG_NORM before it writes
 * its own, so that a SUM all-reduce completes the array; 0 when the caller all-gathers the slices),
 * "sharded_select" (1: a ctx with a communicator — cellector_create_multi, cellector_comm_init_rank — finds the
 * median / quartiles by a radix select over the shards' own keys, exchanging digit histograms: six all-reduces of 48 KB per
 * iteration; 0: every shard's normalised LLs are all-gathered and every shard selects over all of them; default -1: the
 * histograms for runs of 4 Mi cells and more on three or more ranks, else the gather; same bits either way),
 * "balance" (multi-device ctx, default 1: see cellector_set_partition), "ref_arith" (engine 1: every entry with the
 * reference's own ln_gamma arithmetic, stats.rs:41-53, instead of the exact product form), "t2" / "t2_waves" (engine 2:
 * the entries with totals 5..8 through per-(locus, pair) tables, default on unless the matrix has deep coverage),
 * "bank_order" (engine 2, default 1: the tile builder orders every row's entries and the rows of a slice against LDS bank
 * conflicts; 0 keeps file order, the layout whose per-cell sums do not depend on which cells share a shard),
 * "t2_tiles" (engine 2, deep coverage: the cell side of the totals 5..8 (8, the default of a matrix with more than 3 % of
 * its entries outside 1..4), or 5..6 (6), walks a second tile set with chunk tables in LDS; 0: evaluated entry by entry). */
cellector_status cellector_set_option(cellector_ctx *ctx, const char *key, int64_t value);

/* ---- sharding (before ingest) --------------------------------------------------------------- */
/* This ctx owns global cells [cell_begin, cell_end).  Default: all cells. */
cellector_status cellector_set_shard(cellector_ctx *ctx, uint64_t cell_begin, uint64_t cell_end);
/* A ctx with a communicator (cellector_create_multi, cellector_comm_init_rank) shards the cells itself: rank r owns
 * [bounds[r], bounds[r+1]).  Default: n equal contiguous ranges — except that the text / COO ingest of a multi-device ctx
 * cuts the ranges so that every shard holds about the same number of ENTRIES (option "balance", default 1): the reference's
 * per-cell lists (load_data.rs:151-174) differ in length by orders of magnitude on real data, and the slowest shard sets
 * the iteration.  cellector_set_partition gives the ranges explicitly (n_ranks + 1 non-decreasing boundaries from 0 to
 * total_cells, the same array on every rank, before the ingest; NULL / 0 = back to the default); cellector_partition
 * reads the ranges in use back (bounds_out may be NULL to ask for the number of ranks only). */
cellector_status cellector_set_partition(cellector_ctx *ctx, const uint64_t *bounds, int n_bounds);
cellector_status cellector_partition(const cellector_ctx *ctx, uint64_t *bounds_out /*[n_ranks + 1]*/, int *n_ranks);

/* ---- ingest: replaces load_cell_data (load_data.rs:134-181) + get_loci_used (:254-280) ------- */
/* Phase 1 — stage this shard's entries on the device and count, per locus, cells with ref>0 /
 * alt>0 (pass 1, load_data.rs:265-270) and the allele totals into CELLECTOR_XCHG_PASS1. */
cellector_status cellector_ingest_mtx(cellector_ctx *ctx, const char *alt_path, const char *ref_path);
/* Caller COO in file order, 0-based indices (any order; locus-major like vartrix is fastest). */
cellector_status cellector_ingest_coo(cellector_ctx *ctx, uint64_t total_loci, uint64_t total_cells,
                                      uint64_t nnz, const uint32_t *locus0, const uint32_t *cell0,
                                      const uint32_t *alt, const uint32_t *ref);
/* Deterministic synthetic vartrix-like matrix generated on the device (benchmarks; definition in
 * DESIGN.md / cellector_amd/synth.py, which produces bit-identical host data). */
cellector_status cellector_ingest_synthetic(cellector_ctx *ctx, uint64_t total_loci,
                                            uint64_t total_cells, double density, uint64_t seed,
                                            double minority_fraction, double doublet_fraction);
/* Benchmark utility, not a reference seam: writes the staged matrix (any ingest, option keep_coo=1)
 * as a vartrix-style alt.mtx / ref.mtx text pair (3-line header, `locus cell count`, 1-based, file
 * order), formatted on the device — BASELINE-sized inputs for the text path in seconds. */
cellector_status cellector_write_staged_mtx(cellector_ctx *ctx, const char *alt_path, const char *ref_path);
/* >>> multi-shard: all-reduce CELLECTOR_XCHG_PASS1 here <<< */
/* Phase 2 — locus filter `cells_ref >= min_ref && cells_alt >= min_alt` (load_data.rs:273),
 * compaction, CSR (by cell) + CSC (by locus) build on the device (pass 2, load_data.rs:151-174). */
cellector_status cellector_ingest_finish(cellector_ctx *ctx, uint64_t min_alt, uint64_t min_ref);
/* Single-shard conveniences = ingest + finish. */
cellector_status cellector_load_mtx(cellector_ctx *ctx, const char *alt_path, const char *ref_path,
                                    uint64_t min_alt, uint64_t min_ref);
cellector_status cellector_load_coo(cellector_ctx *ctx, uint64_t total_loci, uint64_t total_cells,
                                    uint64_t nnz, const uint32_t *locus0, const uint32_t *cell0,
                                    const uint32_t *alt, const uint32_t *ref, uint64_t min_alt,
                                    uint64_t min_ref);

typedef struct {
    uint64_t total_cells, total_loci; /* header dims (consume_mtx_header, load_data.rs:206-223) */
    uint64_t loci_used;               /* L = loci passing the filter                            */
    uint64_t cell_begin, cell_end;    /* this shard                                             */
    uint64_t nnz_used;                /* this shard's entries at used loci                      */
} cellector_dims_t;
cellector_status cellector_dims(const cellector_ctx *ctx, cellector_dims_t *out);
cellector_status cellector_locus_ids(const cellector_ctx *ctx, uint64_t *out /*[L]*/);
/* locus_counts of load_cell_data: out[2l] = sum ref, out[2l+1] = sum alt (load_data.rs:157-158) */
cellector_status cellector_locus_counts(const cellector_ctx *ctx, double *out /*[2L]*/);
/* cell.cell_loci_data.len() for the min_loci_for_assignment rule (main.rs:153) */
cellector_status cellector_entries_per_cell(const cellector_ctx *ctx, uint32_t *out /*[local cells]*/);
/* CSR rows of local cells [row_begin,row_end): row_ptr rebased to 0, entries packed
 * locus_index | alt << 32 | ref << 48 (diagnostics, tests, CPU-baseline sampling). */
cellector_status cellector_csr_rows(const cellector_ctx *ctx, uint64_t row_begin, uint64_t row_end,
                                    uint64_t *row_ptr /*[rows+1]*/, uint64_t *entries, uint64_t capacity);

/* ---- exchange buffers (device memory, f64) --------------------------------------------------- */
typedef enum {
    CELLECTOR_XCHG_PASS1 = 0, /* [5*total_loci]: cells_ref | cells_alt | sum_ref | sum_alt | n_entries */
    CELLECTOR_XCHG_NORM = 1,  /* [total_cells]: normalised LL of every cell (own slice written, rest 0) */
    CELLECTOR_XCHG_LOCUS = 2  /* [5*L+8]: contrib_min | contrib_maj | cells_min | alt_min | ref_min |
                                 {n_new, n_rescued, n_excluded, ...}                               */
} cellector_xchg;
cellector_status cellector_exchange_buffer(cellector_ctx *ctx, cellector_xchg which, void **dev_ptr,
                                           uint64_t *n_f64);
/* Use caller-allocated device memory (e.g. a torch tensor) for an exchange buffer; must be called
 * before the buffer is first used (PASS1: after cellector_set_shard; NORM/LOCUS: after ingest_finish
 * sizes are known via cellector_exchange_buffer with dev_ptr == NULL). */
cellector_status cellector_bind_exchange_buffer(cellector_ctx *ctx, cellector_xchg which,
                                                void *dev_ptr, uint64_t n_f64);

/* ---- one EM iteration == compute_new_excluded (main.rs:308-347) ------------------------------ */
typedef struct {
    int32_t any_change;                /* main.rs:335                                            */
    uint64_t n_new_excluded, n_rescued; /* main.rs:333-334                                       */
    uint64_t n_excluded;               /* |new exclusion set| over all shards                    */
    uint64_t n_loci_filtered;          /* loci newly masked by the -80 filter (main.rs:444-447)  */
    double median, iqr, threshold;     /* main.rs:325-329                                        */
    /* Not in the reference: cells (all shards) with |normalised LL - threshold| <= band * max(1, |threshold|),
     * band = max(1e-9, 8 * 2^-52 * lgamma(max over the used loci of S_alt + S_ref + 2)).
     * The device evaluates log_beta_binomial_pmf as an exact product ratio; the reference's ln_gamma differences
     * (stats.rs:41-53) carry ~2^-52 * lnGamma(alpha + beta) of cancellation error per term — 1e-11 on a normalised
     * LL at vartrix-like depth (alpha + beta ~ 1e4: the band is its 1e-9 floor), 1e-8 at alpha + beta ~ 1e6 — so such
     * a cell could fall on the other side of main.rs:330-332's strict `<` in the reference.  Non-zero = the
     * bit-identical-assignment claim does not cover those cells of this iteration.  host/cellector prints one
     * stderr warning.  (Option ref_arith, engine 1, evaluates the reference's own formula instead.) */
    uint64_t n_near_threshold;
} cellector_iter_summary;

/* phase A: init_alpha_betas (main.rs:598-611) from the previous exclusion set's tallies, then
 * get_cell_log_likelihoods (main.rs:541-591) over this shard's cells and the normalisation of
 * main.rs:314-323 into this shard's slice of CELLECTOR_XCHG_NORM. */
cellector_status cellector_em_begin(cellector_ctx *ctx);
/* >>> multi-shard: all-reduce CELLECTOR_XCHG_NORM here <<< */
/* phase B: exact median / R-8 quartiles over all cells (statrs Data, main.rs:324-327), threshold
 * (main.rs:328-329), new exclusion flags of this shard's cells (main.rs:330-332) and this shard's
 * part of get_locus_log_likelihoods (main.rs:368-420) into CELLECTOR_XCHG_LOCUS. */
cellector_status cellector_em_threshold(cellector_ctx *ctx, double iqr_multiple);
/* >>> multi-shard: all-reduce CELLECTOR_XCHG_LOCUS here <<< */
/* phase C: locus filter (main.rs:428-451), any_change, state swap. */
cellector_status cellector_em_finish(cellector_ctx *ctx, cellector_iter_summary *out);
/* single shard: A, B, C back to back */
cellector_status cellector_em_iteration(cellector_ctx *ctx, double iqr_multiple,
                                        cellector_iter_summary *out);

/* outputs of the last iteration (host buffers; any pointer may be NULL) */
cellector_status cellector_iter_cell_outputs(const cellector_ctx *ctx, double *ll, double *expected_ll,
                                             double *loci_used_per_cell,
                                             double *normalized /*[local cells] each*/);
/* LocusLogLikelihoodData (main.rs:516-525) after the exchange, global over all shards */
cellector_status cellector_iter_locus_outputs(const cellector_ctx *ctx, double *contrib_min,
                                              double *contrib_maj, uint64_t *cells_min,
                                              uint64_t *cells_maj, uint64_t *alt_min,
                                              uint64_t *ref_min, uint64_t *alt_maj,
                                              uint64_t *ref_maj /*[L] each*/);
cellector_status cellector_loci_mask(const cellector_ctx *ctx, uint8_t *out /*[L]*/);
cellector_status cellector_excluded(const cellector_ctx *ctx, uint8_t *out /*[local cells]*/);
/* alpha/beta that the NEXT em_begin will use = init_alpha_betas(current excluded), main.rs:598 */
cellector_status cellector_alpha_betas(const cellector_ctx *ctx, double *alpha, double *beta /*[L]*/);

/* get_cell_log_likelihoods (main.rs:541-591) alone under caller alpha/beta/mask (host arrays). */
cellector_status cellector_cell_log_likelihoods(cellector_ctx *ctx, const double *alpha,
                                                const double *beta, const uint8_t *mask /*[L] or NULL*/,
                                                double *ll, double *expected_ll,
                                                double *loci_used_per_cell /*[local cells]*/);

/* ---- calculate_posteriors (main.rs:228-280) with the current exclusion set -------------------- */
cellector_status cellector_posteriors(cellector_ctx *ctx, double *posterior, double *doublet_posterior,
                                      double *ll_majority, double *ll_minority /*[local cells]*/);

/* ---- load_mtx_final (load_data.rs:109-132): per-locus allele tallies over ALL loci split by the
 * current exclusion set, for output_final_vcf (main.rs:52-131).  This shard's cells only; sum
 * across shards on the host. */
cellector_status cellector_final_allele_tallies(cellector_ctx *ctx, uint64_t *alt_min, uint64_t *ref_min,
                                                uint64_t *alt_maj, uint64_t *ref_maj /*[total_loci]*/);

/* layout facts of the loaded shard (benchmark accounting) */
typedef struct {
    uint64_t engine;
    uint64_t nnz_regular;   /* entries with 1 <= alt+ref <= 4: handled by table lookup in the tiled passes */
    uint64_t nnz_overflow;  /* the rest: evaluated individually                                           */
    uint64_t tile_bytes;    /* bytes of the tiled cell-pass layout                                        */
    uint64_t cell_blocks, locus_chunks, chunk_groups;
    uint64_t tile_lookups;  /* table lookups one pass of the tile kernel performs: the regular entries plus the padding
                               of the sliced-ELLPACK rows (each lookup = one or two 8-byte LDS reads)              */
} cellector_engine_info_t;
cellector_status cellector_engine_info(const cellector_ctx *ctx, cellector_engine_info_t *out);

/* ---- the order statistics by themselves ----------------------------------------------------- */
/* Median, interquartile range and threshold (statrs Data::median / quantile and main.rs:324-329) of n caller-supplied keys
 * (host array, no NaN): what the scoring loop computes from the normalised log-likelihoods, as a call of its own (tests of the
 * select on keys no run produces; no matrix needs to be loaded).  A multi-device ctx spreads the keys over its shards the
 * way it spreads cells and runs the sharded select (see option sharded_select).  out3 = {median, iqr, threshold}. */
cellector_status cellector_order_statistics(cellector_ctx *ctx, const double *keys, uint64_t n, double iqr_multiple,
                                            double *out3);

/* ---- timing of the dominant kernels (HIP events on the ctx stream; option "timing") ---------- */
typedef enum {
    CELLECTOR_K_CELL_LL = 0,     /* per-cell log-likelihood pass over the CSR  */
    CELLECTOR_K_LOCUS_STATS = 1, /* per-locus pass over the CSC                */
    CELLECTOR_K_SELECT = 2,      /* order statistics                           */
    CELLECTOR_K_POSTERIOR = 3,   /* fused 3-distribution pass + posteriors     */
    CELLECTOR_K_TILE_LL = 4,     /* engine 2: the tiled table-lookup kernel alone (inside K_CELL_LL) */
    CELLECTOR_K_COUNT = 5
} cellector_kernel_id;
cellector_status cellector_kernel_time(cellector_ctx *ctx, cellector_kernel_id which,
                                       double *total_ms, uint64_t *launches);
cellector_status cellector_reset_timing(cellector_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
