/*
 * cellector_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 *
 * Plain-C, single-threaded restatement of the hot path of wheaton5/cellector
 * (cellector/src/{main,stats,load_data}.rs) plus the statrs 0.16.0 routines
 * that path calls (ln_gamma, ln_factorial/ln_binomial, Data::median/quantile,
 * Binomial::pmf).  statrs is an un-vendored Cargo dependency
 * (cellector/Cargo.lock:443-445), so its published algorithm is restated here
 * from SURVEY.md Appendix B.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors
 * and cannot be built in this environment (no cargo/rustc).  This oracle is
 * pinned only by builder-generated known-answer values (tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this library.  The product (cellector_amd/, host/) never does.
 */
#ifndef CELLECTOR_ORACLE_H
#define CELLECTOR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- statrs 0.16.0 restatements --------------------- */
double orc_ln_gamma(double x);                       /* B.1 */
double orc_ln_factorial(uint64_t x);                 /* B.2 */
double orc_ln_binomial(uint64_t n, uint64_t k);      /* B.2 */
double orc_median(const double *x, size_t n);        /* B.3 Data::median          */
double orc_quantile(const double *x, size_t n, double tau); /* B.3 Data::quantile (R-8) */
double orc_binomial_pmf(double p, uint64_t n, uint64_t k);  /* B.4 */

/* ---- stats.rs ------------------------------------------------------------ */
double orc_logsumexp(double a, double b);                               /* stats.rs:35-39 */
double orc_log_beta_calc(double a, double b);                           /* stats.rs:48-53 */
double orc_log_beta_binomial_pmf(double alt, double ref, double alpha,
                                 double beta, double ln_coefficient);   /* stats.rs:41-46 */
void orc_expected_log_beta_binomial_pmf(size_t total, double alpha, double beta,
                                        double *expected, double *variance); /* stats.rs:8-33 */

/* ---- data model (load_data.rs) ------------------------------------------ */
typedef struct orc_ctx orc_ctx;

/* load_cell_data + get_loci_used over two MatrixMarket files (.gz by extension).
 * Returns NULL on error with a message in err. */
orc_ctx *orc_load_mtx(const char *alt_path, const char *ref_path,
                      uint64_t min_alt, uint64_t min_ref, char *err, size_t errlen);

/* Same semantics from caller COO arrays in file order (0-based indices). */
orc_ctx *orc_from_coo(uint64_t total_loci, uint64_t total_cells, uint64_t nnz,
                      const uint32_t *locus0, const uint32_t *cell0,
                      const uint32_t *alt, const uint32_t *ref,
                      uint64_t min_alt, uint64_t min_ref, char *err, size_t errlen);

/* Direct construction from CSR rows over already-compacted loci; locus_counts
 * ([L][2] = {sum_ref, sum_alt}) supplied by the caller so that a cell SAMPLE
 * of a bigger matrix can be scored under the full matrix's alpha/beta
 * (cpu_baseline leg of bench.py).  packed entry = locus | alt<<32 | ref<<48. */
orc_ctx *orc_from_csr(uint64_t n_cells, uint64_t n_loci, const uint64_t *row_ptr,
                      const uint64_t *packed, const double *locus_counts);

void orc_free(orc_ctx *);

void orc_dims(const orc_ctx *, uint64_t *total_cells, uint64_t *total_loci,
              uint64_t *loci_used, uint64_t *nnz_used);
void orc_locus_ids(const orc_ctx *, uint64_t *out /*[L]*/);
void orc_locus_counts(const orc_ctx *, double *out /*[L][2] ref,alt*/);
void orc_entries_per_cell(const orc_ctx *, uint32_t *out /*[N]*/);
void orc_row_ptr(const orc_ctx *, uint64_t *out /*[N+1]*/);
void orc_entries(const orc_ctx *, uint32_t *locus_index, uint32_t *alt, uint32_t *ref,
                 double *lnc /*[nnz] each, CSR order*/);

/* ---- scoring loop (main.rs) ---------------------------------------------- */
typedef struct {
    int any_change;
    uint64_t n_new_excluded, n_rescued;
    uint64_t n_loci_filtered;          /* loci newly masked by the -80 filter */
    double median, iqr, threshold;
    double locus_median;               /* median printed by the filter message */
} orc_iter_summary;

/* One outer iteration == compute_new_excluded (main.rs:308-347).  State
 * (excluded set, loci mask, iteration counter) lives in the ctx. */
void orc_em_iteration(orc_ctx *, double iqr_multiple, orc_iter_summary *out);

/* Outputs of the last iteration. */
void orc_iter_cell_outputs(const orc_ctx *, double *ll, double *expected_ll,
                           double *loci_used_per_cell, double *normalized /*[N] each, may be NULL*/);
void orc_iter_locus_outputs(const orc_ctx *, double *contrib_min, double *contrib_maj,
                            uint64_t *cells_min, uint64_t *cells_maj,
                            uint64_t *alt_min, uint64_t *ref_min,
                            uint64_t *alt_maj, uint64_t *ref_maj /*[L] each, may be NULL*/);
void orc_loci_mask(const orc_ctx *, uint8_t *out /*[L]*/);
void orc_excluded(const orc_ctx *, uint8_t *out /*[N]*/);
void orc_set_excluded(orc_ctx *, const uint8_t *in /*[N]*/);
/* worker threads of the per-cell loop (default 1 = like the single-threaded reference; results are
 * bit-identical for any count) */
void orc_set_threads(int n);
int orc_get_threads(void);
void orc_alpha_betas(const orc_ctx *, double *alpha, double *beta /*[L]*/); /* init_alpha_betas(current excluded) */

/* get_cell_log_likelihoods alone (main.rs:541-591) under caller alpha/beta and
 * mask; materialises the 88-byte PMFData records like the reference.  Used by
 * the cpu_baseline timing leg. */
void orc_cell_log_likelihoods(orc_ctx *, const double *alpha, const double *beta,
                              const uint8_t *mask, double *ll, double *expected_ll,
                              double *loci_used_per_cell);

/* per-locus statistics (main.rs:368-420) of this ctx's cells under caller alpha/beta/mask and exclusion flags */
void orc_locus_stats(orc_ctx *, const double *alpha, const double *beta, const uint8_t *mask,
                     const uint8_t *new_excluded, double *contrib_min, double *contrib_maj, uint64_t *cells_min,
                     uint64_t *cells_maj, uint64_t *alt_min, uint64_t *ref_min, uint64_t *alt_maj, uint64_t *ref_maj);

/* calculate_posteriors (main.rs:228-280) with the ctx's current excluded set. */
void orc_posteriors(orc_ctx *, double *posterior, double *doublet_posterior,
                    double *ll_majority, double *ll_minority /*[N] each*/);

/* output_final_assignments rule (main.rs:141-171).  codes: 0 -> "0",
 * 1 -> "1", 2 -> "doublet", 3 -> "unassigned".  qual as the printed usize. */
void orc_assignments(const orc_ctx *, const double *posterior, const double *doublet_posterior,
                     double posterior_threshold, uint64_t min_loci_used,
                     uint8_t *posterior_assignment, uint8_t *anomaly_assignment,
                     uint64_t *qual /*[N] each*/);

/* load_mtx_final tallies (load_data.rs:109-132) over ALL loci from COO arrays
 * + genotype posterior rule of output_final_vcf (main.rs:79-124).
 * gt codes: 0 "./.", 1 "1/1", 2 "0/1", 3 "0/0". */
void orc_final_tallies_coo(uint64_t total_loci, uint64_t nnz,
                           const uint32_t *locus0, const uint32_t *cell0,
                           const uint32_t *alt, const uint32_t *ref,
                           const uint8_t *excluded,
                           uint64_t *alt_min, uint64_t *ref_min,
                           uint64_t *alt_maj, uint64_t *ref_maj /*[total_loci]*/);
void orc_vcf_genotype(uint64_t alt_min, uint64_t ref_min, uint64_t alt_maj, uint64_t ref_maj,
                      uint8_t *gt_maj, double *maxpost_maj, uint8_t *gt_min, double *maxpost_min);

#ifdef __cplusplus
}
#endif
#endif
