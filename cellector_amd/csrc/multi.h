// Entry points of a root ctx (cellector_create_multi; multi.cpp), called from the C-ABI layer's dispatch.
#pragma once
#include "ctx.h"

cellector_status multi_create(cellector_ctx **out, const int *device_ids, int n);
void multi_destroy(cellector_ctx *root);
int multi_n_shards(const cellector_ctx *root);
cellector_ctx *multi_shard0(const cellector_ctx *root);  // per-locus state is replicated: shard 0 answers for all
cellector_status multi_set_option(cellector_ctx *root, const char *key, int64_t v);
cellector_status multi_set_partition(cellector_ctx *root, const uint64_t *bounds, int n_bounds);
cellector_status multi_ingest_mtx(cellector_ctx *root, const char *alt_path, const char *ref_path);
cellector_status multi_ingest_coo(cellector_ctx *root, uint64_t total_loci, uint64_t total_cells, uint64_t nnz, const uint32_t *locus0,
                                  const uint32_t *cell0, const uint32_t *alt, const uint32_t *ref);
cellector_status multi_ingest_synthetic(cellector_ctx *root, uint64_t total_loci, uint64_t total_cells, double density, uint64_t seed,
                                        double minority_fraction, double doublet_fraction);
cellector_status multi_ingest_finish(cellector_ctx *root, uint64_t min_alt, uint64_t min_ref);
cellector_status multi_dims(const cellector_ctx *root, cellector_dims_t *o);
cellector_status multi_entries_per_cell(const cellector_ctx *root, uint32_t *out);
cellector_status multi_excluded(const cellector_ctx *root, uint8_t *out);
cellector_status multi_iter_cell_outputs(const cellector_ctx *root, double *ll, double *ell, double *nl, double *norm);
cellector_status multi_cell_log_likelihoods(cellector_ctx *root, const double *alpha, const double *beta, const uint8_t *mask,
                                            double *ll, double *ell, double *nl);
cellector_status multi_posteriors(cellector_ctx *root, double *posterior, double *doublet, double *ll_maj, double *ll_min);
cellector_status multi_csr_rows(const cellector_ctx *root, uint64_t rb, uint64_t re, uint64_t *row_ptr, uint64_t *entries, uint64_t capacity);
cellector_status multi_em_iteration(cellector_ctx *root, double iqr_multiple, cellector_iter_summary *out);
cellector_status multi_final_allele_tallies(cellector_ctx *root, uint64_t *alt_min, uint64_t *ref_min, uint64_t *alt_maj, uint64_t *ref_maj);
cellector_status multi_engine_info(const cellector_ctx *root, cellector_engine_info_t *o);
cellector_status multi_reset_timing(cellector_ctx *root);
cellector_status multi_order_statistics(cellector_ctx *root, const double *keys, uint64_t n, double iqr_multiple, double *out3);
