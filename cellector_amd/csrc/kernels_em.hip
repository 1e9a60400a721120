// EM-loop kernels for gfx950 (wave64): per-cell log-likelihood pass over the CSR, per-locus pass over
// the CSC, exclusion flags, locus filter, alpha/beta update, fused posterior pass.
//
// All of these are HBM-bound streaming passes over 8-byte packed entries with one 16-byte gather of the
// per-locus (alpha, beta) pair per entry (cell-major pass) or of the per-cell flag (locus-major pass).
#include "ctx.h"
#include "device_math.h"

#define WAVES_PER_BLOCK 4
#define BLOCK (WAVES_PER_BLOCK * 64)

// ---------------------------------------------------------------------------------------------------
// init_alpha_betas (main.rs:598-611): alpha_l = (S_alt_l + 1) - Σ_{c∈X} alt_{c,l}; the subtrahend is the
// all-reduced ALT_MIN plane of the previous iteration.  Masked loci get alpha = -1 so that the cell pass
// needs no separate mask gather (`if loci_used[locus_index]`, main.rs:556).
// ---------------------------------------------------------------------------------------------------
__global__ void k_alpha_beta(uint64_t L, const double *__restrict__ s_alt, const double *__restrict__ s_ref,
                             const double *__restrict__ alt_min, const double *__restrict__ ref_min,
                             const uint8_t *__restrict__ mask, double2 *__restrict__ ab,
                             double *__restrict__ xl_counters, uint32_t *__restrict__ d_counters,
                             uint32_t *__restrict__ tile_work, uint32_t n_work)
{
    // first kernel of an iteration: it also resets the iteration's small counters (instead of one memset each)
    if (blockIdx.x == 0 && xl_counters) {
        if (threadIdx.x < LC_COUNTERS) xl_counters[threadIdx.x] = 0.0;
        if (threadIdx.x < 8) d_counters[threadIdx.x] = 0u;
        for (uint32_t i = threadIdx.x; i < n_work; i += blockDim.x) tile_work[i] = 0u;
    }
    uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    double2 v;
    v.x = (s_alt[l] + 1.0) - alt_min[l];
    v.y = (s_ref[l] + 1.0) - ref_min[l];
    if (mask && !mask[l]) v.x = v.y = -1.0;
    ab[l] = v;
}

__global__ void k_ab_from_arrays(uint64_t L, const double *__restrict__ alpha, const double *__restrict__ beta,
                                 const uint8_t *__restrict__ mask, double2 *__restrict__ ab)
{
    uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    double2 v = make_double2(alpha[l], beta[l]);
    if (mask && !mask[l]) v.x = v.y = -1.0;
    ab[l] = v;
}

// ---------------------------------------------------------------------------------------------------
// get_cell_log_likelihoods (main.rs:541-591) + normalisation (main.rs:314-323).
// One wave per cell row; lanes stride the row (coalesced 512 B per wave-load), gather (alpha,beta) of the
// entry's locus, accumulate log-pmf / expected-log-pmf / used-locus count, wave shuffle reduction.
// ---------------------------------------------------------------------------------------------------
template <bool EXPECTED, bool REF /*option ref_arith: the reference's ln_gamma arithmetic*/>
__global__ __launch_bounds__(BLOCK) void k_cell_ll(uint64_t n_rows, const uint64_t *__restrict__ row_ptr,
                                                   const uint64_t *__restrict__ ent,
                                                   const double2 *__restrict__ ab,
                                                   const double *__restrict__ lf_g, double *__restrict__ ll,
                                                   double *__restrict__ ell, double *__restrict__ nloci,
                                                   double *__restrict__ norm_out)
{
    __shared__ double lf[LF_TABLE_N];
    for (int i = threadIdx.x; i < LF_TABLE_N; i += BLOCK) lf[i] = lf_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    for (uint64_t row = wave0; row < n_rows; row += nwaves) {
        const uint64_t beg = row_ptr[row], end = row_ptr[row + 1];
        double s = 0.0, e = 0.0;
        uint32_t cnt = 0;
        for (uint64_t i = beg + lane; i < end; i += 64) {
            const uint64_t en = ent[i];
            const double2 p = ab[ENT_IDX(en)];
            if (p.x >= 0.0) {
                const uint32_t a = ENT_ALT(en), r = ENT_REF(en);
                s += REF ? dm_log_bb_pmf_ref(lf, p.x, p.y, a, r) : dm_log_bb_pmf(lf, p.x, p.y, a, r);
                if (EXPECTED) e += REF ? dm_expected_log_pmf_ref(lf, p.x, p.y, a + r) : dm_expected_log_pmf(lf, p.x, p.y, a + r);
                cnt++;
            }
        }
        s = wave_sum(s);
        if (EXPECTED) e = wave_sum(e);
        cnt = wave_sum_u32(cnt);
        if (lane == 0) {
            ll[row] = s;
            if (EXPECTED) ell[row] = e;
            nloci[row] = (double)cnt;
            if (norm_out) norm_out[row] = cnt > 0 ? s / (double)cnt : 0.0;  // main.rs:315-322
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// new_excluded = {i : norm_i < threshold} (main.rs:330-332) and the symmetric-difference counts
// (main.rs:333-334) accumulated as f64 into the LOCUS exchange buffer's counter slots.
// ---------------------------------------------------------------------------------------------------
#define FLAG_THREADS 1024
__global__ __launch_bounds__(FLAG_THREADS) void k_flag(uint64_t n, const double *__restrict__ norm, const double *__restrict__ d_thr,
                                                       const uint8_t *__restrict__ old_flags, uint8_t *__restrict__ new_flags,
                                                       double *__restrict__ counters, uint32_t *__restrict__ minlist /*may be null*/,
                                                       uint32_t *__restrict__ n_min, uint32_t *__restrict__ flag_bits /*with minlist*/,
                                                       double near_rel)
{
    __shared__ uint32_t s_wave[FLAG_THREADS / 64];
    __shared__ uint32_t s_base;
    const double thr = *d_thr;  // computed on the device by k_threshold
    // a block takes a contiguous span of cells (a multiple of the block size), wave w of it the 64-cell pieces w, w+16, ...
    const uint64_t span = ((n + gridDim.x - 1) / gridDim.x + FLAG_THREADS - 1) / FLAG_THREADS * FLAG_THREADS;
    const uint64_t beg = min(n, (uint64_t)blockIdx.x * span), end = min(n, beg + span);
    uint32_t c_new = 0, c_res = 0, c_exc = 0, c_near = 0;
    // near-ties: the device arithmetic is a reformulation of the reference's (device_math.h), ~1e-11 away from it on a
    // normalised log-likelihood at vartrix-like depth (more on deep loci: the band follows the matrix, cellector_ingest_finish);
    // a cell this close to the threshold could fall on the other side of main.rs:330-332's strict `<` in the reference.
    // Counted, reported in the summary, never acted upon.
    const double near_tol = near_rel * fmax(1.0, fabs(thr));
    for (uint64_t i = beg + threadIdx.x; i < end; i += FLAG_THREADS) {
        const double v = norm[i];
        const bool nf = v < thr;
        const bool of = old_flags[i] != 0;
        new_flags[i] = nf ? 1 : 0;
        c_new += (nf && !of) ? 1u : 0u;
        c_res += (of && !nf) ? 1u : 0u;
        c_exc += nf ? 1u : 0u;
        c_near += fabs(v - thr) <= near_tol ? 1u : 0u;
    }
    // one f64 atomic per BLOCK and counter (integers: exact and order independent; atomics on one address serialise)
    __shared__ uint32_t s_new[FLAG_THREADS / 64], s_res[FLAG_THREADS / 64], s_near[FLAG_THREADS / 64];
    c_new = wave_sum_u32(c_new);
    c_res = wave_sum_u32(c_res);
    c_exc = wave_sum_u32(c_exc);
    c_near = wave_sum_u32(c_near);
    if ((threadIdx.x & 63) == 0) {
        s_new[threadIdx.x >> 6] = c_new;
        s_res[threadIdx.x >> 6] = c_res;
        s_wave[threadIdx.x >> 6] = c_exc;
        s_near[threadIdx.x >> 6] = c_near;
    }
    __syncthreads();
    // The members of the new exclusion set as a list (engine 2's minority-driven locus tally): ONE atomic per block
    // reserves the block's slots, then a second walk over the span (L2-hot) fills them.  The list's order depends on
    // the block order only through the bases, and only order-independent integer tallies are derived from it.
    if (threadIdx.x == 0) {
        uint32_t tot = 0, t_new = 0, t_res = 0, t_near = 0;
        for (int w = 0; w < FLAG_THREADS / 64; w++) {
            const uint32_t x = s_wave[w];
            s_wave[w] = tot;
            tot += x;
            t_new += s_new[w];
            t_res += s_res[w];
            t_near += s_near[w];
        }
        if (t_near) atomicAdd(&counters[LC_N_NEAR], (double)t_near);
        if (t_new) atomicAdd(&counters[LC_N_NEW], (double)t_new);
        if (t_res) atomicAdd(&counters[LC_N_RESCUED], (double)t_res);
        if (tot) atomicAdd(&counters[LC_N_EXCLUDED], (double)tot);
        s_base = (tot && minlist) ? atomicAdd(n_min, tot) : 0u;
    }
    if (!minlist) return;
    __syncthreads();
    uint32_t pos = s_base + s_wave[threadIdx.x >> 6];
    const uint64_t end_round = beg + (end - beg + 63) / 64 * 64;  // whole waves take part in the ballot
    for (uint64_t i = beg + threadIdx.x; i < end_round; i += FLAG_THREADS) {
        const bool nf = i < end && norm[i] < thr;
        const unsigned long long m = __ballot(nf);
        if (nf) minlist[pos + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = (uint32_t)i;
        pos += (uint32_t)__popcll(m);
        // the exclusion set as a bitmask too (locus pass): the wave's 64 cells are two words (i is 64-aligned in lane 0)
        if ((threadIdx.x & 31) == 0 && i < end) flag_bits[i >> 5] = (uint32_t)(m >> (threadIdx.x & 32));
    }
}

// ---------------------------------------------------------------------------------------------------
// get_locus_log_likelihoods (main.rs:368-420) for this shard: one wave per locus column of the CSC.
// alpha/beta are wave-uniform; the per-cell flag of the NEW exclusion set is a byte gather.  The log-pmfs
// are those of this iteration's alpha/beta (quirk Q9).  ALT_MIN/REF_MIN are tallied over every entry,
// masked loci included (they feed init_alpha_betas and the posterior alpha/betas, which ignore the mask);
// contributions and cell counts are zero for masked loci (a masked locus has no PMFData, main.rs:556).
// ---------------------------------------------------------------------------------------------------
template <bool REF>
__global__ __launch_bounds__(BLOCK) void k_locus_stats(uint64_t L, const uint64_t *__restrict__ col_ptr,
                                                       const uint64_t *__restrict__ ent,
                                                       const double2 *__restrict__ ab,
                                                       const uint8_t *__restrict__ flags,
                                                       const double *__restrict__ lf_g,
                                                       double *__restrict__ out)
{
    __shared__ double lf[LF_TABLE_N];
    for (int i = threadIdx.x; i < LF_TABLE_N; i += BLOCK) lf[i] = lf_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    for (uint64_t l = wave0; l < L; l += nwaves) {
        const uint64_t beg = col_ptr[l], end = col_ptr[l + 1];
        const double2 p = ab[l];
        const bool live = p.x >= 0.0;
        double cmin = 0.0, cmaj = 0.0;
        uint32_t nmin = 0;
        uint64_t amin = 0, rmin = 0;
        for (uint64_t i = beg + lane; i < end; i += 64) {
            const uint64_t en = ent[i];
            const uint32_t a = ENT_ALT(en), r = ENT_REF(en);
            const bool minority = flags[ENT_IDX(en)] != 0;
            if (minority) { amin += a; rmin += r; }
            if (live) {
                const double lp = REF ? dm_log_bb_pmf_ref(lf, p.x, p.y, a, r) : dm_log_bb_pmf(lf, p.x, p.y, a, r);
                if (minority) { cmin += lp; nmin++; } else cmaj += lp;
            }
        }
        cmin = wave_sum(cmin);
        cmaj = wave_sum(cmaj);
        nmin = wave_sum_u32(nmin);
        amin = wave_sum_u64(amin);
        rmin = wave_sum_u64(rmin);
        if (lane == 0) {
            out[LB_CONTRIB_MIN * L + l] = cmin;
            out[LB_CONTRIB_MAJ * L + l] = cmaj;
            out[LB_CELLS_MIN * L + l] = (double)nmin;
            out[LB_ALT_MIN * L + l] = (double)amin;
            out[LB_REF_MIN * L + l] = (double)rmin;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// locus filter of locus_filter_and_output_locus_data (main.rs:428-451): per-minority-cell contribution
// below the hard-coded -80 masks the locus from the next iteration on.
// ---------------------------------------------------------------------------------------------------
__global__ void k_locus_filter(uint64_t L, const double *__restrict__ buf, const uint8_t *__restrict__ mask,
                               uint8_t *__restrict__ mask_next, uint32_t *__restrict__ n_filtered)
{
    uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const double nmin = buf[LB_CELLS_MIN * L + l];
    const double per_cell = nmin != 0.0 ? buf[LB_CONTRIB_MIN * L + l] / nmin : 0.0;
    uint8_t m = mask[l];
    if (per_cell < -80.0) {
        m = 0;
        atomicAdd(n_filtered, 1u);
    }
    mask_next[l] = m;
}

// ---------------------------------------------------------------------------------------------------
// calculate_posteriors (main.rs:228-280).  k_ab_posterior builds the three alpha/beta sets in the
// reference's operation order; k_posterior is the cell pass with all three distributions fused (one read
// of the CSR instead of three) followed by the logsumexp chain of main.rs:266-278.
// ---------------------------------------------------------------------------------------------------
__global__ void k_ab_posterior(uint64_t L, const double *__restrict__ s_alt, const double *__restrict__ s_ref,
                               const double *__restrict__ alt_min, const double *__restrict__ ref_min,
                               double mf0, double *__restrict__ ab6)
{
    uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    // majority: excluded cells removed (main.rs:239); minority: included cells removed (main.rs:241)
    double a_maj = (s_alt[l] + 1.0) - alt_min[l], b_maj = (s_ref[l] + 1.0) - ref_min[l];
    double a_min = (s_alt[l] + 1.0) - (s_alt[l] - alt_min[l]), b_min = (s_ref[l] + 1.0) - (s_ref[l] - ref_min[l]);
    double a_dbl = (a_maj - 1.0) * mf0 + (a_min - 1.0) + 1.0;  // main.rs:245-246, unclamped fraction
    double b_dbl = (b_maj - 1.0) * mf0 + (b_min - 1.0) + 1.0;
    const double mf = fmax(mf0, 0.01);  // main.rs:250
    a_maj = (a_maj - 1.0) * mf + 1.0;   // main.rs:252-253
    b_maj = (b_maj - 1.0) * mf + 1.0;
    double *o = ab6 + 8 * l;
    o[0] = a_min; o[1] = b_min; o[2] = a_maj; o[3] = b_maj; o[4] = a_dbl; o[5] = b_dbl; o[6] = 0.0; o[7] = 0.0;
}

template <bool REF>
__global__ __launch_bounds__(BLOCK) void k_posterior(uint64_t n_rows, const uint64_t *__restrict__ row_ptr,
                                                     const uint64_t *__restrict__ ent,
                                                     const double *__restrict__ ab6,
                                                     const double *__restrict__ lf_g, double lp_min,
                                                     double lp_maj, double lp_dbl, double *__restrict__ post)
{
    __shared__ double lf[LF_TABLE_N];
    for (int i = threadIdx.x; i < LF_TABLE_N; i += BLOCK) lf[i] = lf_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    for (uint64_t row = wave0; row < n_rows; row += nwaves) {
        const uint64_t beg = row_ptr[row], end = row_ptr[row + 1];
        double s_min = 0.0, s_maj = 0.0, s_dbl = 0.0;
        for (uint64_t i = beg + lane; i < end; i += 64) {
            const uint64_t en = ent[i];
            const double2 *p = reinterpret_cast<const double2 *>(ab6 + 8 * (uint64_t)ENT_IDX(en));
            const double2 pmin = p[0], pmaj = p[1], pdbl = p[2];
            const uint32_t a = ENT_ALT(en), r = ENT_REF(en);
            if (REF) {
                s_min += dm_log_bb_pmf_ref(lf, pmin.x, pmin.y, a, r);
                s_maj += dm_log_bb_pmf_ref(lf, pmaj.x, pmaj.y, a, r);
                s_dbl += dm_log_bb_pmf_ref(lf, pdbl.x, pdbl.y, a, r);
            } else {
                const double lnc = dm_ln_choose(lf, a, r);
                s_min += lnc + dm_log_beta_ratio(pmin.x, pmin.y, a, r);
                s_maj += lnc + dm_log_beta_ratio(pmaj.x, pmaj.y, a, r);
                s_dbl += lnc + dm_log_beta_ratio(pdbl.x, pdbl.y, a, r);
            }
        }
        s_min = wave_sum(s_min);
        s_maj = wave_sum(s_maj);
        s_dbl = wave_sum(s_dbl);
        if (lane == 0) {
            const double log_num = lp_min + s_min;                               // main.rs:267
            double log_den = dm_logsumexp(log_num, lp_maj + s_maj);              // main.rs:268
            const double log_dbl = lp_dbl + s_dbl;                               // main.rs:270
            log_den = dm_logsumexp(log_den, log_dbl);                            // main.rs:271
            post[row] = exp(log_num - log_den);                                  // main.rs:272-273
            post[n_rows + row] = exp(log_dbl - log_den);                         // main.rs:275
            post[2 * n_rows + row] = s_maj;
            post[3 * n_rows + row] = s_min;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// load_mtx_final (load_data.rs:109-132): allele tallies over ALL loci (also those failing
// min_ref/min_alt) by the final exclusion set, from the staged COO.  out = [alt_min|ref_min|alt_maj|ref_maj].
// ---------------------------------------------------------------------------------------------------
__global__ void k_final_tallies(uint64_t n, uint64_t total_loci, const uint32_t *__restrict__ locus,
                                const uint32_t *__restrict__ cell, const uint16_t *__restrict__ alt,
                                const uint16_t *__restrict__ ref, const uint8_t *__restrict__ flags,
                                unsigned long long *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < n;
    const uint64_t l = ok ? locus[i] : ~0ull;
    const bool minority = ok && flags[cell[i]] != 0;
    const uint32_t a = ok ? alt[i] : 0u, r = ok ? ref[i] : 0u;
    // The staged entries are in file order, i.e. locus-major (vartrix): a wave's 64 entries mostly belong to ONE locus, and
    // 64 atomics to two or four addresses run one after the other (0.4 s at 2e9 entries).  Such a wave adds its four sums
    // with one atomic each.
    const unsigned long long act = __ballot(ok);
    if (!act) return;
    const int src = __ffsll((long long)act) - 1;
    const uint64_t l0 = (uint64_t)__shfl((long long)l, src, 64);
    if (__ballot(ok && l != l0) == 0ull) {
        const uint32_t a_min = wave_sum_u32(minority ? a : 0u), r_min = wave_sum_u32(minority ? r : 0u);
        const uint32_t a_maj = wave_sum_u32(!minority ? a : 0u), r_maj = wave_sum_u32(!minority ? r : 0u);
        if ((threadIdx.x & 63) == 0) {  // (lane 0 is active whenever any lane of the wave is: i grows with the lane)
            if (a_min) atomicAdd(&out[l0], (unsigned long long)a_min);
            if (r_min) atomicAdd(&out[total_loci + l0], (unsigned long long)r_min);
            if (a_maj) atomicAdd(&out[2 * total_loci + l0], (unsigned long long)a_maj);
            if (r_maj) atomicAdd(&out[3 * total_loci + l0], (unsigned long long)r_maj);
        }
        return;
    }
    if (!ok) return;
    const uint64_t base = minority ? 0 : 2 * total_loci;
    if (a) atomicAdd(&out[base + l], (unsigned long long)a);
    if (r) atomicAdd(&out[base + total_loci + l], (unsigned long long)r);
}

// the iteration's summary, written straight into pinned host memory (one kernel instead of three small copies); the
// sequence number goes last, released at system scope: the host polls it (cellector_em_finish)
__global__ void k_iter_summary(const double *__restrict__ xl_counters, const uint32_t *__restrict__ d_counters,
                               const double *__restrict__ sel, double *__restrict__ h_sum, double seq)
{
    const int t = threadIdx.x;
    if (t < LC_COUNTERS) h_sum[t] = xl_counters[t];
    else if (t == LC_COUNTERS) h_sum[t] = (double)d_counters[DC_N_FILTERED];
    else if (t < LC_COUNTERS + 4) h_sum[t] = sel[t - LC_COUNTERS - 1];  // median, iqr, threshold
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(&h_sum[CELLECTOR_SUM_SEQ], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ===================================================================================================
// launch wrappers
// ===================================================================================================
static inline unsigned grid_for(uint64_t n, unsigned per_block, unsigned cap = 0x7fffffffu)
{
    uint64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

cellector_status launch_alpha_beta(cellector_ctx *c)
{
    const uint64_t L = c->L;
    const uint32_t n_work = c->tile_work ? 3u * CELLECTOR_TILE_WORK_STRIDE : 0u;
    hipLaunchKernelGGL(k_alpha_beta, dim3(grid_for(L, 256)), dim3(256), 0, c->stream, L, c->s_alt, c->s_ref,
                       c->x_locus + LB_ALT_MIN * L, c->x_locus + LB_REF_MIN * L, c->mask, c->ab,
                       c->x_locus + (uint64_t)LB_PLANES * L, c->d_counters, c->tile_work, n_work);
    c->work_zeroed = n_work != 0;
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

cellector_status launch_ab_from_host(cellector_ctx *c, const double *alpha, const double *beta,
                                     const uint8_t *mask)
{
    const uint64_t L = c->L;
    double *d_a = nullptr, *d_b = nullptr;
    uint8_t *d_m = nullptr;
    CHK(dev_alloc(c, &d_a, L));
    CHK(dev_alloc(c, &d_b, L));
    HIPCHK(c, hipMemcpyAsync(d_a, alpha, L * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_b, beta, L * 8, hipMemcpyHostToDevice, c->stream));
    if (mask) {
        CHK(dev_alloc(c, &d_m, L));
        HIPCHK(c, hipMemcpyAsync(d_m, mask, L, hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_ab_from_arrays, dim3(grid_for(L, 256)), dim3(256), 0, c->stream, L, d_a, d_b, d_m, c->ab);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dev_free(d_a); dev_free(d_b); dev_free(d_m);
    return CELLECTOR_OK;
}

cellector_status launch_cell_ll(cellector_ctx *c, const double2 *ab, double *norm_out)
{
    if (c->nloc == 0) return CELLECTOR_OK;
    // one wave per row, grid-stride beyond 2^20 blocks
    const unsigned grid = grid_for(c->nloc, WAVES_PER_BLOCK, 1u << 20);
    timer_begin(c, CELLECTOR_K_CELL_LL);
#define LAUNCH_CELL(E, R)                                                                                           \
    hipLaunchKernelGGL((k_cell_ll<E, R>), dim3(grid), dim3(BLOCK), 0, c->stream, c->nloc, c->csr_ptr, c->csr_ent, ab, \
                       c->lf, c->ll, c->ell, c->nloci, norm_out)
    if (c->compute_expected) {
        if (c->ref_arith) LAUNCH_CELL(true, true); else LAUNCH_CELL(true, false);
    } else {
        if (c->ref_arith) LAUNCH_CELL(false, true); else LAUNCH_CELL(false, false);
    }
#undef LAUNCH_CELL
    timer_end(c, CELLECTOR_K_CELL_LL);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

cellector_status launch_flag(cellector_ctx *c, const double *d_thr)
{
    if (c->nloc == 0) return CELLECTOR_OK;
    hipLaunchKernelGGL(k_flag, dim3(grid_for(c->nloc, FLAG_THREADS * 4, 256)), dim3(FLAG_THREADS), 0, c->stream, c->nloc,
                       c->x_norm + c->cell_begin, d_thr, c->flags, c->flags_new, c->x_locus + LB_PLANES * c->L,
                       c->tiled_ready ? c->minlist : (uint32_t *)nullptr, c->d_counters + DC_N_MIN, c->flag_bits, c->near_rel);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

cellector_status launch_locus_stats(cellector_ctx *c)
{
    if (c->L == 0) return CELLECTOR_OK;
    const unsigned grid = grid_for(c->L, WAVES_PER_BLOCK, 1u << 20);
    timer_begin(c, CELLECTOR_K_LOCUS_STATS);
    if (c->ref_arith)
        hipLaunchKernelGGL(k_locus_stats<true>, dim3(grid), dim3(BLOCK), 0, c->stream, c->L, c->csc_ptr, c->csc_ent, c->ab,
                           c->flags_new, c->lf, c->x_locus);
    else
        hipLaunchKernelGGL(k_locus_stats<false>, dim3(grid), dim3(BLOCK), 0, c->stream, c->L, c->csc_ptr, c->csc_ent, c->ab,
                           c->flags_new, c->lf, c->x_locus);
    timer_end(c, CELLECTOR_K_LOCUS_STATS);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

cellector_status launch_locus_filter(cellector_ctx *c)
{
    if (c->L == 0) return CELLECTOR_OK;
    hipLaunchKernelGGL(k_locus_filter, dim3(grid_for(c->L, 256)), dim3(256), 0, c->stream, c->L, c->x_locus, c->mask,
                       c->mask_next, c->d_counters);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

cellector_status launch_iter_summary(cellector_ctx *c)
{
    hipLaunchKernelGGL(k_iter_summary, dim3(1), dim3(64), 0, c->stream, c->x_locus + (uint64_t)LB_PLANES * c->L, c->d_counters,
                       c->sel_out + 8, c->h_sum_dev, (double)++c->sum_seq);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

cellector_status launch_posteriors(cellector_ctx *c, double mf0, double lp_min, double lp_maj, double lp_dbl)
{
    const uint64_t L = c->L;
    if (L)
        hipLaunchKernelGGL(k_ab_posterior, dim3(grid_for(L, 256)), dim3(256), 0, c->stream, L, c->s_alt, c->s_ref,
                           c->x_locus + LB_ALT_MIN * L, c->x_locus + LB_REF_MIN * L, mf0, c->ab6);
    HIPCHK(c, hipGetLastError());
    if (c->nloc == 0) return CELLECTOR_OK;
    const unsigned grid = grid_for(c->nloc, WAVES_PER_BLOCK, 1u << 20);
    timer_begin(c, CELLECTOR_K_POSTERIOR);
    if (c->ref_arith)
        hipLaunchKernelGGL(k_posterior<true>, dim3(grid), dim3(BLOCK), 0, c->stream, c->nloc, c->csr_ptr, c->csr_ent, c->ab6,
                           c->lf, lp_min, lp_maj, lp_dbl, c->post);
    else
        hipLaunchKernelGGL(k_posterior<false>, dim3(grid), dim3(BLOCK), 0, c->stream, c->nloc, c->csr_ptr, c->csr_ent, c->ab6,
                           c->lf, lp_min, lp_maj, lp_dbl, c->post);
    timer_end(c, CELLECTOR_K_POSTERIOR);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

cellector_status launch_final_tallies(cellector_ctx *c, uint64_t *d_out)
{
    HIPCHK(c, hipMemsetAsync(d_out, 0, 4 * c->total_loci * 8, c->stream));
    if (c->coo_n)
        hipLaunchKernelGGL(k_final_tallies, dim3(grid_for(c->coo_n, 256)), dim3(256), 0, c->stream, c->coo_n,
                           c->total_loci, c->coo_locus, c->coo_cell, c->coo_alt, c->coo_ref, c->flags,
                           (unsigned long long *)d_out);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}
