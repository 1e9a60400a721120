// Device ingest: staged COO (this shard's entries, file order) -> locus filter + compaction ->
// CSC (by locus, file order) and CSR (by cell, file order) of packed 8-byte entries.
// Replaces get_loci_used (load_data.rs:254-280) and load_cell_data (load_data.rs:134-181).
//
// vartrix / combiner write the matrices sorted by (locus, cell) (combiner/src/main.rs:111-115), so a locus'
// entries are one contiguous run of the file: pass-1 tallies are reduced per run inside a wave before the
// atomic, and the CSC is the filtered file itself.  Unsorted input is first stably sorted by locus.  The
// by-cell transpose is a stable LSD radix sort on the cell index (rocPRIM, called once per load).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <chrono>
#include <cstdlib>

#include "ctx.h"
#include "device_math.h"

#define IB 256

// ---------------------------------------------------------------------------------------------------
// exclusive scan (u64), 2048 elements per block, recursive over block sums
// ---------------------------------------------------------------------------------------------------
#define SCAN_ITEMS 8
#define SCAN_TILE (IB * SCAN_ITEMS)

__global__ __launch_bounds__(IB) void k_scan_block(uint64_t *__restrict__ data, uint64_t n,
                                                   uint64_t *__restrict__ sums)
{
    __shared__ uint64_t wsum[IB / 64];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t tsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? data[base + k] : 0;
        tsum += v[k];
    }
    // inclusive scan of thread sums within the wave
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t inc = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint64_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint64_t woff = 0, total = 0;
#pragma unroll
    for (int k = 0; k < IB / 64; k++) {
        if (k < w) woff += wsum[k];
        total += wsum[k];
    }
    uint64_t run = woff + inc - tsum;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) data[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0 && sums) sums[blockIdx.x] = total;
}

__global__ void k_scan_add(uint64_t *__restrict__ data, uint64_t n, const uint64_t *__restrict__ sums)
{
    const uint64_t i = (uint64_t)blockIdx.x * SCAN_TILE + threadIdx.x;
    const uint64_t add = sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const uint64_t j = i + (uint64_t)k * IB;
        if (j < n) data[j] += add;
    }
}

static cellector_status scan_rec(cellector_ctx *c, uint64_t *data, uint64_t n)
{
    const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint64_t *sums = nullptr;
    if (nb > 1) CHK(dev_alloc(c, &sums, nb));
    hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nb), dim3(IB), 0, c->stream, data, n, sums);
    HIPCHK(c, hipGetLastError());
    if (nb > 1) {
        CHK(scan_rec(c, sums, nb));
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(IB), 0, c->stream, data, n, sums);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dev_free(sums);
    }
    return CELLECTOR_OK;
}

// In-place exclusive scan of data[0..n); data[n-1] should be a trailing 0 so that it ends up holding the
// total, which is also returned to the host.
cellector_status dev_exclusive_scan_u64(cellector_ctx *c, uint64_t *data, uint64_t n, uint64_t *total_out_host)
{
    if (n == 0) {
        if (total_out_host) *total_out_host = 0;
        return CELLECTOR_OK;
    }
    CHK(scan_rec(c, data, n));
    if (total_out_host) {
        HIPCHK(c, hipMemcpyAsync(total_out_host, data + (n - 1), 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return CELLECTOR_OK;
}

cellector_status dev_sort_pairs_u32_u64(cellector_ctx *c, uint32_t *keys_in, uint32_t *keys_out, uint64_t *vals_in,
                                        uint64_t *vals_out, uint64_t n, int end_bit)
{
    if (n == 0) return CELLECTOR_OK;
    size_t tmp_bytes = 0;
    HIPCHK(c, rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                        (unsigned)end_bit, c->stream));
    char *tmp = nullptr;
    CHK(dev_alloc(c, &tmp, tmp_bytes));
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                             (unsigned)end_bit, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(tmp);
    HIPCHK(c, e);
    return CELLECTOR_OK;
}

// The same sort with the caller's two buffer pairs as the sort's ping-pong storage (both get overwritten; on return *keys /
// *vals point at the sorted data, *keys_alt / *vals_alt at the other buffers).  The form above must leave its input alone
// and therefore asks for a second pair of buffers as temporary storage: 24 GB at 2e9 pairs — fresh VRAM whenever the caching
// layer has no block of that size, 0.7-1.5 s of mapping in one run out of three of the 1M x 200k ingest.
cellector_status dev_sort_pairs_u32_u64_inplace(cellector_ctx *c, uint32_t **keys, uint32_t **keys_alt, uint64_t **vals,
                                                uint64_t **vals_alt, uint64_t n, int end_bit)
{
    if (n == 0) return CELLECTOR_OK;
    rocprim::double_buffer<uint32_t> kb(*keys, *keys_alt);
    rocprim::double_buffer<uint64_t> vb(*vals, *vals_alt);
    size_t tmp_bytes = 0;
    HIPCHK(c, rocprim::radix_sort_pairs(nullptr, tmp_bytes, kb, vb, (size_t)n, 0u, (unsigned)end_bit, c->stream));
    char *tmp = nullptr;
    CHK(dev_alloc(c, &tmp, tmp_bytes));
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, kb, vb, (size_t)n, 0u, (unsigned)end_bit, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(tmp);
    HIPCHK(c, e);
    *keys = kb.current(); *keys_alt = kb.alternate();
    *vals = vb.current(); *vals_alt = vb.alternate();
    return CELLECTOR_OK;
}

// ---------------------------------------------------------------------------------------------------
// pass 1 (load_data.rs:265-270) + allele totals: per locus, #cells with ref>0, #cells with alt>0, Σref, Σalt,
// #entries, as f64 (exact: integers < 2^53) into the PASS1 exchange planes.
// ---------------------------------------------------------------------------------------------------
template <bool SORTED>
__global__ __launch_bounds__(IB) void k_pass1(uint64_t n, uint64_t TL, const uint32_t *__restrict__ locus,
                                              const uint16_t *__restrict__ alt, const uint16_t *__restrict__ ref,
                                              double *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * IB + threadIdx.x;
    const bool in = i < n;
    const uint32_t l = in ? locus[i] : 0xffffffffu;
    const uint32_t a = in ? alt[i] : 0, r = in ? ref[i] : 0;
    if (!SORTED) {
        if (!in) return;
        if (r) { atomicAdd(&out[P1_CELLS_REF * TL + l], 1.0); atomicAdd(&out[P1_SUM_REF * TL + l], (double)r); }
        if (a) { atomicAdd(&out[P1_CELLS_ALT * TL + l], 1.0); atomicAdd(&out[P1_SUM_ALT * TL + l], (double)a); }
        atomicAdd(&out[P1_ENTRIES * TL + l], 1.0);
        return;
    }
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(in);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const uint32_t l0 = (uint32_t)__shfl((int)l, src, 64);
        const bool mine = in && l == l0;
        const unsigned long long same = __ballot(mine);
        const uint32_t sa = wave_sum_u32(mine ? a : 0u), sr = wave_sum_u32(mine ? r : 0u);
        const unsigned long long ba = __ballot(mine && a > 0), br = __ballot(mine && r > 0);
        if (lane == 0) {
            if (br) { atomicAdd(&out[P1_CELLS_REF * TL + l0], (double)__popcll(br)); atomicAdd(&out[P1_SUM_REF * TL + l0], (double)sr); }
            if (ba) { atomicAdd(&out[P1_CELLS_ALT * TL + l0], (double)__popcll(ba)); atomicAdd(&out[P1_SUM_ALT * TL + l0], (double)sa); }
            atomicAdd(&out[P1_ENTRIES * TL + l0], (double)__popcll(same));
        }
        todo &= ~same;
    }
}

// locus filter (load_data.rs:273): used flag as u64 for the compaction scan
__global__ void k_used_flag(uint64_t TL, const double *__restrict__ p1, double min_ref, double min_alt,
                            uint64_t *__restrict__ flag)
{
    const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l > TL) return;
    flag[l] = (l < TL && p1[P1_CELLS_REF * TL + l] >= min_ref && p1[P1_CELLS_ALT * TL + l] >= min_alt) ? 1 : 0;
}

// to_used[l] holds the exclusive scan; rewrite as compact index or ~0, and gather per-used-locus data
__global__ void k_compact(uint64_t TL, const double *__restrict__ p1, double min_ref, double min_alt,
                          uint64_t *__restrict__ to_used, uint64_t *__restrict__ locus_ids,
                          double *__restrict__ s_alt, double *__restrict__ s_ref, double *__restrict__ n_ent)
{
    const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= TL) return;
    const bool used = p1[P1_CELLS_REF * TL + l] >= min_ref && p1[P1_CELLS_ALT * TL + l] >= min_alt;
    const uint64_t u = to_used[l];
    if (used) {
        locus_ids[u] = l;
        s_alt[u] = p1[P1_SUM_ALT * TL + l];
        s_ref[u] = p1[P1_SUM_REF * TL + l];
        n_ent[u] = p1[P1_ENTRIES * TL + l];
    } else {
        to_used[l] = ~0ull;
    }
}

// local entry counts per total locus (all entries; gives the file rank inside a locus).  (The per-cell counts used to be taken
// here too, one global atomic per entry: 2e9 scattered atomics = 94 ms at 1M x 200k.  The row pointers now come out of the
// cell-sorted keys after the sort, k_row_ptr_from_keys: no atomics.)
__global__ __launch_bounds__(IB) void k_count(uint64_t n, const uint32_t *__restrict__ locus, unsigned long long *__restrict__ loc_cnt)
{
    const uint64_t i = (uint64_t)blockIdx.x * IB + threadIdx.x;
    const bool in = i < n;
    const uint32_t l = in ? locus[i] : 0xffffffffu;
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(in);
    while (todo) {  // one atomic per distinct locus per wave (1-2 iterations on locus-sorted input)
        const int src = __ffsll((long long)todo) - 1;
        const uint32_t l0 = (uint32_t)__shfl((int)l, src, 64);
        const unsigned long long same = __ballot(in && l == l0);
        if (lane == src) atomicAdd(&loc_cnt[l0], (unsigned long long)__popcll(same));
        todo &= ~same;
    }
}

// row_ptr[r] = number of sorted keys below r, r = 0..n_rows (rows without entries included): thread i looks at the boundary
// between keys i-1 and i and writes the pointers of the rows that start there
__global__ __launch_bounds__(IB) void k_row_ptr_from_keys(uint64_t n, const uint32_t *__restrict__ key_sorted, uint64_t n_rows,
                                                          uint64_t *__restrict__ row_ptr)
{
    const uint64_t i = (uint64_t)blockIdx.x * IB + threadIdx.x;
    if (i > n) return;
    const int64_t prev = i > 0 ? (int64_t)key_sorted[i - 1] : -1;
    const int64_t cur = i < n ? (int64_t)key_sorted[i] : (int64_t)n_rows;
    for (int64_t r = prev + 1; r <= cur; r++) row_ptr[r] = i;
}

__global__ void k_gather_used_counts(uint64_t L, const uint64_t *__restrict__ locus_ids,
                                     const uint64_t *__restrict__ loc_cnt, uint64_t *__restrict__ col_cnt)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u > L) return;
    col_cnt[u] = u < L ? loc_cnt[locus_ids[u]] : 0;
}

// CSC fill for locus-sorted COO: position = col_ptr[compact locus] + (file index - first file index of the locus)
__global__ void k_fill(uint64_t n, const uint32_t *__restrict__ locus, const uint32_t *__restrict__ cell,
                       const uint16_t *__restrict__ alt, const uint16_t *__restrict__ ref,
                       const uint64_t *__restrict__ to_used, const uint64_t *__restrict__ first_i,
                       const uint64_t *__restrict__ col_ptr, uint64_t *__restrict__ csc_ent,
                       uint32_t *__restrict__ key, uint64_t *__restrict__ val)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = locus[i];
    const uint64_t u = to_used[l];
    if (u == ~0ull) return;
    const uint64_t pos = col_ptr[u] + (i - first_i[l]);
    const uint64_t counts = ((uint64_t)alt[i] << 32) | ((uint64_t)ref[i] << 48);
    const uint32_t cl = cell[i];
    csc_ent[pos] = (uint64_t)cl | counts;
    key[pos] = cl;
    val[pos] = u | counts;
}

__global__ void k_iota_u64(uint64_t n, uint64_t *__restrict__ v)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}
__global__ void k_permute_coo(uint64_t n, const uint64_t *__restrict__ perm, const uint32_t *__restrict__ cell,
                              const uint16_t *__restrict__ alt, const uint16_t *__restrict__ ref,
                              uint32_t *__restrict__ cell_o, uint16_t *__restrict__ alt_o, uint16_t *__restrict__ ref_o)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t j = perm[i];
    cell_o[i] = cell[j]; alt_o[i] = alt[j]; ref_o[i] = ref[j];
}

static inline unsigned g1(uint64_t n) { return (unsigned)((n + IB - 1) / IB ? (n + IB - 1) / IB : 1); }

// ---------------------------------------------------------------------------------------------------
cellector_status ingest_stage_host_coo(cellector_ctx *c, uint64_t nnz, const uint32_t *locus0, const uint32_t *cell0,
                                       const uint32_t *alt, const uint32_t *ref)
{
    const uint64_t cb = c->cell_begin, ce = c->cell_end;
    std::vector<uint32_t> hl, hc;
    std::vector<uint16_t> ha, hr;
    hl.reserve(nnz); hc.reserve(nnz); ha.reserve(nnz); hr.reserve(nnz);
    bool sorted = true;
    uint32_t prev = 0;
    for (uint64_t i = 0; i < nnz; i++) {
        if (locus0[i] >= c->total_loci)
            return ctx_fail(c, CELLECTOR_EINVAL, "entry %llu: locus %u out of range (total_loci %llu)",
                            (unsigned long long)i, locus0[i] + 1, (unsigned long long)c->total_loci);
        if (cell0[i] >= c->total_cells)
            return ctx_fail(c, CELLECTOR_EINVAL, "entry %llu: cell %u out of range (total_cells %llu)",
                            (unsigned long long)i, cell0[i] + 1, (unsigned long long)c->total_cells);
        if (alt[i] > CELLECTOR_MAX_COUNT || ref[i] > CELLECTOR_MAX_COUNT)
            return ctx_fail(c, CELLECTOR_EINVAL, "entry %llu: count above %u not supported",
                            (unsigned long long)i, CELLECTOR_MAX_COUNT);
        if (cell0[i] < cb || cell0[i] >= ce) continue;
        if (locus0[i] < prev) sorted = false;
        prev = locus0[i];
        hl.push_back(locus0[i]); hc.push_back((uint32_t)(cell0[i] - cb));
        ha.push_back((uint16_t)alt[i]); hr.push_back((uint16_t)ref[i]);
    }
    const uint64_t n = hl.size();
    c->coo_n = n;
    c->coo_sorted = sorted;
    CHK(dev_alloc(c, &c->coo_locus, n)); CHK(dev_alloc(c, &c->coo_cell, n));
    CHK(dev_alloc(c, &c->coo_alt, n)); CHK(dev_alloc(c, &c->coo_ref, n));
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->coo_locus, hl.data(), n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->coo_cell, hc.data(), n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->coo_alt, ha.data(), n * 2, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->coo_ref, hr.data(), n * 2, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return CELLECTOR_OK;
}

// ---- an all-cells staged COO cut by cell range (multi-device text ingest: the pair is parsed ONCE, its entries routed to
// the shard that owns their cell; file order is kept inside every shard) ---------------------------------------------------
__global__ __launch_bounds__(256) void k_coo_in_range(uint64_t n, const uint32_t *__restrict__ cell, uint64_t cb, uint64_t ce,
                                                      uint64_t *__restrict__ keep /*[n + 1]*/)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    keep[i] = (i < n && cell[i] >= cb && cell[i] < ce) ? 1ull : 0ull;
}
__global__ __launch_bounds__(256) void k_coo_take_range(uint64_t n, const uint32_t *__restrict__ locus, const uint32_t *__restrict__ cell,
                                                        const uint16_t *__restrict__ alt, const uint16_t *__restrict__ ref, uint64_t cb,
                                                        uint64_t ce, const uint64_t *__restrict__ pos, uint32_t *__restrict__ o_locus,
                                                        uint32_t *__restrict__ o_cell, uint16_t *__restrict__ o_alt,
                                                        uint16_t *__restrict__ o_ref)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t c0 = cell[i];
    if (c0 < cb || c0 >= ce) return;
    const uint64_t p = pos[i];
    o_locus[p] = locus[i];
    o_cell[p] = (uint32_t)(c0 - cb);
    o_alt[p] = alt[i];
    o_ref[p] = ref[i];
}
// src holds the staged COO of ALL cells (cell index global).  Writes the entries of cells [cb, ce) — cell index made local,
// order kept — into four new arrays on src's device; `keep` is caller scratch of n + 1 words.
// entries per (global) cell of a staged piece, on the host: what the nnz-balancing partition of a multi-device ingest is cut from
__global__ __launch_bounds__(256) void k_cell_hist(uint64_t n, const uint32_t *__restrict__ cell, uint64_t total_cells, uint32_t *__restrict__ hist)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && cell[i] < total_cells) atomicAdd(&hist[cell[i]], 1u);
}
cellector_status ingest_cell_histogram(cellector_ctx *c, const uint32_t *d_cell, uint64_t n, uint64_t total_cells, std::vector<uint32_t> *out)
{
    uint32_t *d_hist = nullptr;
    out->assign(total_cells, 0u);
    if (!total_cells) return CELLECTOR_OK;
    CHK(dev_alloc(c, &d_hist, total_cells));
    hipError_t e = hipMemsetAsync(d_hist, 0, total_cells * sizeof(uint32_t), c->stream);
    if (e == hipSuccess && n)
        hipLaunchKernelGGL(k_cell_hist, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, d_cell, total_cells, d_hist);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out->data(), d_hist, total_cells * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_hist);
    if (e != hipSuccess) return ctx_fail(c, CELLECTOR_EDEVICE, "cell histogram: %s", hipGetErrorString(e));
    return CELLECTOR_OK;
}

cellector_status ingest_split_coo(cellector_ctx *src, uint64_t cb, uint64_t ce, uint64_t *keep, uint32_t **o_locus, uint32_t **o_cell,
                                  uint16_t **o_alt, uint16_t **o_ref, uint64_t *n_out)
{
    const uint64_t n = src->coo_n;
    const unsigned g = (unsigned)((n + 1 + 255) / 256);
    hipLaunchKernelGGL(k_coo_in_range, dim3(g), dim3(256), 0, src->stream, n, src->coo_cell, cb, ce, keep);
    HIPCHK(src, hipGetLastError());
    uint64_t kept = 0;
    CHK(dev_exclusive_scan_u64(src, keep, n + 1, &kept));
    CHK(dev_alloc(src, o_locus, kept)); CHK(dev_alloc(src, o_cell, kept));
    CHK(dev_alloc(src, o_alt, kept)); CHK(dev_alloc(src, o_ref, kept));
    if (n)
        hipLaunchKernelGGL(k_coo_take_range, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, src->stream, n, src->coo_locus, src->coo_cell,
                           src->coo_alt, src->coo_ref, cb, ce, keep, *o_locus, *o_cell, *o_alt, *o_ref);
    HIPCHK(src, hipGetLastError());
    HIPCHK(src, hipStreamSynchronize(src->stream));
    *n_out = kept;
    return CELLECTOR_OK;
}

cellector_status ingest_pass1(cellector_ctx *c)
{
    const uint64_t TL = c->total_loci;
    HIPCHK(c, hipMemsetAsync(c->x_pass1, 0, P1_PLANES * TL * 8, c->stream));
    if (c->coo_n) {
        if (c->coo_sorted)
            hipLaunchKernelGGL(k_pass1<true>, dim3(g1(c->coo_n)), dim3(IB), 0, c->stream, c->coo_n, TL, c->coo_locus,
                               c->coo_alt, c->coo_ref, c->x_pass1);
        else
            hipLaunchKernelGGL(k_pass1<false>, dim3(g1(c->coo_n)), dim3(IB), 0, c->stream, c->coo_n, TL, c->coo_locus,
                               c->coo_alt, c->coo_ref, c->x_pass1);
        HIPCHK(c, hipGetLastError());
    }
    return CELLECTOR_OK;
}

cellector_status ingest_build(cellector_ctx *c, uint64_t min_alt, uint64_t min_ref)
{
    const uint64_t TL = c->total_loci, n = c->coo_n, nloc = c->nloc;
    const bool timing = getenv("CELLECTOR_TIMING") != nullptr;  // phase wall times on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(c->stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing]     %-22s %8.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
        t_prev = now;
    };
    // ---- filter + compaction map
    CHK(dev_alloc(c, &c->to_used, TL + 1));
    hipLaunchKernelGGL(k_used_flag, dim3(g1(TL + 1)), dim3(IB), 0, c->stream, TL, c->x_pass1, (double)min_ref,
                       (double)min_alt, c->to_used);
    HIPCHK(c, hipGetLastError());
    uint64_t L = 0;
    CHK(dev_exclusive_scan_u64(c, c->to_used, TL + 1, &L));
    c->L = L;
    CHK(dev_alloc(c, &c->locus_ids, L)); CHK(dev_alloc(c, &c->s_alt, L));
    CHK(dev_alloc(c, &c->s_ref, L)); CHK(dev_alloc(c, &c->n_ent, L));
    if (TL)
        hipLaunchKernelGGL(k_compact, dim3(g1(TL)), dim3(IB), 0, c->stream, TL, c->x_pass1, (double)min_ref,
                           (double)min_alt, c->to_used, c->locus_ids, c->s_alt, c->s_ref, c->n_ent);
    HIPCHK(c, hipGetLastError());

    // ---- unsorted input: stable sort of the staged COO by locus
    if (!c->coo_sorted && n) {
        uint32_t *k_out = nullptr, *cell_o = nullptr;
        uint16_t *alt_o = nullptr, *ref_o = nullptr;
        uint64_t *perm = nullptr, *perm_o = nullptr;
        CHK(dev_alloc(c, &k_out, n)); CHK(dev_alloc(c, &perm, n)); CHK(dev_alloc(c, &perm_o, n));
        hipLaunchKernelGGL(k_iota_u64, dim3(g1(n)), dim3(IB), 0, c->stream, n, perm);
        int bits = 1;
        while (bits < 32 && (1ull << bits) < TL) bits++;
        CHK(dev_sort_pairs_u32_u64(c, c->coo_locus, k_out, perm, perm_o, n, bits));
        CHK(dev_alloc(c, &cell_o, n)); CHK(dev_alloc(c, &alt_o, n)); CHK(dev_alloc(c, &ref_o, n));
        hipLaunchKernelGGL(k_permute_coo, dim3(g1(n)), dim3(IB), 0, c->stream, n, perm_o, c->coo_cell, c->coo_alt,
                           c->coo_ref, cell_o, alt_o, ref_o);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dev_free(c->coo_locus); dev_free(c->coo_cell); dev_free(c->coo_alt); dev_free(c->coo_ref);
        dev_free(perm); dev_free(perm_o);
        c->coo_locus = k_out; c->coo_cell = cell_o; c->coo_alt = alt_o; c->coo_ref = ref_o;
        c->coo_sorted = true;
    }

    lap("filter + compaction");
    // ---- counts and pointers
    uint64_t *loc_cnt = nullptr, *col_cnt = nullptr;
    CHK(dev_alloc(c, &c->csr_ptr, nloc + 1));
    CHK(dev_alloc(c, &loc_cnt, TL + 1));
    CHK(dev_alloc(c, &c->csc_ptr, L + 1));
    HIPCHK(c, hipMemsetAsync(loc_cnt, 0, (TL + 1) * 8, c->stream));
    if (n)
        hipLaunchKernelGGL(k_count, dim3(g1(n)), dim3(IB), 0, c->stream, n, c->coo_locus, (unsigned long long *)loc_cnt);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(k_gather_used_counts, dim3(g1(L + 1)), dim3(IB), 0, c->stream, L, c->locus_ids, loc_cnt,
                       c->csc_ptr);
    HIPCHK(c, hipGetLastError());
    uint64_t nnz_cols = 0;
    CHK(dev_exclusive_scan_u64(c, c->csc_ptr, L + 1, &nnz_cols));
    CHK(dev_exclusive_scan_u64(c, loc_cnt, TL + 1, nullptr));  // now: first file index of each locus
    c->nnz = nnz_cols;

    lap("counts + scans");
    // ---- CSC = filtered file order; CSR = stable sort of the same entries by cell
    uint32_t *key = nullptr, *key_o = nullptr;
    uint64_t *val = nullptr;
    CHK(dev_alloc(c, &c->csc_ent, c->nnz)); CHK(dev_alloc(c, &c->csr_ent, c->nnz));
    CHK(dev_alloc(c, &key, c->nnz)); CHK(dev_alloc(c, &key_o, c->nnz)); CHK(dev_alloc(c, &val, c->nnz));
    lap("allocations");
    if (n)
        hipLaunchKernelGGL(k_fill, dim3(g1(n)), dim3(IB), 0, c->stream, n, c->coo_locus, c->coo_cell, c->coo_alt,
                           c->coo_ref, c->to_used, loc_cnt, c->csc_ptr, c->csc_ent, key, val);
    HIPCHK(c, hipGetLastError());
    lap("CSC fill");
    int bits = 1;
    while (bits < 32 && (1ull << bits) < nloc) bits++;
    // (the sorted entries end in one of the two value buffers: that one becomes the CSR, the other is freed below)
    CHK(dev_sort_pairs_u32_u64_inplace(c, &key, &key_o, &val, &c->csr_ent, c->nnz, bits));  // (key / val: sorted now)
    std::swap(val, c->csr_ent);
    hipLaunchKernelGGL(k_row_ptr_from_keys, dim3(g1(c->nnz + 1)), dim3(IB), 0, c->stream, c->nnz, key, nloc, c->csr_ptr);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    lap("sort by cell");
    dev_free(key); dev_free(key_o); dev_free(val); dev_free(loc_cnt);
    lap("frees");
    (void)col_cnt;
    if (!c->keep_coo) {
        dev_free(c->coo_locus); dev_free(c->coo_cell); dev_free(c->coo_alt); dev_free(c->coo_ref);
        c->coo_n = 0;
    }
    return CELLECTOR_OK;
}
