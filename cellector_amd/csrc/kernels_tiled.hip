// Engine v2: table-driven passes over a tiled 16-bit matrix layout (gfx950, wave64).
//
// Observation: under one (alpha_l, beta_l) the log-pmf of an entry depends only on (alt, ref), and vartrix counts
// are tiny (n = alt+ref <= 4 for ~99 % of entries).  So per EM iteration and locus we tabulate the 14 log-pmfs of
// n = 1..4 and the 4 expected terms (18 doubles per locus) and the matrix passes become pure table lookups:
//
//   cell pass   (get_cell_log_likelihoods, main.rs:541-591): the matrix is cut into (1024-cell block x 639-locus
//               chunk) tiles, stored in a sliced-ELLPACK form with a sorting window of one tile (SELL-64-1024): the
//               cells of a tile are ordered by their entry count, every 64 of them form a slice whose rows (one per
//               cell: cell id + entries) are padded to the slice's longest cell, so that all lanes of a wave run the
//               SAME number of lookups — no divergence and ~10 % padding instead of the ~55 % idle lanes of a
//               cell-per-lane walk in file order — and a lane fetches its row with one or two 16-byte loads.
//               A u16 entry is n-1 << 14 | locus slot << 4 | code; padding points at an all-zero slot.
//               A persistent 1024-thread workgroup takes columns of four cell blocks and a group of chunks: the chunk's
//               table lives in LDS locus-major (144 B per locus: the log-pmf and the expected term of an entry are two
//               8-byte reads near each other); a lane keeps its cell's entries of the tile in registers (prefetched two
//               tiles ahead) and adds the tile's sum to the cell's accumulator in LDS (the lane <-> cell assignment
//               changes from tile to tile).  No transcendental, no atomics; a cell's sum runs over its chunks in order
//               and inside a chunk in ascending-locus order: bit-deterministic (the number of chunk groups fixes how
//               the partials associate).
//   locus pass  (get_locus_log_likelihoods, main.rs:368-420): all outputs follow from the minority cells' entry counts per
//               (locus, code): contributions are count x table value; the majority side is (static histogram - minority).
//               The counts come either from walking only the excluded cells' CSR rows into LDS range histograms
//               (k_minority_ranges: the usual case, a few percent of the matrix) or from streaming a compact CSC of
//               24/32-bit entries (cell | code) past the exclusion bitmask in LDS (k_locus_stats2); chosen on the device.
//   overflow    entries with n == 0 or n > 4 (~1 %) live in a small CSR/CSC in the v1 packed format; the cell side
//               evaluates them itself, the locus side from per-locus cumulative-log tables (see below).
#include <algorithm>
#include <type_traits>

#include <hip/hip_ext.h>

#include "ctx.h"
#include "device_math.h"

#define T_K 4           // entries with 1 <= alt+ref <= T_K are "regular": log-pmf and expected term come from tables
#define T_NCODE 14      // (alt, ref) combinations with 1 <= n <= T_K: K(K+3)/2
#ifndef T_LROW
#define T_LROW 18       // table doubles per locus: the T_NCODE log-pmfs, then the T_K expected terms
#endif
#ifndef T_BL
#define T_BL 640        // locus slots per chunk (the table is T_BL * T_LROW * 8 B = 90 KB of LDS); the last slot is all zeros
#endif
#define T_BLU (T_BL - 1)  // loci per chunk
#define T_BC 1024       // cells per block == threads per workgroup
#define T_THREADS 1024
static_assert(T_BC == T_ROWS_PER_TILE, "cellector_engine_info derives the lookup count from this");
#define T_SB_MAX 4      // cell blocks per workgroup sharing one staged table (2 or 4: chosen per launch)
#define T_GROUPS_MAX 64 // upper bound of the chunk groups of a launch
#define T_GROUPS 8      // chunk groups beyond this many are charged for their partial sums (tiled_build's cost model)
#define T_NE 15         // entries per cell of a slice held in registers (two 16-byte loads); longer slices: slow path
// A u16 entry = n-1 << 14 | locus slot << 4 | code: log-pmf at table[slot * T_LROW + code], expected term at
// table[slot * T_LROW + T_NCODE + (n-1)].
#define T_NULL ((uint16_t)(T_BLU << 4))  // padding entry: code 0, n-1 = 0 of the zero slot
// A slice in `tiles` is 64 rows of K+1 u16 (K odd): row i = [cell (0..1023) that lane i works for, K entries of that cell,
// padded with T_NULL].  Tile header (fixed stride, in u16 units): 16 slices x {u64 first u16 of the slice in `tiles`,
// u32 K, u32 pad}.
#define T_HDR 128
#define TAB_ELEMS ((uint64_t)T_LROW * T_BL)  // table doubles per chunk

// code = n(n+1)/2 - 1 + ref:  n=1: (1,0)(0,1)  n=2: (2,0)(1,1)(0,2)  n=3: (3,0)..(0,3)  n=4: (4,0)..(0,4)
__device__ __constant__ uint8_t T_A_OF[T_NCODE] = {1, 0, 2, 1, 0, 3, 2, 1, 0, 4, 3, 2, 1, 0};
__device__ __constant__ uint8_t T_R_OF[T_NCODE] = {0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4};

__device__ __forceinline__ bool ent_regular(uint64_t e)
{
    const uint32_t n = ENT_ALT(e) + ENT_REF(e);
    return n >= 1u && n <= (uint32_t)T_K;
}
__device__ __forceinline__ uint32_t ent_code(uint64_t e)
{
    const uint32_t r = ENT_REF(e), n = ENT_ALT(e) + r;
    return n * (n + 1u) / 2u - 1u + r;
}

// ---------------------------------------------------------------------------------------------------------
// tables, laid out [chunk][locus slot][18] so that a chunk is one contiguous block the tile kernel copies straight into
// LDS: per locus the 14 log-pmfs, then (PAIRS) the 4 expected terms.  All zero for masked loci (alpha < 0), for the
// padding beyond L and in the last slot of every chunk (the padding entries' target).
// ---------------------------------------------------------------------------------------------------------
// In an EM iteration the kernel is the iteration's FIRST one: it then also forms alpha/beta (init_alpha_betas,
// main.rs:598-611: alpha_l = (S_alt_l + 1) - sum over excluded cells, the subtrahend being the all-reduced ALT_MIN plane of
// the previous iteration; masked loci get alpha = -1), stores them for the other kernels and resets the iteration's counters.
struct ab_src_t {
    const double *s_alt, *s_ref, *alt_min, *ref_min;  // null s_alt: alpha/beta are read from `ab`
    const uint8_t *mask;
    double2 *ab_out;
    double *xl_counters;
    uint32_t *d_counters, *tile_work;
    uint32_t n_work;
};
// Six waves per 64 loci, each with its share of a locus' 18 values (about equal arithmetic): one thread per locus left
// three waves per SIMD, too few to hide the dependent divisions and logs (75 us at 200k loci).
#define TB_PARTS 6
template <bool PAIRS>
__global__ __launch_bounds__(64 * TB_PARTS) void k_build_tables(uint64_t L, uint32_t nj, const double2 *__restrict__ ab,
                                                                const double *__restrict__ lf, double *__restrict__ tab,
                                                                ab_src_t src)
{
    if (src.s_alt && blockIdx.x == 0) {
        if (threadIdx.x < LC_COUNTERS) src.xl_counters[threadIdx.x] = 0.0;
        if (threadIdx.x < 8) src.d_counters[threadIdx.x] = 0u;
        for (uint32_t i = threadIdx.x; i < src.n_work; i += blockDim.x) src.tile_work[i] = 0u;
    }
    const int part = threadIdx.x >> 6;  // wave-uniform
    const uint64_t t = (uint64_t)blockIdx.x * 64 + (threadIdx.x & 63);  // (chunk, slot)
    if (t >= (uint64_t)nj * T_BL) return;
    const uint64_t chunk = t / T_BL, slot = t % T_BL;
    const uint64_t l = chunk * T_BLU + slot;
    double2 p = make_double2(-1.0, -1.0);
    if (slot < T_BLU && l < L) {
        if (src.s_alt) {
            p.x = (src.s_alt[l] + 1.0) - src.alt_min[l];
            p.y = (src.s_ref[l] + 1.0) - src.ref_min[l];
            if (!src.mask[l]) p.x = p.y = -1.0;
            if (part == 0) src.ab_out[l] = p;
        } else {
            p = ab[l];
        }
    }
    const bool live = p.x >= 0.0;
    double *row = tab + chunk * TAB_ELEMS + slot * T_LROW;
    // code w <-> (alt, ref) as in T_A_OF / T_R_OF, written out so that the products unroll
#define TB_PMF(W, A, R) row[W] = live ? dm_log_bb_pmf(lf, p.x, p.y, A, R) : 0.0
#define TB_EXP(N) row[T_NCODE + N - 1] = (PAIRS && live) ? dm_expected_log_pmf(lf, p.x, p.y, N) : 0.0
    switch (part) {
    case 0: TB_EXP(4u); TB_PMF(0, 1u, 0u); break;
    case 1: TB_EXP(3u); TB_PMF(1, 0u, 1u); break;
    case 2: TB_EXP(2u); TB_PMF(2, 2u, 0u); TB_PMF(3, 1u, 1u); break;
    case 3: TB_EXP(1u); TB_PMF(4, 0u, 2u); TB_PMF(5, 3u, 0u); break;
    case 4: TB_PMF(6, 2u, 1u); TB_PMF(7, 1u, 2u); TB_PMF(8, 0u, 3u); TB_PMF(9, 4u, 0u); break;
    default: TB_PMF(10, 3u, 1u); TB_PMF(11, 2u, 2u); TB_PMF(12, 1u, 3u); TB_PMF(13, 0u, 4u); break;
    }
#undef TB_PMF
#undef TB_EXP
}

// ---------------------------------------------------------------------------------------------------------
// cell pass over the tiles
// ---------------------------------------------------------------------------------------------------------
#ifndef TILE_ABL
#define TILE_ABL 0  // ablation builds of the tile kernel (tools/gpu_ab.sh; wrong values, timings only): 1 no lookups, 2 rows loaded once, 3 no table
                    // re-staging, 4 conflict-free lookups, 5 = 1 + 3, 6 conflict-free accumulator updates, 7 no barriers and no re-staging, 8 no
                    // accumulator access, 9 rows and slice headers loaded once, 10 header loads of the column's first chunk only (cache hits)
#endif
// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait
// for the prefetch loads in flight and make them synchronous.
#ifndef TILE_PFD
// Prefetch distance of the tile rows in steps (a column of four blocks; two blocks: always 2) = number of row buffers.
// 4 (a whole chunk ahead), round 2 at cfg4: the kernel itself 1.955 -> 1.87 ms, but at 123 VGPRs (97 with two buffers) its four
// waves per SIMD take the whole register file, the overflow kernels of the side stream find no room beside it and run behind it
// instead: cell pass 2.02 -> 2.26 ms.  3: three buffers rotating over the four steps of a chunk, the chunk loop written out three
// times (the buffer of a step must be known at compile time).
#define TILE_PFD 2
#endif
#ifndef TILE_PFD_T2
#define TILE_PFD_T2 2  // ... of the tier-2 tiles (deep coverage): short rows, little work per step
#endif
#ifndef TILE_PINNED
#define TILE_PINNED 1
#endif
#if TILE_ABL == 7  /* ablation: no chunk barriers (and the table staged once) */
#define TILE_BARRIER() do { } while (0)
#else
#define TILE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// Table / entry geometry of a tile set.  geo_reg: the regular entries (totals 1..T_K): 18 doubles per locus, u16 entry =
// n-1 << 14 | slot << 4 | code.  geo_t2<NMAX>: the tier-2 tiles of a deep-coverage matrix (totals 5..NMAX, tiled_build):
// NMAX = 8: 30 log-pmfs + 4 expected terms per locus, 338 loci per chunk, entry = n-5 << 14 | slot << 5 | pair;
// NMAX = 6: 13 + 2 doubles per locus, 767 loci per chunk, entry = n-5 << 14 | slot << 4 | pair.
struct geo_reg {
    static constexpr uint32_t LROW = T_LROW, BL = T_BL, NCODE = T_NCODE, SHIFT = 4, SMASK = 1023u, CMASK = 15u;
    static constexpr uint32_t BLU = BL - 1, NLO = 1, NHI = T_K;  // loci per chunk (the last slot is all zeros); totals covered
};
template <int NMAX>
struct geo_t2 {
    static_assert(NMAX == 6 || NMAX == 8, "tier-2 tile geometries");
    static constexpr uint32_t NCODE = NMAX == 8 ? 30 : 13, NE = NMAX - 4, LROW = NCODE + NE;
    static constexpr uint32_t BL = NMAX == 8 ? 339 : 768, SHIFT = NMAX == 8 ? 5 : 4, SMASK = NMAX == 8 ? 511u : 1023u,
                              CMASK = NMAX == 8 ? 31u : 15u;
    static constexpr uint32_t BLU = BL - 1, NLO = 5, NHI = NMAX;
};

struct __attribute__((packed, aligned(4))) tile_u4 { uint32_t x, y, z, w; };  // 16-byte load at a 4-byte aligned address

template <bool EXPECTED, int T_SB, class G = geo_reg>
__global__ __launch_bounds__(T_THREADS, 4) void k_tile_ll(uint32_t nb, uint32_t nj, uint32_t cpg, uint32_t groups,
                                                          uint32_t n_cols, uint32_t *__restrict__ work /*[groups], zeroed*/,
                                                          const uint16_t *__restrict__ thdr,
                                                          const uint16_t *__restrict__ tiles,
                                                          const double *__restrict__ tab, uint64_t npad,
                                                          double *__restrict__ part_ll, double *__restrict__ part_ell)
{
    // PERSISTENT workgroups, one per CU (the LDS footprint allows no second one): a workgroup belongs to one group of locus
    // chunks and keeps fetching columns of T_SB consecutive 1024-cell blocks from the group's
    // counter until none is left.  A grid of short-lived workgroups instead leaves a CU idle whenever the next one cannot
    // start because waves of the small overflow kernels running beside this one still hold registers there (measured:
    // 2.2 ms alone, 2.7 ms next to them).
    // Per chunk the table is staged ONCE in LDS and the T_SB tiles are walked one after the other; wave w takes slice w of
    // each tile.  Everything a step needs was requested two steps earlier: in every step the values loaded before are
    // consumed FIRST, then the next loads are issued as straight-line, unconditional instructions (indices are clamped
    // instead of guarded).
    using tab_t = typename std::conditional<EXPECTED, double2, double>::type;  // (log-pmf, expected) sums of a cell
    constexpr uint32_t TAB_U = G::LROW * G::BL * sizeof(double) / 16;  // 16-byte units per chunk table
    constexpr int NP = (TAB_U + T_THREADS - 1) / T_THREADS;          // units per thread (the last one partial)
    static_assert(NP == 6, "table prefetch registers are written out by hand");
    __shared__ __attribute__((aligned(16))) double s_tab[G::LROW * G::BL];
    static_assert(G::LROW * G::BL % 2 == 0, "whole 16-byte units");
    __shared__ tab_t s_acc[T_SB * T_BC];  // per-cell sums of the workgroup's blocks
    __shared__ uint32_t s_col;
    const uint32_t tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    // group = linear block id mod groups (the workgroups of a group walk its chunks in step: each XCD's L2 fetches a table
    // chunk once, whatever the number of groups)
    const uint32_t g = blockIdx.x % groups;
    const uint32_t j0 = g * cpg, j1 = min(nj, j0 + cpg);
    if (j0 >= j1) return;
  for (;;) {
    if (tid == 0) s_col = atomicAdd(&work[g], 1u);
    __syncthreads();  // (also: the previous column's partial sums have been read out of s_acc)
    const uint32_t col = s_col;
    if (col >= n_cols) break;
    const uint32_t b0 = col * T_SB;
#pragma unroll
    for (int s = 0; s < T_SB; s++) {
        if constexpr (EXPECTED) s_acc[s * T_BC + tid] = make_double2(0.0, 0.0);
        else s_acc[s * T_BC + tid] = 0.0;
    }

    // explicit registers (an indexed local array ends up in scratch memory)
    double2 p_t0, p_t1, p_t2, p_t3, p_t4, p_t5;
#define TABLE_PREFETCH(J)                                                                                        \
    do {                                                                                                         \
        const double2 *src__ = reinterpret_cast<const double2 *>(tab) + (uint64_t)(J) * TAB_U + tid;             \
        p_t0 = src__[0];                                                                                         \
        p_t1 = src__[T_THREADS];                                                                                 \
        p_t2 = src__[2 * T_THREADS];                                                                             \
        p_t3 = src__[3 * T_THREADS];                                                                             \
        p_t4 = src__[4 * T_THREADS];                                                                             \
        p_t5 = src__[5 * T_THREADS]; /* partial: reads into the next chunk / the tail pad */                     \
    } while (0)

    // Software pipeline over the steps t = (chunk, block) of this workgroup: the rows of step t+2 are requested in step
    // t; the slice header they need (wave-uniform: scalar loads) is requested in step t-1.  Two buffers serve the even
    // and the odd steps; a buffer is consumed and then refilled in place.
    // lo: cell + entries 0..6, hi: entries 7..14; base = first u16 of the slice in `tiles` (wave-uniform, like k: both stay in
    // scalar registers — the row's own address is only needed again by the rare slice with more than T_NE entries per cell)
    struct row_t { tile_u4 lo, hi; uint32_t k; uint64_t base; };
    // slice header of a step (beyond the last chunk: the last chunk's; a workgroup at the ragged end re-reads the last block)
    const uint32_t wv_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)wv);
#define HDR_LOAD(H, J, S) /* S: a compile-time block number; J: the chunk (clamped by the caller) */           \
    do {                                                                                                         \
        /* The slices of a tile are sorted by length: wave w takes slice w, 15-w, w+8, 7-w (mod 16) of the      */ \
        /* chunk's tiles, so that all waves carry about the same load between two chunk barriers.  (Everything  */ \
        /* but the chunk is invariant in the chunk loop: a step forms the address with an add and a shift.)     */ \
        const uint32_t rot__ = (wv_s + 8u * ((uint32_t)(S) >> 1)) & 15u, sl__ = ((S) & 1u) ? 15u - rot__ : rot__; \
        if (TILE_ABL != 9 || first_rows) /* ablation 9: headers loaded in the prologue only; 10: always the column's first chunk's */ \
        (H) = reinterpret_cast<const uint4 *>(thdr + ((uint64_t)min(b0 + (uint32_t)(S), nb - 1) * nj + (TILE_ABL == 10 ? j0 : (J))) * T_HDR)[sl__]; \
    } while (0)
    // this lane's row of the slice with header H.  A row of less than 8 u16 is covered by the first load: the second
    // one then repeats it (never consumed) instead of reading far beyond the row.
#define ROW_LOAD(E, H)                                                                                           \
    do {                                                                                                         \
        (E).k = (SKIP_EMPTY && (H).w) ? 0u : (H).z; /* 0: a slice without entries (tier-2 tiles): the step skips it */ \
        (E).base = ((uint64_t)(H).y << 32) | (H).x;                                                              \
        /* (a wave-uniform 64-bit base in scalar registers plus a 32-bit byte offset per lane: the loads take the  */ \
        /*  scalar-base form, and the address costs one multiply instead of 64-bit vector arithmetic)               */ \
        const char *sb__ = reinterpret_cast<const char *>(tiles + (E).base);                                     \
        const uint32_t off__ = __umul24(lane, ((H).z + 1u) * 2u); /* (full-rate 24-bit multiply) */                     \
        if ((!(TILE_ABL == 2 || TILE_ABL == 9) || first_rows) && !(SKIP_EMPTY && (H).w)) { /* ablations 2, 9: rows loaded in the prologue only */ \
            (E).lo = *reinterpret_cast<const tile_u4 *>(sb__ + off__);                                           \
            (E).hi = *reinterpret_cast<const tile_u4 *>(sb__ + ((H).z > 7u ? 16u : 0u) + off__);                 \
        }                                                                                                        \
    } while (0)

    // PFD = prefetch distance in steps = number of pipeline buffers.  With four blocks per column every block of the chunk has
    // a buffer of its own and the rows are requested a whole chunk ahead.
    // The tier-2 tiles of a deep matrix hold 0.4 entries per row: ten of a tile's sixteen slices (sorted by length) have none, and a
    // wave whose slice is marked so in its header skips the step — no row loads, no lookups, no accumulator update: that kernel alone
    // 1.90 -> 1.46 ms at 10^6 cells x 200k loci deep.  (It costs 6 VGPRs — 106: the per-entry kernels of the side stream, 88, then
    // find no room beside it and wait for it; the iteration still gains, 5.4 -> 5.0 ms.  Sharing ONE all-padding slice between the
    // empty slices instead, with the kernel unchanged, would update the same accumulators from several waves: not an option.)
    constexpr bool SKIP_EMPTY = G::NLO != 1;
    constexpr uint32_t PFD = T_SB == 4 ? (uint32_t)(G::NLO == 1 ? TILE_PFD : TILE_PFD_T2) : 2u;
    static_assert(PFD >= 2 && PFD <= 4, "row buffers");
    uint4 h0, h1, h2, h3;   // at the top of step t (t % PFD == 0): slice headers of steps t+PFD (h0), t+PFD+1 (h1), ...
    row_t e0, e1, e2, e3;   // at the top of step t (t % PFD == 0): rows of steps t (e0), t+1 (e1), ...
    // The table loads are the OLDEST requests when the chunk loop is entered, like on its back edge, so that the wait
    // for them is a counted vmcnt that leaves the younger row requests in flight (the scheduler must not move them
    // behind the loads below: vmcnt counts in issue order).
    bool first_rows = true;  // (ablations 2, 9)
    TABLE_PREFETCH(j0);
    __builtin_amdgcn_sched_barrier(0);
#define HDR_STEP(H, T) HDR_LOAD(H, min(j0 + (uint32_t)(T) / T_SB, j1 - 1), (uint32_t)(T) % T_SB)  /* header of step T of the column */
    HDR_STEP(h0, 0);
    HDR_STEP(h1, 1);
    if constexpr (PFD >= 3) HDR_STEP(h2, 2);
    if constexpr (PFD == 4) HDR_STEP(h3, 3);
    ROW_LOAD(e0, h0);
    ROW_LOAD(e1, h1);
    if constexpr (PFD >= 3) ROW_LOAD(e2, h2);
    if constexpr (PFD == 4) ROW_LOAD(e3, h3);
    HDR_STEP(h0, PFD);
    HDR_STEP(h1, PFD + 1);
    if constexpr (PFD >= 3) HDR_STEP(h2, PFD + 2);
    if constexpr (PFD == 4) HDR_STEP(h3, PFD + 3);
#undef HDR_STEP
    first_rows = false;

    // one step: block S of the current chunk, pipeline buffers E / H
    // lookup of entry KK of the row = u16 number KK + 1 = half (KK + 1) & 1 of dword (KK + 1) >> 1.  The slice's K is odd
    // and wave-uniform, so the lookups come in pairs behind one scalar branch.
#if TILE_ABL == 1 || TILE_ABL == 5  /* ablation: no table lookups */
#define TILE_LOOKUP(V, E16) do { if constexpr (EXPECTED) V = make_double2((double)(E16), 1.0); else V = (double)(E16); } while (0)
#elif TILE_ABL == 4  /* ablation: lookups free of bank conflicts (a lane keeps to its own bank pair; wrong values) */
#define TILE_LOOKUP(V, E16)                                                                                      \
    do {                                                                                                         \
        const uint32_t b__ = (((E16) >> 4) & 127u) * 32u + (lane & 31u), c__ = (((E16) >> 5) & 127u) * 32u + (lane & 31u); \
        if constexpr (EXPECTED) V = make_double2(s_tab[b__], s_tab[c__ + 4096u]);                                \
        else V = s_tab[b__];                                                                                     \
    } while (0)
#else
#define TILE_LOOKUP(V, E16)                                                                                      \
    do {                                                                                                         \
        const uint32_t b__ = (((E16) >> G::SHIFT) & G::SMASK) * G::LROW;                                         \
        if constexpr (EXPECTED) V = make_double2(s_tab[b__ + ((E16) & G::CMASK)], s_tab[b__ + G::NCODE + ((E16) >> 14)]); \
        else V = s_tab[b__ + ((E16) & G::CMASK)];                                                                \
    } while (0)
#endif
#define TILE_RD(V, KK)                                                                                           \
    tab_t V;                                                                                                     \
    do {                                                                                                         \
        const uint32_t idx__ = (((KK) + 1) & 1) ? (w__[((KK) + 1) >> 1] >> 16) : (w__[((KK) + 1) >> 1] & 0xffffu); \
        TILE_LOOKUP(V, idx__);                                                                                   \
    } while (0)
#define TILE_ADD(V)                                                                                              \
    do {                                                                                                         \
        if constexpr (EXPECTED) { a_ll__ += (V).x; a_el__ += (V).y; }                                            \
        else a_ll__ += (V);                                                                                      \
    } while (0)
    // one level of the written-out lookup chain: request the next pair, THEN add the pair requested one level earlier (the
    // sums still run in entry order), so that four lookups of a wave are in flight instead of two
/* (TILE_PIN: without it the compiler hoists the two adds, common to both arms, above the branch — right behind the    */
/*  previous level's requests, which they then wait for: no lookup of the next level would be in flight meanwhile.  */
/*  The empty asm makes the value a different one in this arm and, with its memory clobber, keeps the requests above */
/*  it.)                                                                                                            */
#if TILE_PINNED
#define TILE_PIN(V)                                                                                              \
    do {                                                                                                         \
        if constexpr (EXPECTED) asm volatile("" : "+v"((V).x), "+v"((V).y) : : "memory");                        \
        else asm volatile("" : "+v"(V) : : "memory");                                                            \
    } while (0)
#else
#define TILE_PIN(V) do { } while (0)
#endif
#define TILE_LEVEL(KA, KB, PA, PB, BODY)                                                                         \
    if (K__ > (KA)) {                                                                                            \
        TILE_RD(v##KA, KA); TILE_RD(v##KB, KB);                                                                  \
        TILE_PIN(PA); TILE_PIN(PB);                                                                              \
        TILE_ADD(PA); TILE_ADD(PB);                                                                              \
        BODY                                                                                                     \
    } else { TILE_ADD(PA); TILE_ADD(PB); }
#define TILE_STEP(S, E, H)                                                                                       \
    do {                                                                                                         \
        /* 1. consume the row requested PFD steps ago: u16 number i of the row sits in half i & 1 of dword i >> 1 */ \
        const uint32_t w__[8] = {(E).lo.x, (E).lo.y, (E).lo.z, (E).lo.w, (E).hi.x, (E).hi.y, (E).hi.z, (E).hi.w}; \
        const uint32_t K__ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(E).k); /* wave-uniform, odd */       \
        const uint64_t cur_base__ = (E).base;                                                                    \
        const uint32_t cell__ = w__[0] & 0xffffu;                                                                \
        /* 2. issue the next requests: rows of step t+PFD (their header is here), header of step t+2 PFD */      \
        ROW_LOAD(E, H);                                                                                          \
        HDR_LOAD(H, min(j + ((S) + 2 * PFD) / T_SB, j1 - 1), ((S) + 2 * PFD) % T_SB);                            \
        /* 3. the cell's accumulator is requested first (LDS answers in order: it has landed when the sums are done), */ \
        /*    then this lane's cell of the slice: K lookups for every lane (padding entries hit the zero row) */  \
        if (!(SKIP_EMPTY && K__ == 0u)) {                                                                        \
        tab_t acc__;                                                                                             \
        if constexpr (TILE_ABL == 8) { if constexpr (EXPECTED) acc__ = make_double2(0.0, 0.0); else acc__ = 0.0; }        \
        else acc__ = s_acc[(S) * T_BC + (TILE_ABL == 6 ? tid : cell__)];                                         \
        double a_ll__ = 0.0, a_el__ = 0.0;                                                                       \
        /* (written out: a loop with an early exit gets re-rolled and then selects its register at run time) */  \
        TILE_RD(v0, 0);                                                                                          \
        if (K__ > 1) {                                                                                           \
            TILE_RD(v1, 1); TILE_RD(v2, 2);                                                                      \
            TILE_PIN(v0);                                                                                        \
            TILE_ADD(v0);                                                                                        \
            TILE_LEVEL(3, 4, v1, v2,                                                                             \
            TILE_LEVEL(5, 6, v3, v4,                                                                             \
            TILE_LEVEL(7, 8, v5, v6,                                                                             \
            TILE_LEVEL(9, 10, v7, v8,                                                                            \
            TILE_LEVEL(11, 12, v9, v10,                                                                          \
            TILE_LEVEL(13, 14, v11, v12,                                                                         \
                TILE_ADD(v13); TILE_ADD(v14);                                                                    \
                for (uint32_t k = T_NE; k < K__; k++) { /* rare: a slice with more than T_NE entries per cell */ \
                    tab_t v__;                                                                                   \
                    TILE_LOOKUP(v__, (uint32_t)(tiles + cur_base__ + lane * (K__ + 1u))[k + 1]);                 \
                    TILE_ADD(v__);                                                                               \
                } ))))))                                                                                         \
        } else TILE_ADD(v0);                                                                                     \
        /* 4. add the tile's sums to the cell's accumulator (one lane per cell and tile; tiles of the same block */ \
        /*    are separated by the chunk barriers).  A read and a write: two ds_add_f64 measured 15 % slower.      */ \
        if constexpr (EXPECTED) { acc__.x += a_ll__; acc__.y += a_el__; }                                        \
        else acc__ += a_ll__;                                                                                    \
        if (TILE_ABL != 8 || a_ll__ == 12345.678) s_acc[(S) * T_BC + (TILE_ABL == 6 ? tid : cell__)] = acc__;   \
        }                                                                                                        \
    } while (0)

    static_assert(T_SB == 2 || T_SB == 4, "even and odd steps use different pipeline buffers");
    // the table of chunk j into LDS between two barriers, the next chunk's table requested
#if TILE_ABL == 3 || TILE_ABL == 5 || TILE_ABL == 7  /* ablation: table staged for the first chunk only */
#define TILE_STAGE_IF if (j == j0)
#define TILE_NEXT_TABLE() do { } while (0)
#else
#define TILE_STAGE_IF
#define TILE_NEXT_TABLE() TABLE_PREFETCH(min(j + 1, j1 - 1))
#endif
#define TILE_STAGE()                                                                                             \
    do {                                                                                                         \
        TILE_BARRIER(); /* every wave is done with the previous chunk's table (and, first time, s_acc is zeroed) */ \
        TILE_STAGE_IF {                                                                                          \
            double2 *dst = reinterpret_cast<double2 *>(s_tab) + tid;                                             \
            dst[0] = p_t0;                                                                                       \
            dst[T_THREADS] = p_t1;                                                                               \
            dst[2 * T_THREADS] = p_t2;                                                                           \
            dst[3 * T_THREADS] = p_t3;                                                                           \
            dst[4 * T_THREADS] = p_t4;                                                                           \
            if (tid + 5 * T_THREADS < TAB_U) dst[5 * T_THREADS] = p_t5;                                          \
        }                                                                                                        \
        TILE_NEXT_TABLE();                                                                                       \
        TILE_BARRIER(); /* table visible */                                                                      \
    } while (0)
    if constexpr (T_SB == 4 && PFD == 3) {
        // three row buffers over four steps per chunk: the assignment repeats every three chunks
        for (uint32_t j = j0;;) {
            TILE_STAGE();
            TILE_STEP(0, e0, h0); TILE_STEP(1, e1, h1); TILE_STEP(2, e2, h2); TILE_STEP(3, e0, h0);
            if (++j >= j1) break;
            TILE_STAGE();
            TILE_STEP(0, e1, h1); TILE_STEP(1, e2, h2); TILE_STEP(2, e0, h0); TILE_STEP(3, e1, h1);
            if (++j >= j1) break;
            TILE_STAGE();
            TILE_STEP(0, e2, h2); TILE_STEP(1, e0, h0); TILE_STEP(2, e1, h1); TILE_STEP(3, e2, h2);
            if (++j >= j1) break;
        }
    } else {
        for (uint32_t j = j0; j < j1; j++) {
            TILE_STAGE();
            TILE_STEP(0, e0, h0);
            TILE_STEP(1, e1, h1);
            if constexpr (T_SB == 4 && PFD == 4) {
                TILE_STEP(2, e2, h2);
                TILE_STEP(3, e3, h3);
            } else if constexpr (T_SB == 4) {
                TILE_STEP(2, e0, h0);
                TILE_STEP(3, e1, h1);
            }
        }
    }
#undef TILE_STAGE
#undef TILE_STAGE_IF
#undef TILE_NEXT_TABLE
    TILE_BARRIER();
#pragma unroll
    for (int s = 0; s < T_SB; s++) {
        if (b0 + s < nb) {
            const uint64_t c = (uint64_t)g * npad + (uint64_t)(b0 + s) * T_BC + tid;
            const tab_t a = s_acc[s * T_BC + tid];
            if constexpr (EXPECTED) { part_ll[c] = a.x; part_ell[c] = a.y; }
            else part_ll[c] = a;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);  // the pipeline's last (clamped, unused) loads have landed before the registers are reused
  }
#undef TILE_STEP
#undef TILE_LEVEL
#undef TILE_PIN
#undef TILE_ADD
#undef TILE_RD
#undef TILE_LOOKUP
#undef ROW_LOAD
#undef HDR_LOAD
#undef TABLE_PREFETCH
}

// ---------------------------------------------------------------------------------------------------------
// overflow entries (alt+ref == 0 or > T_K; 0.8 % at cfg4).  Two consumers, two evaluations:
//   cell side   k_ovf_cell_direct: a thread per cell row evaluates the row's overflow entries itself from (alpha, beta) of
//               their loci (a gather out of an L x 16 B table that stays in L2) and takes the expected terms from a compact
//               E table (k_ovf_tables_e).  Runs on the side stream beside the tile kernel.
//   locus side  the locus pass needs every overflow entry's log-pmf in by-locus order: k_ovf_tables builds per-locus
//               cumulative log tables, k_locus_finalize evaluates lnC + LA[alt] + LB[ref] - LAB[n] per entry as it walks
//               the locus (the lanes of a locus share its table row).  Only the EM pass needs the tables, and only at the
//               locus finalize: the table kernel follows the cell side on the side stream.
// ---------------------------------------------------------------------------------------------------------
#define LF_LANES 16  // lanes that share a locus (k_locus_finalize) or a row (k_ovf_cell_wide)
template <typename T>
__device__ __forceinline__ T group16_sum(T v)
{
#pragma unroll
    for (int m = LF_LANES / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, LF_LANES);
    return v;
}
#define OV_NT 18  // cumulative tables cover counts 0..17; larger counts take the generic device_math path
#define OV_NE 17  // expected terms E(n) tabulated for n = 4..17
#define OV_FAST_N DM_CHUNK  // the cell side's fast kernel takes totals up to this (99 % of the overflow entries)

// tier 2 (see k_t2_tables below)
#define T2_NMIN 5u
#define T2_NMAX 8u
#define T2_NCODE 30    // (alt, ref) pairs with 5 <= alt+ref <= 8; code = n(n+1)/2 - 15 + ref
#define T2_CSTRIDE 32  // u32 counters per locus (hist_all2, cnt2)
#define T2_ROW 48      // table doubles per locus
static_assert(T2_NMAX == (unsigned)OV_FAST_N, "the tier lists take the totals above tier 2");
__device__ __forceinline__ bool t2_total(uint32_t n) { return n - T2_NMIN <= T2_NMAX - T2_NMIN; }
__device__ __forceinline__ uint32_t t2_code(uint32_t n, uint32_t r) { return n * (n + 1u) / 2u - 15u + r; }
// position of the pair's log-pmf in the locus' table row; its sector's first double is E(n)
__device__ __forceinline__ uint32_t t2_pos(uint32_t n, uint32_t r)
{
    const uint32_t hi = r >= 7u ? 1u : 0u;
    return ((n - T2_NMIN) + (n == 8u ? 1u : 0u) + hi) * 8u + 1u + (hi ? r - 7u : r);
}

// rare cases kept out of line so that the common path stays small
__device__ __noinline__ double ov_slow_log_pmf(const double *lf, double alpha, double beta, uint32_t a, uint32_t r)
{
    return dm_log_bb_pmf(lf, alpha, beta, a, r);
}

// ln sum_k pmf(k)^2 (stats.rs:8-22) for 4 <= n <= OV_NE by the pmf ratio recurrence of dm_expected_log_pmf, with pmf(0) built
// from per-factor ratios (each in (0, 1]: no overflow of long products at these n)
// 1 / y to about an ulp: v_rcp_f64 (good to ~27 bits) and one Newton step; no scaling / fix-up for subnormals or
// infinities, which the arguments here (sums and products of counts) never are
__device__ __forceinline__ double ov_rcp(double y)
{
    const double r = __builtin_amdgcn_rcp(y);
    return __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
}
__device__ __forceinline__ double ov_expected_rec(double alpha, double beta, uint32_t n)
{
    // x / y as x * ov_rcp(y) instead of the ~25-instruction IEEE division: the few lanes that carry an E(n) keep their
    // whole wave waiting, and 2 n ulp-sized errors are far inside the pass' tolerance
    const double ab = alpha + beta;
    double p = 1.0;
    for (uint32_t j = 0; j < n; ++j) p *= (beta + (double)j) * ov_rcp(ab + (double)j);
    double s = p * p;
    for (uint32_t k = 0; k < n; ++k) {
        p *= ((double)(n - k) * (alpha + (double)k)) * ov_rcp((double)(k + 1) * (beta + (double)(n - k - 1)));
        s += p * p;
    }
    return log(s);
}

// Per-locus overflow table (OV_ROW doubles): [0..17] LA[i] = sum_{m<i} ln(alpha+m), [18..35] LB, [36..53] LAB,
// [64..] E(n) for n = 4..OV_NE.  Two dense kernels (a fused one left most lanes of a wave idle while a few ran the long
// loops): k_ovf_tables = one thread per (locus, family A / B / AB) running the 17 logs and their prefix sums;
// k_ovf_tables_e = one thread per (locus, n) for the totals n that occur among the locus' overflow entries (nmask: static).
#define OV_ROW 128
#define OV_EOFF 64
#define OV_REC 8  // the cell side's per-locus record: alpha, beta, E(5..8), pad = ONE 64-byte sector per overflow entry
__global__ __launch_bounds__(256) void k_ovf_tables(uint64_t L, const double2 *__restrict__ ab, const uint32_t *__restrict__ nmask,
                                                    double *__restrict__ otab)
{
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t l = idx / 3;
    if (l >= L) return;
    const int fam = (int)(idx % 3);
    const double2 p = ab[l];
    double *row = otab + l * OV_ROW + fam * OV_NT;
    if (!(p.x >= 0.0)) {
        if (fam == 0) row[0] = -1.0;  // marks a masked locus
        return;
    }
    // only as far as the largest tabulated total among the locus' overflow entries (nmask, static): the logs are the
    // kernel's whole cost, and most loci stop at a total of 6 or 7 of the 17
    const uint32_t m = nmask[l];
    const int top = m ? 4 + (31 - __clz((int)m)) : 0;
    const double x0 = fam == 0 ? p.x : (fam == 1 ? p.y : p.x + p.y);
    double acc = 0.0;
    row[0] = 0.0;
    for (int i = 0; i < OV_NT - 1 && i < top; i++) {
        acc += log(x0 + (double)i);
        row[i + 1] = acc;
    }
}
__global__ __launch_bounds__(256) void k_ovf_tables_e(uint64_t L, const double2 *__restrict__ ab,
                                                      const uint32_t *__restrict__ nmask, double *__restrict__ otab,
                                                      double *__restrict__ etab /*[L][OV_REC]: the cell side's record*/)
{
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t l = idx >> 4;  // 16 slots per locus: 14 totals, the last one copies alpha / beta
    if (l >= L) return;
    const uint32_t i = (uint32_t)(idx & 15u);
    if (i == 15u) {  // (a masked locus carries its negative alpha along: the cell side tests that)
        if (etab) *reinterpret_cast<double2 *>(etab + l * OV_REC) = ab[l];
        return;
    }
    if (i > (uint32_t)OV_NE - 4u || !((nmask[l] >> i) & 1u)) return;
    const double2 p = ab[l];
    if (!(p.x >= 0.0)) return;
    const double e = ov_expected_rec(p.x, p.y, 4u + i);
    otab[l * OV_ROW + OV_EOFF + i] = e;
    if (etab && i >= 1 && i <= 4) etab[l * OV_REC + 2 + (i - 1)] = e;
}

// The overflow entries' log-pmfs in by-locus order for the locus finalize, shallow-coverage form: a thread per entry
// (neighbouring threads share a locus: the three table words an entry needs come out of L1), on the side stream beside the
// tile kernel; the finalize then reads 8 bytes per entry.  (A deep-coverage matrix, ovf_deep, evaluates them inside
// k_locus_finalize instead: storing and re-reading 2.6e8 values was 4 GB of traffic per iteration at 1M x 200k.)
__global__ __launch_bounds__(256) void k_ovf_values(uint64_t n_ovf, const uint32_t *__restrict__ ovc_locus,
                                                    const uint64_t *__restrict__ ovc_ent,
                                                    const double2 *__restrict__ ab, const double *__restrict__ lf,
                                                    const double *__restrict__ otab, double *__restrict__ lp_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ovf) return;
    const uint32_t l = ovc_locus[i];
    const uint64_t en = ovc_ent[i];
    const double *row = otab + (uint64_t)l * OV_ROW;
    double lp = 0.0;
    if (row[0] >= 0.0) {  // else a masked locus: no PMFData (main.rs:556)
        const uint32_t a = ENT_ALT(en), r = ENT_REF(en), n = a + r;
        if (n == 0) lp = 0.0;  // quirk Q14: exactly zero
        else if (n < (uint32_t)OV_NT) lp = (lf[n] - lf[a] - lf[r]) + (row[a] + row[OV_NT + r] - row[2 * OV_NT + n]);
        else {
            const double2 p = ab[l];
            lp = ov_slow_log_pmf(lf, p.x, p.y, a, r);
        }
    }
    lp_out[i] = lp;
}

// locus of every overflow entry (by-locus order): wave per locus
__global__ __launch_bounds__(256) void k_ovf_locus_ids(uint64_t L, const uint64_t *__restrict__ ovc_ptr, uint32_t *__restrict__ ovc_locus)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t l = wave0; l < L; l += nwaves)
        for (uint64_t i = ovc_ptr[l] + lane; i < ovc_ptr[l + 1]; i += 64) ovc_locus[i] = (uint32_t)l;
}

// nmask[l]: bit (n - 4) set iff an overflow entry of locus l has alt+ref == n, 4 <= n <= OV_NE (static)
__global__ __launch_bounds__(256) void k_ovf_nmask(uint64_t L, const uint64_t *__restrict__ ovc_ptr,
                                                   const uint64_t *__restrict__ ovc_ent, uint32_t *__restrict__ nmask)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t l = wave0; l < L; l += nwaves) {
        uint32_t m = 0;
        for (uint64_t i = ovc_ptr[l] + lane; i < ovc_ptr[l + 1]; i += 64) {
            const uint64_t en = ovc_ent[i];
            const uint32_t n = ENT_ALT(en) + ENT_REF(en);
            if (n >= 4u && n <= (uint32_t)OV_NE) m |= 1u << (n - 4u);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m |= (uint32_t)__shfl_xor((int)m, off, 64);
        if (lane == 0) nmask[l] = m;
    }
}

// cell side of the overflow entries: a thread per cell row evaluates them itself from (alpha, beta) of their loci (a 16-byte
// gather out of a table of L x 16 B that stays in L2) — one log of a ratio of short products per entry — in ascending-locus
// order (sequential: deterministic), and takes the expected terms from the compact E table.  The row's 8-byte entries are
// all the HBM traffic; gathering stored per-entry values instead fetched a 128-byte line per entry (2.3 GB at cfg4 for
// 0.26 GB of values) and slowed the tile kernel beside it more.  k_cell_finalize adds the result to the tile partials.
// (48 VGPRs and no calls: the persistent tile workgroups leave 96 VGPRs per SIMD, so two waves of this kernel fit beside
// them on every SIMD; the entry loop is not unrolled — a row has ~16 overflow entries and hundreds of thousands of rows
// are in flight.  Entries with a total above OV_FAST_N = 8 (1 % of them; longer products, and beyond the tables the generic
// paths: Lanczos ln_gamma, the log-space fold of the expected term) are left to k_ovf_cell_listed, which works off two
// short static lists of exactly those entries.)
// The rows' entries come from a 64-row ELLPACK copy (ovf_ell: entry k of the 64 rows of a group side by side, padded with
// all-ones to the group's longest row): a wave's loads are coalesced and every line is used once.  Reading the CSR row by
// row, a lane per row, re-fetched each row's 128-byte line for every one of its entries (measured 2.1 GB for 0.13 GB).
#define OVF_PAD (~0ull)
// PACKED (a big shard's EM pass): alpha, beta and the expected terms out of ONE 64-byte record per locus (k_ovf_tables_e).
// Gathered from two tables an entry fetches two sectors — 2.0 GB per pass at cfg4 on the fabric the tile kernel streams
// through, 1.0 GB this way.  The variant needs 64 VGPRs, i.e. one wave per SIMD beside the tile kernel: all a big shard's
// launch asks for (residency throttle, launch_overflow_cell); a small shard wants two and keeps the 48-VGPR form.
// FULL (a matrix whose overflow share is large — deep coverage — runs the kernel with the machine to itself): totals up to
// OV_NE (17) are taken here too — chunked products, up to three logs, E(n) from the per-locus table — instead of the tier-0
// list, whose one-thread-per-row walk is built for a handful of entries per row (at 13 % overflow entries that kernel alone
// took 12 ms of a 21 ms iteration at 1M x 200k).
template <bool EXPECTED, bool PACKED, bool FULL>
__global__ __launch_bounds__(256) void k_ovf_cell_direct(
    uint64_t n_rows, const uint64_t *__restrict__ ell_ptr, const uint64_t *__restrict__ ell,
    const double2 *__restrict__ ab, const double *__restrict__ lf, const double *__restrict__ etab,
    const double *__restrict__ otab, double *__restrict__ o_ll, double *__restrict__ o_ell)
{
    static_assert(EXPECTED || !PACKED, "the record is built for the EM pass");
    static_assert(!(FULL && PACKED), "the full form runs unthrottled: it takes the unpacked tables");
    const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;  // (rows beyond the end have no lane; their slots are padding)
    double s = 0.0, e = 0.0;
    const uint64_t grp = row >> 6, base = ell_ptr[grp] + (row & 63), end = ell_ptr[grp + 1];
#pragma unroll 1
    for (uint64_t i = base; i < end; i += 64) {
        const uint64_t en = ell[i];
        if (en == OVF_PAD) continue;
        const uint32_t l = ENT_IDX(en), a = ENT_ALT(en), r = ENT_REF(en), n = a + r;
        const double *rec = etab + (uint64_t)l * OV_REC;
        const double2 p = PACKED ? *reinterpret_cast<const double2 *>(rec) : ab[l];
        double ev = 0.0;  // (an overflow entry has a total of 0 or above T_K: the record covers T_K + 1 .. OV_FAST_N)
        if (PACKED) ev = n > (uint32_t)T_K && n <= (uint32_t)OV_FAST_N ? rec[n - 3] : 0.0;
        // masked locus: no PMFData (main.rs:556); 0/0 entry: exactly zero (Q14); a total above OV_FAST_N: k_ovf_cell_listed
        if (!(p.x >= 0.0) || n == 0 || n > (uint32_t)(FULL ? OV_NE : OV_FAST_N)) continue;
        if (FULL && n > (uint32_t)OV_FAST_N) {  // totals 9..17: the tier-0 arithmetic (k_ovf_cell_listed), in the row's order
            s += (lf[n] - lf[a] - lf[r]) + dm_log_beta_ratio(p.x, p.y, a, r);
            if (EXPECTED) e += otab[(uint64_t)l * OV_ROW + OV_EOFF + (n - 4)];
            continue;
        }
        // = dm_log_bb_pmf for these totals: ln C out of the factorial table, one log of a ratio of products of <= 8 factors
        const double abs_ = p.x + p.y;
        double num = 1.0, den = 1.0;
        for (uint32_t k = 0; k < n; ++k) {
            num *= (k < a) ? (p.x + (double)k) : (p.y + (double)(k - a));
            den *= abs_ + (double)k;
        }
        s += (lf[n] - lf[a] - lf[r]) + log(num / den);
        if (EXPECTED) {
            if (PACKED) e += ev;
            else e += n >= 5 ? rec[2 + (n - 5)] : otab[(uint64_t)l * OV_ROW + OV_EOFF + (n - 4)];
        }
    }
    o_ll[row] = s;
    if (EXPECTED) o_ell[row] = e;
}

// Deep coverage (ovf_deep): hundreds of overflow entries per row.  A thread per row walking them one after the other left the
// chip waiting (7.5 ms at 1M x 200k with 13 % overflow entries).  Here 16 lanes share a row: lane j takes the row's entries
// j, j + 16, ... straight out of the overflow CSR (a 128-byte line per 16 lanes), 16 gathers of a row in flight at once, and
// every total up to OV_NE = 17 takes the SAME instruction path — one product loop, one reciprocal, one log: a ratio of two
// products of at most 17 factors stays far inside the f64 range for alpha + beta below 1e17, so no chunking — because a wave
// whose lanes split between a short-total and a long-total path pays for both.  alpha, beta and E(5..8) come out of ONE
// 64-byte record per locus in the EM pass.  The per-lane sums are added by a fixed-shape butterfly (deterministic).
// (Measured and dropped: one launch per locus range whose records fit an L2 — 9.3 -> 11-14 ms: the kernel is bound by its
// arithmetic, not by the gathers.)
template <bool EXPECTED>
__global__ __launch_bounds__(256) void k_ovf_cell_wide(uint64_t n_rows, const uint64_t *__restrict__ ovf_ptr,
                                                       const uint64_t *__restrict__ ovf_ent, const double2 *__restrict__ ab,
                                                       const double *__restrict__ lf, const double *__restrict__ etab,
                                                       const double *__restrict__ otab, double *__restrict__ o_ll,
                                                       double *__restrict__ o_ell)
{
    const uint32_t j = threadIdx.x % LF_LANES;
    const uint64_t row_raw = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LF_LANES;
    const bool in = row_raw < n_rows;
    const uint64_t row = in ? row_raw : n_rows - 1;  // whole waves stay in the butterflies
    const uint64_t beg = ovf_ptr[row], end = in ? ovf_ptr[row + 1] : beg;
    double s = 0.0, e = 0.0;
#pragma unroll 2
    for (uint64_t i = beg + j; i < end; i += LF_LANES) {
        const uint64_t en = ovf_ent[i];
        const uint32_t l = ENT_IDX(en), a = ENT_ALT(en), r = ENT_REF(en), n = a + r;
        const double *rec = etab + (uint64_t)l * OV_REC;
        const double2 p = EXPECTED ? *reinterpret_cast<const double2 *>(rec) : ab[l];
        double ev = 0.0;
        if (EXPECTED) {
            if (n > (uint32_t)T_K && n <= (uint32_t)OV_FAST_N) ev = rec[n - 3];
            else if (n > (uint32_t)OV_FAST_N && n <= (uint32_t)OV_NE) ev = otab[(uint64_t)l * OV_ROW + OV_EOFF + (n - 4)];
        }
        // masked locus: no PMFData (main.rs:556); 0/0 entry: exactly zero (Q14); totals above OV_NE: the generic list
        if (!(p.x >= 0.0) || n == 0 || n > (uint32_t)OV_NE) continue;
        double lp;
        if (p.x + p.y < 1e17) {
            double num = 1.0, den = 1.0, fa = p.x, fb = p.y, fab = p.x + p.y;
            for (uint32_t k = 0; k < n; ++k) {
                const bool isa = k < a;
                num *= isa ? fa : fb;
                den *= fab;
                fa += isa ? 1.0 : 0.0;
                fb += isa ? 0.0 : 1.0;
                fab += 1.0;
            }
            lp = (lf[n] - lf[a] - lf[r]) + log(num * ov_rcp(den));
        } else {
            lp = ov_slow_log_pmf(lf, p.x, p.y, a, r);
        }
        s += lp;
        if (EXPECTED) e += ev;
    }
    s = group16_sum(s);
    if (EXPECTED) e = group16_sum(e);
    if (in && j == 0) {
        o_ll[row] = s;
        if (EXPECTED) o_ell[row] = e;
    }
}

// ---------------------------------------------------------------------------------------------------------
// tier 2: the overflow entries with 5 <= alt+ref <= 8 (99 % of them at vartrix-like coverage) are table-driven as well.
// Under one (alpha_l, beta_l) their log-pmf depends on (alt, ref) only and a locus carries few distinct pairs, so a pass
// evaluates every (locus, pair) that occurs in the shard ONCE (k_t2_tables: 30 pairs + 4 expected terms per locus at most, a
// static histogram says which occur) into a table in global memory, and both consumers read it:
//   cell side   k_t2_cell: a thread per cell row, four entries' lookups in flight — no arithmetic, a 64-byte sector per entry
//               (it used to evaluate a log of a product ratio per entry with one wave per SIMD beside the tile kernel: 1.2 ms
//               at 10^6 cells x 200k loci for 1.6e7 entries, f64-latency bound);
//   locus side  the minority cells' tier-2 entries are COUNTED per (locus, pair) (k_t2_minority walks the excluded cells'
//               overflow rows: integer atomics, exact and order independent) and k_locus_finalize turns counts x table values
//               into the locus' contributions, like it does for the regular entries.  No per-entry values are stored or read.
// The table row of a locus is six 64-byte sectors [E(n), up to seven log-pmfs of total n]: n = 5, 6, 7 (ref 0..6), 7 (ref 7),
// 8 (ref 0..6), 8 (ref 7, 8) — an entry's log-pmf and expected term share a sector.
// What is left for the per-entry paths: totals 0 (quirk Q14) and above 8 (the tier lists on the cell side, `ovx` on the locus side).
// ---------------------------------------------------------------------------------------------------------
// The pairs that occur in the shard and the sectors that hold at least one of them are STATIC: two lists made at ingest
// (k_t2_lists, in locus order).  One thread per listed pair (blocks [0, gp)) or sector (blocks [gp, ...)): dense waves
// whose lanes all run the same loops.  (A thread per table slot — 48 per locus, most of them idle while a few lanes ran
// the long expected-term loop — took 0.41 ms beside the tile kernel at 200k loci, 0.30 ms on a 125k-cell shard.)
// Slots of pairs that do not occur are never written and never read.  A masked locus (alpha < 0) gets zeros: its
// entries add nothing.
template <bool EXPECTED>
__global__ __launch_bounds__(256) void k_t2_tables(uint32_t n_pairs, const uint32_t *__restrict__ plist /*locus << 5 | pair*/,
                                                   uint32_t n_sec, const uint32_t *__restrict__ slist /*locus << 3 | sector*/,
                                                   uint32_t gp, const double2 *__restrict__ ab, const double *__restrict__ lf,
                                                   double *__restrict__ tab2)
{
    if (blockIdx.x < gp) {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i >= n_pairs) return;
        const uint32_t key = plist[i], l = key >> 5, c2 = key & 31u;
        const uint32_t n = 5u + (c2 >= 6u) + (c2 >= 13u) + (c2 >= 21u), r = c2 - t2_code(n, 0u), a = n - r;
        const double2 p = ab[l];
        double v = 0.0;
        if (p.x >= 0.0) {  // = dm_log_bb_pmf for a total of at most DM_CHUNK: one log of a ratio of two short products
            double num = 1.0, den = 1.0, fa = p.x, fb = p.y, fab = p.x + p.y;
#pragma unroll
            for (uint32_t k = 0; k < T2_NMAX; ++k) {  // (one instruction path for every total and split)
                const bool on = k < n, isa = k < a;
                num *= on ? (isa ? fa : fb) : 1.0;
                den *= on ? fab : 1.0;
                fa += isa ? 1.0 : 0.0;
                fb += (on && !isa) ? 1.0 : 0.0;
                fab += 1.0;
            }
            v = (lf[n] - lf[a] - lf[r]) + log(num / den);
        }
        tab2[(uint64_t)l * T2_ROW + t2_pos(n, r)] = v;
    } else if (EXPECTED) {
        const uint32_t i = (blockIdx.x - gp) * 256 + threadIdx.x;
        if (i >= n_sec) return;
        const uint32_t key = slist[i], l = key >> 3, sec = key & 7u;
        const uint32_t n = 5u + (sec >= 1u) + (sec >= 2u) + (sec >= 4u);
        const double2 p = ab[l];
        tab2[(uint64_t)l * T2_ROW + sec * 8u] = p.x >= 0.0 ? ov_expected_rec(p.x, p.y, n) : 0.0;
    }
}

// The same values in the CHUNKED layout of the tier-2 tiles (deep coverage: k_tile_ll<.., geo_t2> reads them out of LDS like the
// regular tables): [chunk][slot][G::LROW] = the pairs' log-pmfs of totals 5..G::NHI, then E(5..G::NHI).  A deep matrix carries
// every pair at nearly every locus, so no lists here: blocks [0, gp) take a (locus, pair) per thread, the others a (locus, total)
// — dense waves that run the same loops (the expected term's recurrence is much longer than a pair's product).  The last slot
// of a chunk and the slots beyond L are never written (zeroed once at build); a masked locus gets zeros.
template <bool EXPECTED, class G>
__global__ __launch_bounds__(256) void k_t2c_tables(uint64_t L, uint32_t gp, const double2 *__restrict__ ab, const double *__restrict__ lf,
                                                    double *__restrict__ tabc)
{
    constexpr uint64_t ELEMS = (uint64_t)G::LROW * G::BL;
    if (blockIdx.x < gp) {
        const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
        const uint64_t l = i / G::NCODE;
        if (l >= L) return;
        const uint32_t c2 = (uint32_t)(i % G::NCODE);
        const uint32_t n = 5u + (c2 >= 6u) + (c2 >= 13u) + (c2 >= 21u), r = c2 - t2_code(n, 0u), a = n - r;
        const double2 p = ab[l];
        double v = 0.0;
        if (p.x >= 0.0) {  // (the arithmetic of k_t2_tables: both sides of a pass add the same bits)
            double num = 1.0, den = 1.0, fa = p.x, fb = p.y, fab = p.x + p.y;
#pragma unroll
            for (uint32_t k = 0; k < T2_NMAX; ++k) {
                const bool on = k < n, isa = k < a;
                num *= on ? (isa ? fa : fb) : 1.0;
                den *= on ? fab : 1.0;
                fa += isa ? 1.0 : 0.0;
                fb += (on && !isa) ? 1.0 : 0.0;
                fab += 1.0;
            }
            v = (lf[n] - lf[a] - lf[r]) + log(num / den);
        }
        tabc[(l / G::BLU) * ELEMS + (l % G::BLU) * G::LROW + c2] = v;
    } else if (EXPECTED) {
        constexpr uint32_t NE = G::NHI - G::NLO + 1;
        const uint64_t i = (uint64_t)(blockIdx.x - gp) * 256 + threadIdx.x;
        const uint64_t l = i / NE;
        if (l >= L) return;
        const uint32_t n = G::NLO + (uint32_t)(i % NE);
        const double2 p = ab[l];
        tabc[(l / G::BLU) * ELEMS + (l % G::BLU) * G::LROW + G::NCODE + (n - G::NLO)] = p.x >= 0.0 ? ov_expected_rec(p.x, p.y, n) : 0.0;
    }
}

// o[row] += the tier-2 tile pass' partial sums of the row, in group order (the overflow kernels of the side stream wrote o before)
template <bool EXPECTED>
__global__ __launch_bounds__(256) void k_t2_tile_add(uint64_t n_rows, uint32_t groups, uint64_t npad, const double *__restrict__ part_ll,
                                                     const double *__restrict__ part_ell, double *__restrict__ o_ll,
                                                     double *__restrict__ o_ell)
{
    const uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= n_rows) return;
    double s = 0.0, e = 0.0;
    for (uint32_t g = 0; g < groups; g++) {
        s += part_ll[(uint64_t)g * npad + row];
        if (EXPECTED) e += part_ell[(uint64_t)g * npad + row];
    }
    o_ll[row] += s;
    if (EXPECTED) o_ell[row] += e;
}

// the two lists, from the static pair histogram: a thread per locus.  COUNT: pairs / sectors of the locus; FILL: at the offsets
template <bool FILL>
__global__ __launch_bounds__(256) void k_t2_lists(uint64_t L, const uint32_t *__restrict__ hist_all2, uint64_t *__restrict__ np,
                                                  uint64_t *__restrict__ ns, uint32_t *__restrict__ plist, uint32_t *__restrict__ slist,
                                                  uint32_t *__restrict__ pmask /*FILL: bit c2 = the pair occurs at the locus*/)
{
    const uint64_t l = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= L) return;
    uint64_t kp = FILL ? np[l] : 0, ks = FILL ? ns[l] : 0;
    uint32_t secs = 0, pm = 0;
    for (uint32_t c2 = 0; c2 < (uint32_t)T2_NCODE; c2++) {
        if (hist_all2[l * T2_CSTRIDE + c2] == 0u) continue;
        const uint32_t n = 5u + (c2 >= 6u) + (c2 >= 13u) + (c2 >= 21u), r = c2 - t2_code(n, 0u);
        secs |= 1u << (t2_pos(n, r) >> 3);
        pm |= 1u << c2;
        if (FILL) plist[kp] = (uint32_t)(l << 5) | c2;
        kp++;
    }
    for (uint32_t sec = 0; sec < 6u; sec++) {
        if (!((secs >> sec) & 1u)) continue;
        if (FILL) slist[ks] = (uint32_t)(l << 3) | sec;
        ks++;
    }
    if (!FILL) { np[l] = kp; ns[l] = ks; }
    else pmask[l] = pm;
}

// static: how often every tier-2 pair occurs at every locus (all cells of the shard); a thread per overflow entry (by-locus order)
__global__ __launch_bounds__(256) void k_t2_hist(uint64_t n_ovf, const uint32_t *__restrict__ ovc_locus, const uint64_t *__restrict__ ovc_ent,
                                                 uint32_t *__restrict__ hist_all2)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ovf) return;
    const uint64_t en = ovc_ent[i];
    const uint32_t r = ENT_REF(en), n = ENT_ALT(en) + r;
    if (t2_total(n)) atomicAdd(&hist_all2[(uint64_t)ovc_locus[i] * T2_CSTRIDE + t2_code(n, r)], 1u);
}

// cell side: a thread per cell row of the 64-row ELLPACK copy (coalesced entry loads), four entries' lookups in flight;
// sums in the row's order (ascending locus): deterministic.  Entries of other totals (0, above 8) are skipped.
// One-wave blocks that stride over the 64-row groups: the launch's grid size sets how many gathers are in flight chip-wide.
// Beside the tile kernel a SMALL grid is the point — every lookup fetches a line out of a table far bigger than an L2, and
// 1.6e7 of them in a burst (2 GB at 5 TB/s) starve the tile kernel's stream for the burst's duration (+0.2 ms at 10^6 cells x 200k
// loci); spread over the tile kernel's run time the same lines cost it far less (launch_overflow_cell).
template <bool EXPECTED>
__global__ __launch_bounds__(64) void k_t2_cell(uint64_t n_rows, const uint64_t *__restrict__ ell_ptr, const uint64_t *__restrict__ ell,
                                                const double *__restrict__ tab2, double *__restrict__ o_ll, double *__restrict__ o_ell)
{
  const uint64_t n_grp = (n_rows + 63) >> 6;
  for (uint64_t grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
    const uint64_t row = grp * 64 + threadIdx.x;
    if (row >= n_rows) continue;
    double s = 0.0, e = 0.0;
    const uint64_t base = ell_ptr[grp] + (row & 63), end = ell_ptr[grp + 1];
    constexpr int U = 4;
    for (uint64_t i = base; i < end; i += (uint64_t)U * 64) {
        uint64_t en[U];
        double lp[U], ev[U];
#pragma unroll
        for (int u = 0; u < U; u++) en[u] = i + (uint64_t)u * 64 < end ? ell[i + (uint64_t)u * 64] : OVF_PAD;  // (wave-uniform bound)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t r = ENT_REF(en[u]), n = ENT_ALT(en[u]) + r;
            const bool ok = en[u] != OVF_PAD && t2_total(n);
#ifdef T2_ABL  /* ablation: every lookup falls into the first 1024 loci's rows (384 KB: L2 hits; wrong values) */
            const uint64_t at = (uint64_t)(ENT_IDX(en[u]) & 1023u) * T2_ROW + t2_pos(ok ? n : T2_NMIN, ok ? r : 0u);
#else
            const uint64_t at = (uint64_t)ENT_IDX(en[u]) * T2_ROW + t2_pos(ok ? n : T2_NMIN, ok ? r : 0u);
#endif
            lp[u] = ok ? tab2[at] : 0.0;
            if (EXPECTED) ev[u] = ok ? tab2[at & ~7ull] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            s += lp[u];
            if (EXPECTED) e += ev[u];
        }
    }
    o_ll[row] = s;
    if (EXPECTED) o_ell[row] = e;
  }
}

// locus side: the tier-2 entries of the cells of the new exclusion set, counted per (locus, pair).  16 lanes per excluded
// cell walk its overflow row (~16 entries: one 128-byte line); ~10^6 scattered integer atomics per iteration at 10^6 cells.
// k_locus_finalize reads the counters and clears them again.
__global__ __launch_bounds__(256) void k_t2_minority(const uint32_t *__restrict__ n_min_p, const uint32_t *__restrict__ minlist,
                                                     const uint64_t *__restrict__ ovf_ptr, const uint64_t *__restrict__ ovf_ent,
                                                     uint32_t *__restrict__ cnt2)
{
    const uint32_t n_min = *n_min_p;
    const uint32_t j = threadIdx.x % LF_LANES;
    const uint32_t ng = gridDim.x * (256 / LF_LANES);
    for (uint32_t k = (blockIdx.x * 256 + threadIdx.x) / LF_LANES; k < n_min; k += ng) {
        const uint32_t cell = minlist[k];
        for (uint64_t i = ovf_ptr[cell] + j, end = ovf_ptr[cell + 1]; i < end; i += LF_LANES) {
            const uint64_t en = ovf_ent[i];
            const uint32_t r = ENT_REF(en), n = ENT_ALT(en) + r;
            if (t2_total(n)) atomicAdd(&cnt2[(uint64_t)ENT_IDX(en) * T2_CSTRIDE + t2_code(n, r)], 1u);
        }
    }
}

// the log-pmfs of the overflow entries OUTSIDE tier 2 (totals 0 and above 8: 1 % of the overflow entries) in by-locus order for
// the locus finalize: a thread per entry, the tier lists' arithmetic (the cell side adds the same bits)
__global__ __launch_bounds__(256) void k_ovx_values(uint64_t n_ovx, const uint32_t *__restrict__ ovx_locus, const uint64_t *__restrict__ ovx_ent,
                                                    const double2 *__restrict__ ab, const double *__restrict__ lf, double *__restrict__ lp_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ovx) return;
    const uint64_t en = ovx_ent[i];
    const double2 p = ab[ovx_locus[i]];
    const uint32_t a = ENT_ALT(en), r = ENT_REF(en), n = a + r;
    double lp = 0.0;  // a masked locus has no PMFData (main.rs:556); a 0/0 entry is exactly zero (quirk Q14)
    if (p.x >= 0.0 && n != 0u)
        lp = n <= (uint32_t)OV_NE ? (lf[n] - lf[a] - lf[r]) + dm_log_beta_ratio(p.x, p.y, a, r) : ov_slow_log_pmf(lf, p.x, p.y, a, r);
    lp_out[i] = lp;
}

// 64-row ELLPACK copy of the overflow CSR.  COUNT: slots of group g = 64 x its longest row; FILL: lane = row & 63 writes
// its entries at ell_ptr[g] + k * 64 + lane and pads.  One wave per group.
template <bool FILL>
__global__ __launch_bounds__(256) void k_ovf_ell_build(uint64_t n_rows, const uint64_t *__restrict__ ovf_ptr,
                                                       const uint64_t *__restrict__ ovf_ent, uint64_t *__restrict__ ell_ptr,
                                                       uint64_t *__restrict__ ell)
{
    const int lane = threadIdx.x & 63;
    const uint64_t grp = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint64_t n_grp = (n_rows + 63) >> 6;
    if (grp >= n_grp) return;
    const uint64_t row = grp * 64 + lane;
    uint64_t beg = 0, len = 0;
    if (row < n_rows) { beg = ovf_ptr[row]; len = ovf_ptr[row + 1] - beg; }
    uint64_t kmax = len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, (uint64_t)__shfl_xor((long long)kmax, off, 64));
    if (!FILL) {
        if (lane == 0) ell_ptr[grp] = kmax * 64;
        return;
    }
    const uint64_t base = ell_ptr[grp] + lane;
    for (uint64_t k = 0; k < kmax; k++) ell[base + k * 64] = k < len ? ovf_ent[beg + k] : OVF_PAD;
}

// The overflow entries the fast kernel leaves out, as two small lists in row order (static): tier 0 = totals 9..OV_NE (longer
// products; expected term still tabulated), tier 1 = totals above OV_NE (generic paths).  COUNT: entries per row and tier;
// FILL: (row, entry) pairs at the row's offset.  A thread per row.
__device__ __forceinline__ int ovf_tier(uint64_t en)
{
    const uint32_t n = ENT_ALT(en) + ENT_REF(en);
    return n <= (uint32_t)OV_FAST_N ? -1 : (n <= (uint32_t)OV_NE ? 0 : 1);
}
template <bool FILL>
__global__ __launch_bounds__(256) void k_ovf_tier_lists(uint64_t n_rows, const uint64_t *__restrict__ ovf_ptr,
                                                        const uint64_t *__restrict__ ovf_ent,
                                                        uint64_t *__restrict__ cnt0 /*count: out; fill: offsets*/,
                                                        uint64_t *__restrict__ cnt1, uint32_t *__restrict__ row0,
                                                        uint64_t *__restrict__ ent0, uint32_t *__restrict__ row1,
                                                        uint64_t *__restrict__ ent1)
{
    const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    uint64_t k0 = FILL ? cnt0[row] : 0, k1 = FILL ? cnt1[row] : 0;
    for (uint64_t i = ovf_ptr[row], end = ovf_ptr[row + 1]; i < end; i++) {
        const uint64_t en = ovf_ent[i];
        const int t = ovf_tier(en);
        if (t == 0) {
            if (FILL) { row0[k0] = (uint32_t)row; ent0[k0] = en; }
            k0++;
        } else if (t == 1) {
            if (FILL) { row1[k1] = (uint32_t)row; ent1[k1] = en; }
            k1++;
        }
    }
    if (!FILL) { cnt0[row] = k0; cnt1[row] = k1; }
}
// Their log-pmfs and expected terms, added to the rows' sums: a thread per listed entry; the thread of a row's FIRST listed
// entry walks the row's run (entries of a row are adjacent, in ascending-locus order: deterministic) and updates the row.
// GENERIC = false (tier 0): call-free and light enough to run beside the tile kernel; true (tier 1): dm_* generic paths.
template <bool EXPECTED, bool GENERIC>
__global__ __launch_bounds__(256) void k_ovf_cell_listed(uint64_t n_list, const uint32_t *__restrict__ rows,
                                                         const uint64_t *__restrict__ ents, const double2 *__restrict__ ab,
                                                         const double *__restrict__ lf, const double *__restrict__ otab,
                                                         double *__restrict__ o_ll, double *__restrict__ o_ell)
{
    const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 >= n_list) return;
    const uint32_t row = rows[i0];
    if (i0 > 0 && rows[i0 - 1] == row) return;
    double s = 0.0, e = 0.0;
    for (uint64_t i = i0; i < n_list && rows[i] == row; i++) {
        const uint64_t en = ents[i];
        const uint32_t l = ENT_IDX(en), a = ENT_ALT(en), r = ENT_REF(en), n = a + r;
        const double2 p = ab[l];
        if (!(p.x >= 0.0)) continue;  // masked locus: no PMFData (main.rs:556)
        if (GENERIC) {
            s += dm_log_bb_pmf(lf, p.x, p.y, a, r);
            if (EXPECTED) e += dm_expected_log_pmf(lf, p.x, p.y, n);
        } else {
            s += (lf[n] - lf[a] - lf[r]) + dm_log_beta_ratio(p.x, p.y, a, r);
            if (EXPECTED) e += otab[(uint64_t)l * OV_ROW + OV_EOFF + (n - 4)];
        }
    }
    o_ll[row] += s;
    if (EXPECTED) o_ell[row] += e;
}

// ln sum_k pmf(k)^2 (stats.rs:8-22) for any n, in O(n): the pmfs relative to the one next to the mode, k* = round(n alpha / (alpha +
// beta)), by the ratio recurrence upwards and downwards from there (terms far from the mode underflow to zero harmlessly, none
// overflows: pmf(k) / pmf(k*) stays near or below one), then ln(sum) + 2 ln pmf(k*).  The reference folds all n + 1 log-pmfs in
// log space (O(n^2) the way a thread has to evaluate them); this differs from it by rounding only — engine 1 keeps the fold, and
// every parity test runs both engines.
__device__ __noinline__ double ov_expected_mode(const double *lf, double alpha, double beta, uint32_t n)
{
    const double fn = (double)n;
    uint32_t ks = (uint32_t)(fn * alpha * ov_rcp(alpha + beta) + 0.5);
    if (ks > n) ks = n;
    double s = 1.0, t = 1.0;
    for (uint32_t k = ks; k < n; ++k) {  // pmf(k+1) / pmf(k) = (n-k)(alpha+k) / ((k+1)(beta+n-k-1))
        t *= ((double)(n - k) * (alpha + (double)k)) * ov_rcp((double)(k + 1) * (beta + (double)(n - k - 1)));
        s += t * t;
    }
    t = 1.0;
    for (uint32_t k = ks; k > 0; --k) {  // pmf(k-1) / pmf(k) = k (beta+n-k) / ((n-k+1)(alpha+k-1))
        t *= ((double)k * (beta + (double)(n - k))) * ov_rcp((double)(n - k + 1) * (alpha + (double)(k - 1)));
        s += t * t;
    }
    return log(s) + 2.0 * dm_log_bb_pmf(lf, alpha, beta, ks, n - ks);
}

// Totals above OV_NE (tier 1: a few entries in 10^4 even with deep coverage): chunked products for the log-pmf, the mode-anchored
// recurrence above for the expected term — a thread per listed entry, at most 64 VGPRs so that a wave fits beside the tile kernel's
// four per SIMD.  (Until round 3 the expected term was the reference's log-space fold, 16 lanes per entry.)  The values go to two
// small arrays; k_ovf_listed_add walks each row's run and adds them in list order.
template <bool EXPECTED>
__global__ __launch_bounds__(256, 8) void k_ovf_listed_values(uint64_t n_list, const uint64_t *__restrict__ ents,
                                                              const double2 *__restrict__ ab, const double *__restrict__ lf,
                                                              double *__restrict__ v_lp, double *__restrict__ v_e)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_list) return;
    const uint64_t en = ents[i];
    const uint32_t l = ENT_IDX(en), a = ENT_ALT(en), r = ENT_REF(en), n = a + r;
    const double2 p = ab[l];
    const bool live = p.x >= 0.0;  // else a masked locus: no PMFData (main.rs:556)
    v_lp[i] = live ? dm_log_bb_pmf(lf, p.x, p.y, a, r) : 0.0;
    if (EXPECTED) v_e[i] = live ? ov_expected_mode(lf, p.x, p.y, n) : 0.0;
}
template <bool EXPECTED>
__global__ __launch_bounds__(256) void k_ovf_listed_add(uint64_t n_list, const uint32_t *__restrict__ rows, const double *__restrict__ v_lp,
                                                        const double *__restrict__ v_e, double *__restrict__ o_ll, double *__restrict__ o_ell)
{
    const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 >= n_list) return;
    const uint32_t row = rows[i0];
    if (i0 > 0 && rows[i0 - 1] == row) return;
    double s = 0.0, e = 0.0;
    for (uint64_t i = i0; i < n_list && rows[i] == row; i++) {
        s += v_lp[i];
        if (EXPECTED) e += v_e[i];
    }
    o_ll[row] += s;
    if (EXPECTED) o_ell[row] += e;
}

// chunk-group partials in group order, then the row's overflow sum, then the normalisation (main.rs:314-323).
// Two rows per thread: 16-byte loads and stores (8-byte ones left this 190 MB stream at 1.8 TB/s).
template <bool EXPECTED>
__global__ __launch_bounds__(256) void k_cell_finalize(uint64_t n_rows, uint32_t groups, uint64_t npad,
                                                       const double *__restrict__ part_ll,
                                                       const double *__restrict__ part_ell,
                                                       const double *__restrict__ o_ll /*null: no overflow entries*/,
                                                       const double *__restrict__ o_ell,
                                                       const uint64_t *__restrict__ csr_ptr,
                                                       const uint32_t *__restrict__ masked_cnt, double *__restrict__ ll,
                                                       double *__restrict__ ell, double *__restrict__ nloci,
                                                       double *__restrict__ norm_out)
{
    const uint64_t row = 2 * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (row >= n_rows) return;
    const bool two = row + 1 < n_rows;  // (npad is even and the arrays are 16-byte aligned: a pair never straddles)
    double2 s = make_double2(0.0, 0.0), e = make_double2(0.0, 0.0);
    for (uint32_t g0 = 0; g0 < groups; g0 += 8) {  // eight groups' partials in flight, added in group order
        double2 pl[8], pe[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const bool ok = g0 + u < groups;
            pl[u] = ok ? *reinterpret_cast<const double2 *>(part_ll + (uint64_t)(g0 + u) * npad + row) : make_double2(0.0, 0.0);
            if (EXPECTED)
                pe[u] = ok ? *reinterpret_cast<const double2 *>(part_ell + (uint64_t)(g0 + u) * npad + row) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            if (g0 + u < groups) {
                s.x += pl[u].x; s.y += pl[u].y;
                if (EXPECTED) { e.x += pe[u].x; e.y += pe[u].y; }
            }
        }
    }
    if (o_ll) {
        s.x += o_ll[row];
        if (two) s.y += o_ll[row + 1];
        if (EXPECTED) {
            e.x += o_ell[row];
            if (two) e.y += o_ell[row + 1];
        }
    }
    const uint64_t p0 = csr_ptr[row], p1 = csr_ptr[row + 1], p2 = two ? csr_ptr[row + 2] : p1;
    const double c0 = (double)((p1 - p0) - (uint64_t)masked_cnt[row]);
    const double c1 = two ? (double)((p2 - p1) - (uint64_t)masked_cnt[row + 1]) : 0.0;
    const double n0 = c0 > 0.0 ? s.x / c0 : 0.0, n1 = c1 > 0.0 ? s.y / c1 : 0.0;  // main.rs:315-322
    if (two) {
        *reinterpret_cast<double2 *>(ll + row) = s;
        if (EXPECTED) *reinterpret_cast<double2 *>(ell + row) = e;
        *reinterpret_cast<double2 *>(nloci + row) = make_double2(c0, c1);
    } else {
        ll[row] = s.x; nloci[row] = c0;
        if (EXPECTED) ell[row] = e.x;
        if (two) { ll[row + 1] = s.y; nloci[row + 1] = c1; if (EXPECTED) ell[row + 1] = e.y; }
    }
    if (norm_out) {  // a slice of the exchange buffer: starts at the shard's first cell, any parity
        norm_out[row] = n0;
        if (two) norm_out[row + 1] = n1;
    }
}

// ---------------------------------------------------------------------------------------------------------
// locus pass over the compact CSC.  Two entry widths: 24 bits (cell 20 | code 4; four entries = three dwords) when
// the shard has at most 2^20 cells, else 32 bits (cell 28 | code 4).  The pass is a pure HBM stream, so bytes matter.
// ---------------------------------------------------------------------------------------------------------
template <int EB>
__device__ __forceinline__ void c4_read1(const uint32_t *__restrict__ base, uint64_t i, uint32_t *cell, uint32_t *code)
{
    if (EB == 32) {
        const uint32_t x = base[i];
        *cell = x & 0x0fffffffu;
        *code = x >> 28;
    } else {
        const uint8_t *b = reinterpret_cast<const uint8_t *>(base) + 3 * i;
        const uint32_t v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
        *cell = v & 0xfffffu;
        *code = v >> 20;
    }
}
template <int EB>
__device__ __forceinline__ void c4_write1(uint32_t *__restrict__ base, uint64_t i, uint32_t cell, uint32_t code)
{
    if (EB == 32) {
        base[i] = cell | (code << 28);
    } else {
        uint8_t *b = reinterpret_cast<uint8_t *>(base) + 3 * i;
        const uint32_t v = cell | (code << 20);
        b[0] = (uint8_t)v; b[1] = (uint8_t)(v >> 8); b[2] = (uint8_t)(v >> 16);
    }
}
// Form of the locus pass, decided on the device from this shard's exclusion-set size (see k_minority_hist below)
#define LM_NUM 1  // minority-driven when n_min / nloc <= LM_NUM / LM_DEN
#define LM_DEN 8
// (n_sub = partial planes of the minority-driven form: its 16-bit LDS counters hold a subset of at most 32767 cells — half
// the range, so that even a file that lists every (locus, cell) pair twice cannot carry into the neighbouring counter)
__device__ __forceinline__ bool locus_by_minority(int mode, uint32_t n_min, uint64_t nloc, uint32_t n_sub)
{
    if ((uint64_t)n_min > (uint64_t)n_sub * 32767u) return false;
    return mode == 2 || (mode == 0 && (uint64_t)n_min * LM_DEN <= nloc * LM_NUM);
}

struct __attribute__((packed, aligned(4))) ls_u3 { uint32_t x, y, z; };  // 12-byte load at a 4-byte aligned address
#define LS_THREADS 1024
template <bool BITS_IN_LDS, int EB>
__global__ __launch_bounds__(LS_THREADS) void k_locus_stats2(uint64_t L, uint32_t nbits_words,
                                                             const uint64_t *__restrict__ c4_ptr,
                                                             const uint32_t *__restrict__ c4_ent,
                                                             const uint32_t *__restrict__ flag_bits,
                                                             uint32_t *__restrict__ hist_min /*plane 0*/, int locus_mode,
                                                             uint64_t nloc, const uint32_t *__restrict__ n_min, uint32_t n_sub)
{
    if (locus_by_minority(locus_mode, *n_min, nloc, n_sub)) return;  // k_minority_ranges counts this iteration
    extern __shared__ uint32_t s_bits[];
    __shared__ uint32_t s_whist[LS_THREADS / 64][16];
    if (threadIdx.x < (LS_THREADS / 64) * 16) (&s_whist[0][0])[threadIdx.x] = 0;
    if (BITS_IN_LDS)
        for (uint32_t i = threadIdx.x; i < nbits_words; i += LS_THREADS) s_bits[i] = flag_bits[i];
    __syncthreads();
    uint32_t *whist = s_whist[threadIdx.x >> 6];
    const uint32_t *bits = BITS_IN_LDS ? s_bits : flag_bits;
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * (LS_THREADS / 64) + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * (LS_THREADS / 64);
    // the column pointers of a locus are requested one locus ahead, its per-code constants at the top of its turn: a
    // short column (a small shard) then costs one memory latency, not three in a row
    uint64_t n_c4b = 0, n_c4e = 0;
    if (wave0 < L) { n_c4b = c4_ptr[wave0]; n_c4e = c4_ptr[wave0 + 1]; }
    for (uint64_t l = wave0; l < L; l += nwaves) {
        // columns are padded to groups of four entries (code 15 = no entry): 16 bytes at 32 bits, 12 bytes at 24 bits
        const uint64_t vbeg = n_c4b >> 2, nvec = (n_c4e >> 2) - vbeg;
        const uint32_t *wp = c4_ent + vbeg * (EB == 32 ? 4 : 3);
        {
            const uint64_t ln = min(l + nwaves, L - 1);  // clamped: the last round re-reads a valid locus
            n_c4b = c4_ptr[ln]; n_c4e = c4_ptr[ln + 1];
        }
        // minority entries are few: they vote into this wave's 16-bin LDS histogram (integer atomics: exact).  Two
        // register buffers of four vectors (16 entries) per lane: one is processed while the other one loads.
        struct raw_t { uint32_t w[4][EB == 32 ? 4 : 3]; uint32_t in; };  // in: bit u = vector u lies inside the column
        // (the macros keep the buffers in named registers: no indexed local arrays).  The loads are unconditional (index
        // clamped) and nothing touches the loaded registers before LS_PROCESS: they stay in flight meanwhile.
#define LS_LOAD(R, I0)                                                                                           \
        (R).in = 0;                                                                                              \
        _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                          \
            const uint64_t i__ = (I0) + (uint64_t)u * 64 + lane;                                                 \
            const bool in__ = i__ < nvec;                                                                        \
            (R).in |= in__ ? (1u << u) : 0u;                                                                     \
            const uint32_t *q__ = wp + (in__ ? i__ : 0) * (EB == 32 ? 4 : 3);                                    \
            if (EB == 32) {                                                                                      \
                const uint4 v__ = *reinterpret_cast<const uint4 *>(q__);                                         \
                (R).w[u][0] = v__.x; (R).w[u][1] = v__.y; (R).w[u][2] = v__.z; (R).w[u][EB == 32 ? 3 : 0] = v__.w; \
            } else {                                                                                             \
                const ls_u3 v__ = *reinterpret_cast<const ls_u3 *>(q__);                                         \
                (R).w[u][0] = v__.x; (R).w[u][1] = v__.y; (R).w[u][2] = v__.z;                                   \
            }                                                                                                    \
        }
        // unpack 16 (cell, code) pairs, fetch their 16 bitmask words back to back, then vote.  Padding (and a vector
        // beyond the column: all ones) is cell = all ones, code 15: with the bitmask in LDS its word is inside the
        // allocation (sized for the largest cell index of the entry width) and bin 15 of the histogram is never read.
#define LS_PROCESS(R)                                                                                            \
        do {                                                                                                     \
            uint32_t cell__[16], code__[16], word__[16];                                                         \
            _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                      \
                const bool in__ = (((R).in >> u) & 1u) != 0;                                                     \
                if (EB == 32) {                                                                                  \
                    _Pragma("unroll") for (int q = 0; q < 4; q++) {                                              \
                        const uint32_t x__ = in__ ? (R).w[u][EB == 32 ? q : 0] : ~0u;                            \
                        cell__[4 * u + q] = x__ & 0x0fffffffu; code__[4 * u + q] = x__ >> 28;                    \
                    }                                                                                            \
                } else {                                                                                         \
                    const uint32_t w0 = in__ ? (R).w[u][0] : ~0u, w1 = in__ ? (R).w[u][1] : ~0u,                 \
                                   w2 = in__ ? (R).w[u][2] : ~0u;                                                \
                    const uint32_t es__[4] = {w0 & 0xffffffu, (w0 >> 24) | ((w1 & 0xffffu) << 8),               \
                                              (w1 >> 16) | ((w2 & 0xffu) << 16), w2 >> 8};                       \
                    _Pragma("unroll") for (int q = 0; q < 4; q++) {                                              \
                        cell__[4 * u + q] = es__[q] & 0xfffffu; code__[4 * u + q] = es__[q] >> 20;               \
                    }                                                                                            \
                }                                                                                                \
            }                                                                                                    \
            _Pragma("unroll") for (int e = 0; e < 16; e++) {                                                     \
                if (!(BITS_IN_LDS && EB == 24) && code__[e] >= (uint32_t)T_NCODE) cell__[e] = 0u; /* stay in range */ \
                word__[e] = bits[cell__[e] >> 5];                                                                \
            }                                                                                                    \
            _Pragma("unroll") for (int e = 0; e < 16; e++)                                                       \
                if ((word__[e] >> (cell__[e] & 31)) & 1u) atomicAdd(&whist[code__[e]], 1u);                      \
        } while (0)

        if (nvec) {
            raw_t ra, rb;
            LS_LOAD(ra, 0);
            for (uint64_t i0 = 0; i0 < nvec; i0 += 2 * 4 * 64) {
                LS_LOAD(rb, i0 + 4 * 64);
                LS_PROCESS(ra);
                LS_LOAD(ra, i0 + 2 * 4 * 64);
                LS_PROCESS(rb);
            }
        }
#undef LS_PROCESS
#undef LS_LOAD
        // the wave's bins are this locus' minority counts per code: plane 0 of hist_min (same wave wrote the bins: LDS
        // operations of one wave complete in order); k_locus_finalize turns them into the pass' outputs
        if (lane < 16) {
            hist_min[l * 16 + lane] = whist[lane];
            whist[lane] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// locus pass, minority-driven form.  Everything the pass reports follows from the per-(locus, code) counts of the
// MINORITY cells' regular entries (the majority side is the static histogram minus that), and the exclusion set is a
// small part of the cells (below the lower quartile by construction, a few percent in practice).  So instead of
// streaming the whole compact CSC past the exclusion bitmask, walk only the by-cell CSR rows of the excluded cells.
// Global atomics are out (measured: 2.7e10 scattered u32 atomics/s on this chip, 3.7 ms for cfg4's 1e8 entries), so
// the counting happens in LDS: the loci are cut into ranges of LR_LOCI, a workgroup owns one range and one of n_sub
// subsets of the excluded cells, finds every such cell's entries of its range through a per-(cell, range) offset
// table built at ingest (roff), counts them into a u32 histogram in LDS (integer atomics: exact, order independent)
// and writes the histogram out as one of n_sub partial planes; k_locus_finalize adds the planes.  The result is
// bit-identical to the streamed form.  Cost: the excluded cells' entries instead of all of them.  The form is chosen
// on the device from this shard's exclusion-set size (no host round trip); the kernels of the other form return at once.
// ---------------------------------------------------------------------------------------------------------
// A (cell, range) segment is a short run inside a long row and memory comes in 128-byte lines, so short segments waste
// most of what they fetch (measured at 1024 loci per range: 80-byte segments, 2 GB fetched for 0.8 GB of entries, the
// kernel at the HBM rate).  Hence wide ranges, with 16-bit counters so that the histogram still fits in LDS.
#define LR_LOCI 4096     // loci per range: the LDS histogram is 14 codes x LR_LOCI x u16 = 112 KB
#define LR_SUB_MAX 16    // at most this many subsets of the exclusion set (partial planes); chosen per matrix
#define LR_THREADS 1024  // one workgroup per CU, 128 VGPRs: 16 entry loads per lane stay in flight
#define LR_GROUP 64      // lanes per (cell, range) segment: ~41 entries at 1 % density (a tail loop would serialise)
#define LR_ROW (LR_LOCI + 2)  // u16 counters per code row (even: a row starts on a word)

// compact by-cell entries for k_minority_ranges: locus | code << 28 (code 15: an overflow entry, not counted there) —
// half the bytes of the packed CSR entry, and that kernel runs at the memory rate
__global__ __launch_bounds__(256) void k_cell_compact(uint64_t nnz, const uint64_t *__restrict__ csr_ent, uint16_t *__restrict__ c4r)
{
    static_assert(LR_LOCI == 4096, "12 bits of locus inside its range + 4 bits of code");
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nnz) return;
    const uint64_t e = csr_ent[i];
    c4r[i] = (uint16_t)((ENT_IDX(e) % LR_LOCI) | ((ent_regular(e) ? ent_code(e) : 15u) << 12));
}

// roff[cell][r] = number of the row's entries with locus < r * LR_LOCI, r = 0..R (row sorted by locus).  Wave per row.
__global__ __launch_bounds__(256) void k_range_offsets(uint64_t n_rows, uint32_t R, uint32_t width,
                                                       const uint64_t *__restrict__ csr_ptr,
                                                       const uint64_t *__restrict__ csr_ent, uint32_t *__restrict__ roff)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t row = wave0; row < n_rows; row += nwaves) {
        const uint64_t beg = csr_ptr[row], end = csr_ptr[row + 1];
        uint32_t *o = roff + row * (R + 1);
        // position i (0..len): ranges above that of entry i-1 up to that of entry i start at i (entry -1: range -1,
        // entry len: range R)
        for (uint64_t i0 = beg; i0 <= end; i0 += 64) {
            const uint64_t i = i0 + lane;
            if (i > end) continue;
            const int r_prev = i > beg ? (int)(ENT_IDX(csr_ent[i - 1]) / width) : -1;
            const int r_here = i < end ? (int)(ENT_IDX(csr_ent[i]) / width) : (int)R;
            for (int r = r_prev + 1; r <= r_here; r++) o[r] = (uint32_t)(i - beg);
        }
    }
}

// The excluded cells' offset rows, gathered and transposed: mroff[r][k] for the k-th cell of minlist, mbeg[k] = start of its
// row.  k_minority_ranges then reads its range's offsets as contiguous runs instead of one 64-byte line per (cell, range)
// out of the big per-cell table (measured: those line fetches were a third of that kernel's traffic).
#define LT_CELLS 64
__global__ __launch_bounds__(256) void k_minority_offsets(int locus_mode, uint64_t nloc, uint32_t n_sub, uint32_t R, uint64_t mstride,
                                                          const uint32_t *__restrict__ n_min_p,
                                                          const uint32_t *__restrict__ minlist,
                                                          const uint64_t *__restrict__ csr_ptr,
                                                          const uint32_t *__restrict__ roff, uint32_t *__restrict__ mroff,
                                                          uint64_t *__restrict__ mbeg)
{
    const uint32_t n_min = *n_min_p;
    if (!locus_by_minority(locus_mode, n_min, nloc, n_sub)) return;
    const uint32_t k0 = blockIdx.x * LT_CELLS;
    if (k0 >= n_min) return;
    extern __shared__ uint32_t s_t[];  // [LT_CELLS][row], row odd: the column read below strides by it
    __shared__ uint32_t s_cell[LT_CELLS];
    const uint32_t row = (R + 1) | 1u, nk = min((uint32_t)LT_CELLS, n_min - k0);
    if (threadIdx.x < nk) {
        const uint32_t cell = minlist[k0 + threadIdx.x];
        s_cell[threadIdx.x] = cell;
        mbeg[k0 + threadIdx.x] = csr_ptr[cell];
    }
    __syncthreads();
    // every cell's offsets are one contiguous run of R + 1 words: all threads stride over the (cell, range) pairs, so
    // that many independent loads are in flight
    const uint32_t tot = nk * (R + 1);
#pragma unroll 8
    for (uint32_t i = threadIdx.x; i < tot; i += 256) {
        const uint32_t c = i / (R + 1), r = i - c * (R + 1);
        s_t[c * row + r] = roff[(uint64_t)s_cell[c] * (R + 1) + r];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (R + 1) * LT_CELLS; i += 256) {
        const uint32_t r = i / LT_CELLS, c = i % LT_CELLS;
        if (c < nk) mroff[(uint64_t)r * mstride + k0 + c] = s_t[c * row + r];
    }
}

__global__ __launch_bounds__(LR_THREADS) void k_minority_ranges(int locus_mode, uint64_t nloc, uint64_t L, uint32_t R,
                                                               uint32_t n_sub, uint64_t mstride,
                                                               const uint32_t *__restrict__ n_min_p,
                                                               const uint32_t *__restrict__ mroff,
                                                               const uint64_t *__restrict__ mbeg,
                                                               const uint16_t *__restrict__ c4r,
                                                               uint32_t *__restrict__ hist_min /*[n_sub][L][16] u16*/)
{
    const uint32_t n_min = *n_min_p;
    if (!locus_by_minority(locus_mode, n_min, nloc, n_sub)) return;
    // u16 counters, two per word, code-major: the bank of a counter follows the locus (spread out), not the code (most
    // entries are single reads: codes 0 and 1).  A subset has at most 32767 cells (locus_by_minority): no carry.
    __shared__ uint32_t s_hist[T_NCODE * LR_ROW / 2];
    __shared__ uint64_t s_beg[LR_THREADS];
    __shared__ uint32_t s_len[LR_THREADS];
    const uint32_t tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const uint32_t r = blockIdx.x % R, sub = blockIdx.x / R;
    const uint32_t per = (n_min + n_sub - 1) / n_sub;
    const uint32_t k0 = min(n_min, sub * per), k1 = min(n_min, k0 + per);
    const uint32_t l0 = r * LR_LOCI;
    for (uint32_t i = tid; i < T_NCODE * LR_ROW / 2; i += LR_THREADS) s_hist[i] = 0;
    const uint32_t grp = lane / LR_GROUP, gl = lane % LR_GROUP;
#define LR_COUNT(E)                                                                                              \
    do {                                                                                                         \
        if (((E) >> 12) < (uint32_t)T_NCODE) {                                                                   \
            const uint32_t idx__ = ((E) >> 12) * LR_ROW + ((E) & 0x0fffu);                                       \
            atomicAdd(&s_hist[idx__ >> 1], 1u << ((idx__ & 1u) * 16u));                                          \
        }                                                                                                        \
    } while (0)
    // one excluded cell per thread: where its entries of this range start, and how many there are (requested one batch
    // ahead, so the loads fly during the batch before)
    uint64_t p_b = 0;
    uint32_t p_n = 0;
#define LR_PREFETCH(KB)                                                                                          \
    do {                                                                                                         \
        const uint32_t k__ = (KB) + tid;                                                                         \
        p_b = 0; p_n = 0;                                                                                        \
        if (k__ < k1) {                                                                                          \
            const uint32_t o0__ = mroff[(uint64_t)r * mstride + k__], o1__ = mroff[(uint64_t)(r + 1) * mstride + k__]; \
            p_b = mbeg[k__] + o0__;                                                                              \
            p_n = o1__ - o0__;                                                                                   \
        }                                                                                                        \
    } while (0)
    LR_PREFETCH(k0);
    for (uint32_t kb = k0; kb < k1; kb += LR_THREADS) {
        s_beg[tid] = p_b;
        s_len[tid] = p_n;
        __syncthreads();  // (also: the histogram is zeroed)
        LR_PREFETCH(kb + LR_THREADS);
        // a wave takes 64 of the segments, LR_GROUP lanes per segment; per trip the first LR_GROUP entries of 32 of them
        // are requested back to back (16 loads per lane in flight), then counted; longer segments finish in a loop
        constexpr int SPW = 64 / LR_GROUP;  // segments per wave load
        constexpr int NQ = 16;
        for (uint32_t q0 = 0; q0 < 64 / SPW; q0 += NQ) {
            uint32_t e[NQ];
#pragma unroll
            for (int u = 0; u < NQ; u++) {
                const uint32_t seg = wv * 64 + (q0 + u) * SPW + grp;
                e[u] = gl < s_len[seg] ? (uint32_t)c4r[s_beg[seg] + gl] : 0xffffu;  // all ones: code 15, not counted
            }
#pragma unroll
            for (int u = 0; u < NQ; u++) LR_COUNT(e[u]);
            for (int u = 0; u < NQ; u++) {
                const uint32_t seg = wv * 64 + (q0 + u) * SPW + grp;
                const uint32_t sn = s_len[seg];
                const uint64_t sb = s_beg[seg];
                for (uint32_t j = gl + LR_GROUP; j < sn; j += LR_GROUP) {
                    const uint32_t x = (uint32_t)c4r[sb + j];
                    LR_COUNT(x);
                }
            }
        }
        __syncthreads();  // segments consumed before the next batch overwrites them
    }
#undef LR_PREFETCH
#undef LR_COUNT
    __syncthreads();
    // this subset's plane of the range, as [locus][16] u16 like the counters (codes 14, 15: zero): 32 bytes per locus
    // and plane for this kernel to write and k_locus_finalize to read
    const uint64_t nl = min((uint64_t)LR_LOCI, L - l0);
    uint32_t *dst = reinterpret_cast<uint32_t *>(reinterpret_cast<uint16_t *>(hist_min) + ((uint64_t)sub * L + l0) * 16);
    for (uint32_t i = tid; i < nl * 8; i += LR_THREADS) {
        const uint32_t lo = i >> 3, c0 = (i & 7u) * 2u;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) {
            const uint32_t code = c0 + h, idx = code * LR_ROW + lo;
            if (code < (uint32_t)T_NCODE) v |= ((s_hist[idx >> 1] >> ((idx & 1u) * 16u)) & 0xffffu) << (16u * h);
        }
        dst[i] = v;
    }
}

// Outputs of the locus pass from the minority counts (both forms end here, so they agree to the bit).  16 lanes per
// locus: lane j < 14 takes code j's count (sum of the planes), static histogram and log-pmf; the locus' overflow entries
// (alt+ref == 0 or > T_K; ~1 %, their log-pmfs evaluated from the locus' cumulative-log row) are walked 16 at a time; per-lane partial results
// are added by a 4-step butterfly over the 16 lanes (fixed shape: deterministic).
template <bool INLINE_OVF>  // the overflow entries' log-pmfs: evaluated here (deep coverage) or read from ovf_lp (k_ovf_values)
__global__ __launch_bounds__(256) void k_locus_finalize(uint64_t L, int locus_mode, uint64_t nloc, uint32_t n_sub,
                                                        const uint32_t *__restrict__ n_min_p,
                                                        const uint32_t *__restrict__ hist_min,
                                                        const uint32_t *__restrict__ flag_bits,
                                                        const uint32_t *__restrict__ hist_all,
                                                        const double *__restrict__ tab, uint32_t tab_stride,
                                                        const uint8_t *__restrict__ mask,
                                                        const uint64_t *__restrict__ ovc_ptr /*null: no overflow*/,
                                                        const uint64_t *__restrict__ ovc_ent,
                                                        const double *__restrict__ ovf_lp /*null: evaluate here*/,
                                                        const double *__restrict__ otab, const double *__restrict__ lf,
                                                        const double2 *__restrict__ ab, double *__restrict__ out,
                                                        uint32_t *__restrict__ cnt2 /*null: no tier 2*/,
                                                        const uint32_t *__restrict__ hist_all2, const uint32_t *__restrict__ pmask2,
                                                        const double *__restrict__ tab2,
                                                        uint8_t *__restrict__ mask_next /*null: the locus filter is a kernel of its own*/,
                                                        uint32_t *__restrict__ n_filtered)
{
    // the minority-driven form left n_sub planes of u16 counts, the streamed form one plane of u32 counts
    const bool by_min = locus_by_minority(locus_mode, *n_min_p, nloc, n_sub);
    const uint32_t j = threadIdx.x % LF_LANES;
    const uint64_t l_raw = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LF_LANES;
    const bool in = l_raw < L;
    const uint64_t l = in ? l_raw : L - 1;  // whole waves stay in the butterflies
    const bool live = mask[l] != 0;
    uint64_t obeg = 0, oend = 0;
    if (ovc_ptr) { obeg = ovc_ptr[l]; oend = ovc_ptr[l + 1]; }
    double cmin = 0.0, cmaj = 0.0;
    uint32_t nmin = 0;
    uint64_t amin = 0, rmin = 0;
    if (j < T_NCODE) {
        uint32_t cnt = 0;
        if (by_min) {
            const uint16_t *h16 = reinterpret_cast<const uint16_t *>(hist_min);
            for (uint32_t p = 0; p < n_sub; p++) cnt += h16[((uint64_t)p * L + l) * 16 + j];
        } else {
            cnt = hist_min[l * 16 + j];
        }
        const uint32_t h_all = hist_all[l * T_NCODE + j];
        // (element stride 2 when the table holds (log-pmf, expected) pairs)
        const double t_code = tab[(l / T_BLU) * TAB_ELEMS + (l % T_BLU) * T_LROW + j];
        amin = (uint64_t)cnt * T_A_OF[j];
        rmin = (uint64_t)cnt * T_R_OF[j];
        if (live) {
            nmin = cnt;
            cmin = (double)cnt * t_code;
            cmaj = (double)(h_all - cnt) * t_code;
        }
    }
    // tier 2 (the overflow entries with totals 5..8, counted per (locus, pair) by k_t2_minority): lane j takes the pairs j and
    // j + 16; count x table value, like the regular codes.  The counters are cleared for the next iteration.
    if (cnt2) {
        const uint32_t pm = pmask2[l];  // (the pairs that occur at this locus, static: 8 of the 30 on a 125k-cell shard)
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) {
            const uint32_t c2 = j + h * LF_LANES;
            if (!((pm >> c2) & 1u)) continue;  // the pair does not occur at this locus: no counts, no table slot
            const uint32_t h_all = hist_all2[l * T2_CSTRIDE + c2];
            const uint32_t cnt = cnt2[l * T2_CSTRIDE + c2];
            if (cnt && in) cnt2[l * T2_CSTRIDE + c2] = 0u;  // (the clamped lanes beyond L only read)
            const uint32_t n2 = 5u + (c2 >= 6u) + (c2 >= 13u) + (c2 >= 21u), r2 = c2 - t2_code(n2, 0u);
            const double t_code = tab2[l * T2_ROW + t2_pos(n2, r2)];
            amin += (uint64_t)cnt * (n2 - r2);
            rmin += (uint64_t)cnt * r2;
            if (live) {
                nmin += cnt;
                cmin += (double)cnt * t_code;
                cmaj += (double)(h_all - cnt) * t_code;
            }
        }
    }
    // overflow entries: four independent entry loads per lane in flight, then their exclusion-bitmask words and their
    // log-pmfs = lnC + LA[alt] + LB[ref] - LAB[n] out of the locus' cumulative-log row (k_ovf_tables; the 16 lanes share the
    // row: L1) — evaluated here instead of being stored by a kernel of their own and read back (12 + 8 + 8 bytes per entry)
    const double *orow = otab + l * OV_ROW;
    for (uint64_t i0 = obeg + j; i0 < oend; i0 += 4 * LF_LANES) {
        uint64_t en[4];
        double lp[4];
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t i = i0 + (uint64_t)u * LF_LANES;
            en[u] = i < oend ? ovc_ent[i] : ~0ull;
            lp[u] = (!INLINE_OVF && i < oend) ? ovf_lp[i] : 0.0;  // (stored by k_ovf_values: zero at masked loci)
        }
#pragma unroll
        for (int u = 0; u < 4; u++) w[u] = en[u] != ~0ull ? flag_bits[ENT_IDX(en[u]) >> 5] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!INLINE_OVF) break;
            lp[u] = 0.0;
            if (en[u] == ~0ull || !live) continue;  // (a masked locus has no PMFData, main.rs:556: only its tallies count)
            const uint32_t a = ENT_ALT(en[u]), r = ENT_REF(en[u]), n = a + r;
            if (n == 0) lp[u] = 0.0;  // quirk Q14: exactly zero
            else if (n < (uint32_t)OV_NT)  // (totals below 18: ln C straight out of the factorial table — no ln_gamma branch inlined here)
                lp[u] = (lf[n] - lf[a] - lf[r]) + (orow[a] + orow[OV_NT + r] - orow[2 * OV_NT + n]);
            else {
                const double2 p = ab[l];
                lp[u] = ov_slow_log_pmf(lf, p.x, p.y, a, r);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (en[u] == ~0ull) continue;
            const bool minority = ((w[u] >> (ENT_IDX(en[u]) & 31)) & 1u) != 0;
            if (minority) { amin += ENT_ALT(en[u]); rmin += ENT_REF(en[u]); }
            if (live) {
                if (minority) { cmin += lp[u]; nmin++; } else cmaj += lp[u];
            }
        }
    }
    cmin = group16_sum(cmin);
    cmaj = group16_sum(cmaj);
    nmin = group16_sum(nmin);
    amin = group16_sum(amin);
    rmin = group16_sum(rmin);
    if (in && j == 0) {
        out[LB_CONTRIB_MIN * L + l] = cmin;
        out[LB_CONTRIB_MAJ * L + l] = cmaj;
        out[LB_CELLS_MIN * L + l] = (double)nmin;
        out[LB_ALT_MIN * L + l] = (double)amin;
        out[LB_REF_MIN * L + l] = (double)rmin;
        // An unsharded ctx has the locus' final sums right here: the -80 filter of locus_filter_and_output_locus_data
        // (main.rs:428-451; k_locus_filter's arithmetic on the values just written) without a launch of its own.
        if (mask_next) {
            const double fn = (double)nmin;
            const double per_cell = fn != 0.0 ? cmin / fn : 0.0;
            uint8_t m = mask[l];
            if (per_cell < -80.0) {
                m = 0;
                atomicAdd(n_filtered, 1u);
            }
            mask_next[l] = m;
        }
    }
}

// entries of newly masked loci no longer count as used loci of their cells (main.rs:556,575)
template <int EB>
__global__ __launch_bounds__(256) void k_masked_update(uint64_t L, const uint8_t *__restrict__ mask_old,
                                                       const uint8_t *__restrict__ mask_new,
                                                       const uint64_t *__restrict__ c4_ptr,
                                                       const uint32_t *__restrict__ c4_ent,
                                                       const uint64_t *__restrict__ ovc_ptr,
                                                       const uint64_t *__restrict__ ovc_ent,
                                                       uint32_t *__restrict__ masked_cnt)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t l = wave0; l < L; l += nwaves) {
        if (!(mask_old[l] && !mask_new[l])) continue;
        for (uint64_t i = c4_ptr[l] + lane; i < c4_ptr[l + 1]; i += 64) {
            uint32_t cell, code;
            c4_read1<EB>(c4_ent, i, &cell, &code);
            if (code < (uint32_t)T_NCODE) atomicAdd(&masked_cnt[cell], 1u);
        }
        for (uint64_t i = ovc_ptr[l] + lane; i < ovc_ptr[l + 1]; i += 64) atomicAdd(&masked_cnt[ENT_IDX(ovc_ent[i])], 1u);
    }
}

// ---------------------------------------------------------------------------------------------------------
// posterior phase (calculate_posteriors, main.rs:228-280)
// ---------------------------------------------------------------------------------------------------------
__global__ void k_ab_posterior3(uint64_t L, const double *__restrict__ s_alt, const double *__restrict__ s_ref,
                                const double *__restrict__ alt_min, const double *__restrict__ ref_min, double mf0,
                                double2 *__restrict__ ab3, double *__restrict__ ab6)
{
    const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    double a_maj = (s_alt[l] + 1.0) - alt_min[l], b_maj = (s_ref[l] + 1.0) - ref_min[l];          // main.rs:239
    const double a_min = (s_alt[l] + 1.0) - (s_alt[l] - alt_min[l]);                                // main.rs:241
    const double b_min = (s_ref[l] + 1.0) - (s_ref[l] - ref_min[l]);
    const double a_dbl = (a_maj - 1.0) * mf0 + (a_min - 1.0) + 1.0;                                 // main.rs:245
    const double b_dbl = (b_maj - 1.0) * mf0 + (b_min - 1.0) + 1.0;
    const double mf = fmax(mf0, 0.01);                                                              // main.rs:250
    a_maj = (a_maj - 1.0) * mf + 1.0;                                                               // main.rs:252
    b_maj = (b_maj - 1.0) * mf + 1.0;
    ab3[l] = make_double2(a_min, b_min);
    ab3[L + l] = make_double2(a_maj, b_maj);
    ab3[2 * L + l] = make_double2(a_dbl, b_dbl);
    double *o = ab6 + 8 * l;
    o[0] = a_min; o[1] = b_min; o[2] = a_maj; o[3] = b_maj; o[4] = a_dbl; o[5] = b_dbl; o[6] = 0.0; o[7] = 0.0;
}

__global__ __launch_bounds__(256) void k_posterior_finalize(uint64_t n_rows, const double *__restrict__ osum /*[3][2][n_rows] or null*/,
                                                            uint32_t groups, uint64_t npad,
                                                            const double *__restrict__ part /*[3][2][G][npad]*/,
                                                            double lp_min, double lp_maj, double lp_dbl,
                                                            double *__restrict__ post)
{
    const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const uint64_t set_stride = 2ull * groups * npad;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (uint32_t g = 0; g < groups; g++) {
        s0 += part[(uint64_t)g * npad + row];
        s1 += part[set_stride + (uint64_t)g * npad + row];
        s2 += part[2 * set_stride + (uint64_t)g * npad + row];
    }
    if (osum) {
        s0 += osum[row];
        s1 += osum[2 * n_rows + row];
        s2 += osum[4 * n_rows + row];
    }
    const double s_min = s0, s_maj = s1, s_dbl = s2;
    const double log_num = lp_min + s_min;                       // main.rs:267
    double log_den = dm_logsumexp(log_num, lp_maj + s_maj);      // main.rs:268
    const double log_dbl = lp_dbl + s_dbl;                       // main.rs:270
    log_den = dm_logsumexp(log_den, log_dbl);                    // main.rs:271
    post[row] = exp(log_num - log_den);
    post[n_rows + row] = exp(log_dbl - log_den);
    post[2 * n_rows + row] = s_maj;
    post[3 * n_rows + row] = s_min;
}

// ---------------------------------------------------------------------------------------------------------
// build: tiles + overflow CSR from the by-cell CSR; compact CSC + overflow CSC + per-locus code histogram
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t row_lower_bound(const uint64_t *__restrict__ ent, uint64_t lo, uint64_t hi, uint32_t locus)
{
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (ENT_IDX(ent[mid]) < locus) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One 1024-thread block per tile, thread = cell of the block.  The cells are ordered by their number of regular entries
// in this chunk (stable counting sort: deterministic layout), every 64 of them form a slice of 64 rows [cell, K entries]
// with K = the slice's longest cell rounded up to odd.  FILL = false: tile size; FILL = true: write slices + header.
#define TB_BINS 64  // entry counts >= TB_BINS-1 share the last bin (they sort to the end, in cell order)
// which entries a tile set takes, and their 16-bit form inside chunk j
template <class G>
__device__ __forceinline__ bool geo_take(uint64_t e)
{
    const uint32_t n = ENT_ALT(e) + ENT_REF(e);
    return n - G::NLO <= G::NHI - G::NLO;
}
template <class G>
__device__ __forceinline__ uint16_t geo_encode(uint64_t e, uint32_t j)
{
    const uint32_t r = ENT_REF(e), n = ENT_ALT(e) + r;
    const uint32_t code = G::NLO == 1 ? ent_code(e) : t2_code(n, r);
    return (uint16_t)(((n - G::NLO) << 14) | ((ENT_IDX(e) - j * G::BLU) << G::SHIFT) | code);
}
template <bool FILL, class G = geo_reg>
__global__ __launch_bounds__(T_BC) void k_tile_build(uint64_t nloc, uint32_t nj, uint64_t tile0,
                                                     const uint64_t *__restrict__ csr_ptr,
                                                     const uint64_t *__restrict__ csr_ent,
                                                     uint64_t *__restrict__ tile_elems /*count pass: out; fill: tile_ptr*/,
                                                     uint16_t *__restrict__ tiles, uint16_t *__restrict__ thdr,
                                                     const uint32_t *__restrict__ toff /*[row][nj + 1] or null*/)
{
    __shared__ uint32_t s_cnt[T_BC / 64][TB_BINS];  // cells per (source wave, bin)
    __shared__ uint32_t s_base[TB_BINS];            // first rank of a bin
    __shared__ uint32_t s_kmax[T_BC / 64];          // longest cell of a slice
    __shared__ uint32_t s_sbase[T_BC / 64 + 1];     // first entry of a slice inside the tile
    const uint64_t t = tile0 + blockIdx.x;
    const uint32_t b = (uint32_t)(t / nj), j = (uint32_t)(t % nj);
    const uint32_t cl = threadIdx.x, lane = cl & 63, wv = cl >> 6;
    const uint64_t row = (uint64_t)b * T_BC + cl;
    for (uint32_t i = cl; i < (T_BC / 64) * TB_BINS; i += T_BC) (&s_cnt[0][0])[i] = 0;
    if (cl < T_BC / 64) s_kmax[cl] = 0;
    uint64_t lo = 0, hi = 0;
    uint32_t len = 0;
    if (row < nloc) {
        const uint64_t beg = csr_ptr[row], end = csr_ptr[row + 1];
        if (toff) {  // where the row's entries of every chunk start (k_range_offsets): two reads instead of two searches
            lo = beg + toff[row * (nj + 1) + j];
            hi = beg + toff[row * (nj + 1) + j + 1];
        } else {
            lo = row_lower_bound(csr_ent, beg, end, j * G::BLU);
            hi = row_lower_bound(csr_ent, lo, end, (j + 1u) * G::BLU);
        }
        for (uint64_t i = lo; i < hi; i++) len += geo_take<G>(csr_ent[i]) ? 1u : 0u;
    }
    __syncthreads();
    // stable counting sort by bin = min(len, TB_BINS-1): rank inside (wave, bin) from ballots
    const uint32_t bin = min(len, (uint32_t)TB_BINS - 1u);
    uint32_t within = 0;
    {
        unsigned long long todo = ~0ull;  // lanes whose bin has not been handled yet
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            const uint32_t v = (uint32_t)__shfl((int)bin, src, 64);
            const unsigned long long m = __ballot(bin == v);
            if (bin == v) within = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if ((int)lane == src) s_cnt[wv][v] = (uint32_t)__popcll(m);
            todo &= ~m;
        }
    }
    __syncthreads();
    if (cl < TB_BINS) {  // exclusive prefix over bins of the bin totals
        uint32_t tot = 0;
        for (int w = 0; w < T_BC / 64; w++) tot += s_cnt[w][cl];
        uint32_t inc = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if ((int)cl >= off) inc += o;
        }
        s_base[cl] = inc - tot;
    }
    __syncthreads();
    uint32_t rank = s_base[bin] + within;
    for (uint32_t w = 0; w < wv; w++) rank += s_cnt[w][bin];
    const uint32_t dw = rank >> 6, dl = rank & 63;  // destination slice and lane
    atomicMax(&s_kmax[dw], len);
    __syncthreads();
    if (cl == 0) {
        uint32_t acc = 0;
        for (int w = 0; w < T_BC / 64; w++) {
            s_sbase[w] = acc;
            acc += 64u * ((s_kmax[w] | 1u) + 1u);
        }
        s_sbase[T_BC / 64] = acc;
    }
    __syncthreads();
    if (!FILL) {
        if (cl == 0) tile_elems[t] = (uint64_t)s_sbase[T_BC / 64];  // a multiple of 128 u16
        return;
    }
    const uint64_t tbase = tile_elems[t];
    uint16_t *hp = thdr + t * T_HDR;
    if (cl < T_BC / 64) {
        uint32_t *hd = reinterpret_cast<uint32_t *>(hp) + 4 * cl;
        const uint64_t first = tbase + s_sbase[cl];
        hd[0] = (uint32_t)first;
        hd[1] = (uint32_t)(first >> 32);
        hd[2] = s_kmax[cl] | 1u;  // K: padded entries per cell of the slice (odd: a row is K + 1 u16)
        hd[3] = s_kmax[cl] == 0u;  // no row of the slice has an entry (the nearly empty tier-2 tiles of a deep matrix: most slices)
    }
    const uint32_t K = s_kmax[dw] | 1u;
    uint16_t *dst = tiles + tbase + s_sbase[dw] + dl * (K + 1u);  // this cell's row
    dst[0] = (uint16_t)cl;
    uint32_t k = 0;
    for (uint64_t i = lo; i < hi; i++) {
        const uint64_t e = csr_ent[i];
        if (geo_take<G>(e)) dst[1 + k++] = geo_encode<G>(e, j);
    }
    for (; k < K; k++) dst[1 + k] = (uint16_t)(G::BLU << G::SHIFT);  // padding: the chunk's all-zero slot
}

// Bank-aware order of the entries inside the rows of one slice (option "bank_order"; a wave per slice, lane = row, the slice's rows
// in LDS).  The tile kernel's lookup step k reads, for the 32 lanes of a half-wave, the log-pmf at slot * 18 + code and the
// expected term at slot * 18 + 14 + (n - 1): two 8-byte LDS reads per lane, served at one cycle per DISTINCT address on the busiest
// of the 32 bank pairs.  In file order the banks are random — 3.3 cycles per step and half-wave instead of 1 — and those conflicts
// are 42 % of the kernel's time (SQ_LDS_BANK_CONFLICT).  A row's sum does not care about the order of its entries beyond rounding,
// so the builder picks it: step by step, every lane whose entry of this step is still open proposes the cheapest of its remaining
// entries given the bank loads of the lanes already placed in the step; of the proposers that share a bank pair the lowest lane is
// placed, the others propose again; after four rounds whoever is left takes its proposal.  Simulated (tools/probe/bank_sim.py):
// 3.3 -> 2.4 cycles per step, the same as placing the lanes one after the other.  Deterministic; the order inside a row then depends
// on the 31 rows that share its half-wave, i.e. on the shard's cell set: per-cell sums of differently sharded runs differ in the
// last bits (as they already do between different chunk-group counts).
#define TBO_ROUNDS 4
__device__ __forceinline__ void tbo_banks(uint32_t e, uint32_t *a, uint32_t *b)
{
    const uint32_t base = ((e >> 4) & 1023u) * (uint32_t)T_LROW;
    *a = (base + (e & 15u)) & 31u;
    *b = (base + (uint32_t)T_NCODE + (e >> 14)) & 31u;
}
// Which lane of its slice a row takes (any permutation of a slice's 64 rows is a valid layout).  The tile kernel adds a row's
// sums to its cell's accumulator in LDS with one 16-byte read and one 16-byte write per lane: address = cell * 16, served in
// groups of 16 lanes (read: {0-3,12-15,20-27}, {4-11,16-19,28-31}, and the same + 32) resp. 8 contiguous lanes (write), one cycle
// per distinct address on a bank quad = cell mod 16.  Rows in count order carry arbitrary cells — 2.7 addresses on the busiest
// quad of a read group.  Here the slice's rows are ranked by (cell mod 16, lane) and dealt round-robin to the four read groups,
// the second and fourth group shifted by half a group so that two rows of one class never share a write group either: classes of
// up to four rows (the average) become conflict-free.  Returns the row count that now belongs to this lane.
__device__ uint32_t tile_lane_assign(uint16_t *slice /*64 rows of Kw + 1 u16*/, uint32_t Kw, uint32_t lane, uint32_t cnt, uint32_t *scr)
{
#define TLA_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
    const uint32_t cls = (uint32_t)slice[lane * (Kw + 1u)] & 15u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t t = 0;
#pragma unroll
    for (uint32_t c = 0; c < 16u; c++) {
        const unsigned long long m = __ballot(cls == c);
        if (c < cls) t += (uint32_t)__popcll(m);
        else if (c == cls) t += (uint32_t)__popcll(m & lt);
    }
    const uint32_t g = t & 3u, pp = ((t >> 2) + ((g & 1u) ? 8u : 0u)) & 15u;
    // lane number pp of read group g & 1 (G0 = 0-3, 12-15, 20-27; G1 = 4-11, 16-19, 28-31), upper half for g >= 2
    const uint32_t l0 = pp < 4u ? pp : (pp < 8u ? pp + 8u : pp + 12u);          // G0: 0..3 | 12..15 | 20..27
    const uint32_t l1 = pp < 8u ? pp + 4u : (pp < 12u ? pp + 8u : pp + 16u);    // G1: 4..11 | 16..19 | 28..31
    const uint32_t dst = ((g & 1u) ? l1 : l0) + ((g & 2u) ? 32u : 0u);
    scr[dst] = lane;
    scr[64 + dst] = cnt;
    TLA_SYNC();
    const uint32_t src = scr[lane], cnt_new = scr[64 + lane];
    for (uint32_t k = 0; k <= Kw; k++) {  // column by column: all of a column's reads before its writes
        const uint16_t v = slice[src * (Kw + 1u) + k];
        TLA_SYNC();
        slice[lane * (Kw + 1u) + k] = v;
        TLA_SYNC();
    }
    return cnt_new;
#undef TLA_SYNC
}

#define TBO_WIN 4  // candidates per lane and round: its next four remaining entries (the whole rest is no better: 2.46 vs 2.50 cycles)
__device__ void tile_bank_order(uint16_t *row /*this lane's K entries*/, uint32_t K, uint32_t cnt /*real entries: the first cnt*/,
                                uint32_t lane, uint32_t *scr /*[256]: this wave's bank loads and claims*/)
{
    const uint32_t h = lane >> 5;
    uint32_t *ld_a = scr + h * 64, *ld_b = ld_a + 32, *win_a = scr + 128 + h * 64, *win_b = win_a + 32;
    // (one wave: its LDS operations complete in program order; the asm statements only keep the compiler from moving them)
#define TBO_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
    if (K <= 1u) return;  // (wave-uniform) nothing to choose
    for (uint32_t k = 0; k + 1u < K; k++) {
        if (__ballot(k + 1u < cnt) == 0ull) break;  // no lane has two entries left to choose from
        scr[lane] = 0u; scr[64 + lane] = 0u;  // the step's loads
        // this lane's candidates: its next TBO_WIN remaining entries, and their bank pairs
        uint32_t e[TBO_WIN], ca[TBO_WIN], cb[TBO_WIN];
#pragma unroll
        for (uint32_t u = 0; u < (uint32_t)TBO_WIN; u++) {
            e[u] = k + u < cnt ? (uint32_t)row[k + u] : 0xffffffffu;
            tbo_banks(e[u], &ca[u], &cb[u]);
        }
        bool open = k < cnt;  // (a row out of real entries keeps its padding entry: one shared address)
        TBO_SYNC();
        for (uint32_t rd = 0; rd < (uint32_t)TBO_ROUNDS; rd++) {
            if (__ballot(open) == 0ull) break;
            scr[128 + lane] = ~0u; scr[192 + lane] = ~0u;  // the round's claims
            uint32_t cost[TBO_WIN];
#pragma unroll
            for (uint32_t u = 0; u < (uint32_t)TBO_WIN; u++) cost[u] = ld_a[ca[u]] + ld_b[cb[u]];
            uint32_t best = 0, bc = cost[0];
#pragma unroll
            for (uint32_t u = 1; u < (uint32_t)TBO_WIN; u++)
                if (e[u] != 0xffffffffu && cost[u] < bc) { bc = cost[u]; best = u; }
            uint32_t ba = ca[0], bb = cb[0];
#pragma unroll
            for (uint32_t u = 1; u < (uint32_t)TBO_WIN; u++)
                if (best == u) { ba = ca[u]; bb = cb[u]; }
            TBO_SYNC();
            if (open) {
                atomicMin(&win_a[ba], lane);
                atomicMin(&win_b[bb], lane);
            }
            TBO_SYNC();
            if (open && (rd == (uint32_t)TBO_ROUNDS - 1u || (win_a[ba] == lane && win_b[bb] == lane))) {
                if (best) {  // swap the chosen entry into position k
                    uint32_t eb = e[0];
#pragma unroll
                    for (uint32_t u = 1; u < (uint32_t)TBO_WIN; u++)
                        if (best == u) eb = e[u];
                    row[k] = (uint16_t)eb;
                    row[k + best] = (uint16_t)e[0];
                }
                atomicAdd(&ld_a[ba], 1u);
                atomicAdd(&ld_b[bb], 1u);
                open = false;
            }
            TBO_SYNC();
        }
    }
#undef TBO_SYNC
}

// The same tiles, built the way the memory system likes (used whenever the per-(cell, chunk) offsets table exists):
//   * a PERSISTENT workgroup takes whole cell blocks and walks a block's tiles chunk by chunk: a thread's reads of its row move
//     forward through one cache line after the other (a grid of one workgroup per tile spread the chunks of a block over the
//     XCDs, and every tile fetched its 1024 row segments afresh: 2.5x the bytes);
//   * it reads the 2-byte compact by-cell entries (c4r: locus mod 4096 | code << 12, code 15 = not a table entry) instead of the
//     8-byte packed ones — the chunk is narrower than 4096 loci, so the slot inside it follows from the low 12 bits;
//   * the slices are assembled in LDS and leave as whole 16-byte stores (rows written two bytes at a time straight to global
//     memory cost 65x their bytes in partial-line traffic: 0.35 TB for 5.3 GB of tiles at 10^6 cells x 200k loci).
// Same layout, bit for bit, as k_tile_build.
#define TB_STAGE (24 * 1024)  // u16 of a tile staged in LDS (48 KB: two workgroups per CU); a bigger tile is written directly (rows of hundreds of entries)
__device__ __constant__ uint8_t T_NM1_OF[16] = {0, 0, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 3, 0, 0};  // alt+ref-1 of a code
template <bool FILL, bool ORDER = false>
__global__ __launch_bounds__(T_BC, 8) void k_tile_build2(uint64_t nloc, uint32_t nb, uint32_t nj, const uint64_t *__restrict__ csr_ptr,
                                                      const uint16_t *__restrict__ c4r, const uint32_t *__restrict__ toff /*[row][nj + 1]*/,
                                                      uint64_t *__restrict__ tile_elems /*count pass: out; fill: tile_ptr*/,
                                                      uint16_t *__restrict__ tiles, uint16_t *__restrict__ thdr)
{
    static_assert(T_BLU < LR_LOCI, "a chunk's loci are told apart by their low 12 bits");
    __shared__ uint32_t s_cnt[T_BC / 64][TB_BINS];
    __shared__ uint32_t s_base[TB_BINS];
    __shared__ uint32_t s_kmax[T_BC / 64];
    __shared__ uint32_t s_sbase[T_BC / 64 + 1];
    __shared__ __attribute__((aligned(16))) uint16_t s_tile[FILL ? TB_STAGE : 8];
    __shared__ uint32_t s_scr[ORDER ? (T_BC / 64) * 256 : 1];  // bank-aware order: a wave's loads and claims
    __shared__ uint16_t s_rcnt[ORDER ? T_BC : 1];              // ... real entries of every row of the tile, by (slice, lane)
    const uint32_t cl = threadIdx.x, lane = cl & 63, wv = cl >> 6;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint64_t row = (uint64_t)b * T_BC + cl;
        const bool have = row < nloc;
        const uint64_t beg = have ? csr_ptr[row] : 0;
        const uint32_t *orow = toff + (have ? row : 0) * ((uint64_t)nj + 1);
        uint32_t o_lo = have ? orow[0] : 0u, o_hi = have ? orow[1] : 0u;
        for (uint32_t j = 0; j < nj; j++) {
            const uint64_t t = (uint64_t)b * nj + j;
            for (uint32_t i = cl; i < (T_BC / 64) * TB_BINS; i += T_BC) (&s_cnt[0][0])[i] = 0;
            if (cl < T_BC / 64) s_kmax[cl] = 0;
            const uint32_t o_next = have ? orow[min(j + 2u, nj)] : 0u;  // (the next chunk's end, requested a step ahead)
            const uint64_t lo = beg + o_lo, hi = beg + o_hi;
            // the segment's first TB_SEG entries with independent loads (one memory latency instead of one per entry); longer
            // segments finish in loops
            constexpr uint32_t TB_SEG = 16;
            uint32_t seg[TB_SEG];
            if (FILL) {
#pragma unroll
                for (uint32_t u = 0; u < TB_SEG; u++) seg[u] = lo + u < hi ? (uint32_t)c4r[lo + u] : 0xffffu;  // (all ones: code 15)
            }
            // The sort key and the row length are the segment's length INCLUDING its few overflow entries (0.8 %): the size pass then
            // reads the offsets table only, and a row that holds one gets a padding entry in its place — sums unchanged to the bit
            // (a row's entries stay in locus order, a padding entry adds an exact zero), 0.3 % more tile bytes.
            const uint32_t len = o_hi - o_lo;
            __syncthreads();
            // stable counting sort by bin = min(len, TB_BINS-1): rank inside (wave, bin) from ballots
            const uint32_t bin = min(len, (uint32_t)TB_BINS - 1u);
            uint32_t within = 0;
            {
                unsigned long long todo = ~0ull;
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    const uint32_t v = (uint32_t)__shfl((int)bin, src, 64);
                    const unsigned long long m = __ballot(bin == v);
                    if (bin == v) within = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if ((int)lane == src) s_cnt[wv][v] = (uint32_t)__popcll(m);
                    todo &= ~m;
                }
            }
            __syncthreads();
            if (cl < TB_BINS) {
                uint32_t tot = 0;
                for (int w = 0; w < T_BC / 64; w++) tot += s_cnt[w][cl];
                uint32_t inc = tot;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t o = __shfl_up(inc, off, 64);
                    if ((int)cl >= off) inc += o;
                }
                s_base[cl] = inc - tot;
            }
            __syncthreads();
            uint32_t rank = s_base[bin] + within;
            for (uint32_t w = 0; w < wv; w++) rank += s_cnt[w][bin];
            const uint32_t dw = rank >> 6, dl = rank & 63;
            atomicMax(&s_kmax[dw], len);
            __syncthreads();
            if (cl == 0) {
                uint32_t acc = 0;
                for (int w = 0; w < T_BC / 64; w++) {
                    s_sbase[w] = acc;
                    acc += 64u * ((s_kmax[w] | 1u) + 1u);
                }
                s_sbase[T_BC / 64] = acc;
            }
            __syncthreads();
            const uint32_t total = s_sbase[T_BC / 64];  // a multiple of 128 u16
            if (!FILL) {
                if (cl == 0) tile_elems[t] = (uint64_t)total;
            } else {
                const uint64_t tbase = tile_elems[t];
                if (cl < T_BC / 64) {
                    uint32_t *hd = reinterpret_cast<uint32_t *>(thdr + t * T_HDR) + 4 * cl;
                    const uint64_t first = tbase + s_sbase[cl];
                    hd[0] = (uint32_t)first;
                    hd[1] = (uint32_t)(first >> 32);
                    hd[2] = s_kmax[cl] | 1u;
                    hd[3] = s_kmax[cl] == 0u;
                }
                const uint32_t K = s_kmax[dw] | 1u;
                const bool staged = total <= (uint32_t)TB_STAGE;  // (uniform)
                uint16_t *dst = (staged ? s_tile : tiles + tbase) + s_sbase[dw] + dl * (K + 1u);
                dst[0] = (uint16_t)cl;
                uint32_t k = 0;
                const uint32_t cbase = (j * (uint32_t)T_BLU) & (LR_LOCI - 1u);
#define TB_PUT(E)                                                                                                       \
                do {                                                                                                   \
                    const uint32_t e__ = (E), code__ = e__ >> 12;                                                      \
                    if (code__ < (uint32_t)T_NCODE)                                                                    \
                        dst[1 + k++] = (uint16_t)(((uint32_t)T_NM1_OF[code__] << 14) |                                 \
                                                  ((((e__ & (LR_LOCI - 1u)) - cbase) & (LR_LOCI - 1u)) << 4) | code__); \
                } while (0)
#pragma unroll
                for (uint32_t u = 0; u < TB_SEG; u++) TB_PUT(seg[u]);
                for (uint64_t i = lo + TB_SEG; i < hi; i++) TB_PUT((uint32_t)c4r[i]);
#undef TB_PUT
                if (ORDER) s_rcnt[rank] = (uint16_t)k;
                for (; k < K; k++) dst[1 + k] = T_NULL;
                if (staged) {
                    __syncthreads();
                    if (ORDER) {  // wave wv = slice wv, lane = row
                        const uint32_t Kw = s_kmax[wv] | 1u;
                        const uint32_t rc = tile_lane_assign(s_tile + s_sbase[wv], Kw, lane, s_rcnt[wv * 64 + lane], s_scr + wv * 256);
                        tile_bank_order(s_tile + s_sbase[wv] + lane * (Kw + 1u) + 1u, Kw, rc, lane, s_scr + wv * 256);
                        __syncthreads();
                    }
                    uint4 *out = reinterpret_cast<uint4 *>(tiles + tbase);  // (tile starts are multiples of 128 u16)
                    const uint4 *in = reinterpret_cast<const uint4 *>(s_tile);
                    for (uint32_t i = cl; i < total / 8u; i += T_BC) out[i] = in[i];
                }
            }
            __syncthreads();  // (the sort's arrays and the staged tile are reused by the next chunk)
            o_lo = o_hi;
            o_hi = o_next;
        }
    }
}

// wave per row/column: count entries that are NOT regular (FILL = false) or copy them in order (FILL = true);
// REST > 0: only those outside the totals 5..REST as well (8: tier 2, i.e. totals 0 and above 8 remain)
template <bool FILL, int REST = 0>
__global__ __launch_bounds__(256) void k_ovf_build(uint64_t n_rows, const uint64_t *__restrict__ ptr,
                                                   const uint64_t *__restrict__ ent, uint64_t *__restrict__ optr,
                                                   uint64_t *__restrict__ oent)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t row = wave0; row < n_rows; row += nwaves) {
        const uint64_t beg = ptr[row], end = ptr[row + 1];
        uint64_t base = FILL ? optr[row] : 0, cnt = 0;
        for (uint64_t i0 = beg; i0 < end; i0 += 64) {
            const uint64_t i = i0 + lane;
            const uint64_t e = i < end ? ent[i] : 0;
            const bool ov = i < end && !ent_regular(e) && !(REST && ENT_ALT(e) + ENT_REF(e) - T2_NMIN <= (uint32_t)REST - T2_NMIN);
            const unsigned long long m = __ballot(ov);
            if (FILL && ov) oent[base + __popcll(m & ((1ull << lane) - 1ull))] = e;
            base += __popcll(m);
            cnt += __popcll(m);
        }
        if (!FILL && lane == 0) optr[row] = cnt;
    }
}

// wave per locus column: regular entries -> compact u32 (cell | code<<28) in order; per-code histogram
template <bool FILL, int EB>
__global__ __launch_bounds__(256) void k_c4_build(uint64_t L, const uint64_t *__restrict__ csc_ptr,
                                                  const uint64_t *__restrict__ csc_ent, uint64_t *__restrict__ c4_ptr,
                                                  uint32_t *__restrict__ c4_ent, uint32_t *__restrict__ hist_all)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t l = wave0; l < L; l += nwaves) {
        const uint64_t beg = csc_ptr[l], end = csc_ptr[l + 1];
        uint64_t base = FILL ? c4_ptr[l] : 0, cnt = 0;
        uint32_t myhist = 0;
        for (uint64_t i0 = beg; i0 < end; i0 += 64) {
            const uint64_t i = i0 + lane;
            const uint64_t e = i < end ? csc_ent[i] : 0;
            const bool reg = i < end && ent_regular(e);
            const uint32_t code = reg ? ent_code(e) : 0xffu;
            const unsigned long long m = __ballot(reg);
            if (FILL) {
                if (reg) c4_write1<EB>(c4_ent, base + __popcll(m & ((1ull << lane) - 1ull)), ENT_IDX(e), code);
            } else {
#pragma unroll
                for (int k = 0; k < T_NCODE; k++) {
                    const unsigned long long mk = __ballot(code == (uint32_t)k);
                    if (lane == k) myhist += (uint32_t)__popcll(mk);
                }
            }
            base += __popcll(m);
            cnt += __popcll(m);
        }
        if (!FILL) {
            if (lane == 0) c4_ptr[l] = (cnt + 3) & ~3ull;  // whole 16-byte vectors
            if (lane < T_NCODE) hist_all[l * T_NCODE + lane] = myhist;
        } else if ((uint64_t)lane < ((4 - (cnt & 3)) & 3)) {
            c4_write1<EB>(c4_ent, c4_ptr[l] + cnt + lane, EB == 32 ? 0x0fffffffu : 0xfffffu, 15u);  // padding: code 15 = no entry
        }
    }
}

// ===========================================================================================================
static inline unsigned gcap(uint64_t n, unsigned per_block, unsigned cap = 1u << 20)
{
    uint64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

void tiled_free(cellector_ctx *c)
{
    dev_free(c->tile_ptr); dev_free(c->tiles); dev_free(c->thdr); dev_free(c->ovf_ptr); dev_free(c->ovf_ent);
    dev_free(c->c4_ptr); dev_free(c->c4_ent); dev_free(c->ovc_ptr); dev_free(c->ovc_ent);
    dev_free(c->hist_all); dev_free(c->tab); dev_free(c->part); dev_free(c->ab3);
    dev_free(c->masked_cnt); dev_free(c->flag_bits); dev_free(c->ovf_tab); dev_free(c->ovf_etab);
    dev_free(c->ovf_sum); dev_free(c->ovf_lp); dev_free(c->ovc_locus); dev_free(c->ovf_tier_row[0]); dev_free(c->ovf_tier_row[1]); dev_free(c->ovf_tier_ent[0]); dev_free(c->ovf_tier_ent[1]); dev_free(c->ovf_tier_val); dev_free(c->ovf_ell_ptr); dev_free(c->ovf_ell); dev_free(c->ovf_nmask); dev_free(c->tile_work); dev_free(c->minlist); dev_free(c->hist_min); dev_free(c->roff); dev_free(c->c4r); dev_free(c->mroff); dev_free(c->mbeg);
    dev_free(c->t2_plist); dev_free(c->t2_slist); dev_free(c->t2_pmask); dev_free(c->hist_all2); dev_free(c->cnt2); dev_free(c->tab2); dev_free(c->ovx_ptr); dev_free(c->ovx_ent); dev_free(c->ovx_locus); dev_free(c->ovx_lp);
    dev_free(c->tile2_ptr); dev_free(c->tiles2); dev_free(c->thdr2); dev_free(c->tab2c); dev_free(c->part2); dev_free(c->tile_work2); dev_free(c->ovr_ptr); dev_free(c->ovr_ent);
    c->ovr_n = 0;
    c->t2_tiles = 0;
    c->ovx_n = 0; c->t2 = false;
    c->mroff_cap = 0;
    c->tiled_ready = false;
    c->ovf_n = 0; c->n_masked_loci = 0;
}

// Chunk groups of a tile pass (see tiled_build): the count with the shortest modelled makespan of the persistent workgroups.
static uint32_t tile_groups_for(const cellector_ctx *c, uint32_t nb, uint32_t nj)
{
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
    const uint64_t cols = (nb + T_SB_MAX - 1) / T_SB_MAX;
    uint64_t groups = 1;
    double best = 1e300;
    for (uint64_t g = 1; g <= T_GROUPS_MAX && g <= (uint64_t)nj; g++) {
        uint64_t per = (uint64_t)ncu / g;
        if (per < 1) per = 1;
        if (per > cols) per = cols;
        const uint64_t rounds = (cols + per - 1) / per, chunks = ((uint64_t)nj + g - 1) / g;
        const double cost = (double)(rounds * (chunks + 3)) * (1.0 + 0.03 * (g > T_GROUPS ? (double)(g - T_GROUPS) / T_GROUPS : 0.0));
        if (cost < best) { best = cost; groups = g; }
    }
    if (c->tile_groups_opt > 0) groups = (uint64_t)c->tile_groups_opt;  // (A/B runs)
    if (groups > nj) groups = nj;
    return (uint32_t)groups;
}

// ---- tier-2 tiles (deep coverage): a second tile set over the overflow CSR's entries with totals 5..G::NHI ----
// Built with the per-tile builder (one workgroup per tile, two searches per row in the short overflow rows); the rows keep
// their file order.  A tile holds all 1024 rows of its block, most of them with one padding entry: 4 bytes per row.
template <class G>
static cellector_status t2_tiles_build(cellector_ctx *c)
{
    const uint64_t nloc = c->nloc, L = c->L;
    c->t2_nj = (uint32_t)((L + G::BLU - 1) / G::BLU);
    if (c->t2_nj == 0) c->t2_nj = 1;
    c->t2_groups = tile_groups_for(c, c->t_nb, c->t2_nj);
    c->t2_cpg = (c->t2_nj + c->t2_groups - 1) / c->t2_groups;
    c->t2_groups = (c->t2_nj + c->t2_cpg - 1) / c->t2_cpg;
    const uint64_t nt = (uint64_t)c->t_nb * c->t2_nj, maxg = 1ull << 30;
    CHK(dev_alloc(c, &c->tile2_ptr, nt + 1));
    HIPCHK(c, hipMemsetAsync(c->tile2_ptr + nt, 0, 8, c->stream));
    for (uint64_t t0 = 0; t0 < nt; t0 += maxg) {
        const uint64_t g = nt - t0 < maxg ? nt - t0 : maxg;
        hipLaunchKernelGGL((k_tile_build<false, G>), dim3((unsigned)g), dim3(T_BC), 0, c->stream, nloc, c->t2_nj, t0, c->ovf_ptr, c->ovf_ent,
                           c->tile2_ptr, (uint16_t *)nullptr, (uint16_t *)nullptr, (const uint32_t *)nullptr);
    }
    HIPCHK(c, hipGetLastError());
    uint64_t elems = 0;
    CHK(dev_exclusive_scan_u64(c, c->tile2_ptr, nt + 1, &elems));
    CHK(dev_alloc(c, &c->tiles2, elems + 64));  // tail pad: the 16-byte load of the last row runs past its end
    CHK(dev_alloc(c, &c->thdr2, nt * T_HDR));
    for (uint64_t t0 = 0; t0 < nt; t0 += maxg) {
        const uint64_t g = nt - t0 < maxg ? nt - t0 : maxg;
        hipLaunchKernelGGL((k_tile_build<true, G>), dim3((unsigned)g), dim3(T_BC), 0, c->stream, nloc, c->t2_nj, t0, c->ovf_ptr, c->ovf_ent,
                           c->tile2_ptr, c->tiles2, c->thdr2, (const uint32_t *)nullptr);
    }
    HIPCHK(c, hipGetLastError());
    const uint64_t tab_elems = (uint64_t)c->t2_nj * G::LROW * G::BL + 4 * T_THREADS;  // tail pad: the partial last table load
    CHK(dev_alloc(c, &c->tab2c, tab_elems));
    HIPCHK(c, hipMemsetAsync(c->tab2c, 0, tab_elems * sizeof(double), c->stream));  // (zero slots and the slots beyond L stay zero)
    CHK(dev_alloc(c, &c->part2, 3ull * 2 * c->t2_groups * c->t_npad));
    CHK(dev_alloc(c, &c->tile_work2, T_GROUPS_MAX));
    // what the tiles leave to the per-entry kernel (totals 0 and above G::NHI), as a by-cell CSR of its own: walking the whole
    // overflow CSR and skipping the tiles' entries kept that kernel's waves as long as before (a wave waits for its slowest lane)
    CHK(dev_alloc(c, &c->ovr_ptr, nloc + 1));
    HIPCHK(c, hipMemsetAsync(c->ovr_ptr + nloc, 0, 8, c->stream));
    hipLaunchKernelGGL((k_ovf_build<false, (int)G::NHI>), dim3(gcap(nloc, 4)), dim3(256), 0, c->stream, nloc, c->ovf_ptr, c->ovf_ent, c->ovr_ptr,
                       (uint64_t *)nullptr);
    CHK(dev_exclusive_scan_u64(c, c->ovr_ptr, nloc + 1, &c->ovr_n));
    CHK(dev_alloc(c, &c->ovr_ent, c->ovr_n));
    hipLaunchKernelGGL((k_ovf_build<true, (int)G::NHI>), dim3(gcap(nloc, 4)), dim3(256), 0, c->stream, nloc, c->ovf_ptr, c->ovf_ent, c->ovr_ptr,
                       c->ovr_ent);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

// one tier-2 tile pass on the main stream: this pass' chunked tables, then the tile kernel over the second tile set
template <class G>
static cellector_status t2_tiles_pass_g(cellector_ctx *c, const double2 *ab, int set, bool expected)
{
    constexpr uint32_t NE = G::NHI - G::NLO + 1;
    const unsigned gp = gcap(c->L * G::NCODE, 256, 0x7fffffffu), ge = expected ? gcap(c->L * NE, 256, 0x7fffffffu) : 0u;
    if (expected)
        hipLaunchKernelGGL((k_t2c_tables<true, G>), dim3(gp + ge), dim3(256), 0, c->stream, c->L, gp, ab, (const double *)c->lf, c->tab2c);
    else
        hipLaunchKernelGGL((k_t2c_tables<false, G>), dim3(gp), dim3(256), 0, c->stream, c->L, gp, ab, (const double *)c->lf, c->tab2c);
    double *part_ll = c->part2 + (uint64_t)set * 2 * c->t2_groups * c->t_npad;
    double *part_ell = part_ll + (uint64_t)c->t2_groups * c->t_npad;
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
    int sb = T_SB_MAX;
    while (sb > 2 && (uint64_t)((c->t_nb + sb - 1) / sb) * c->t2_groups * 10 < (uint64_t)ncu * 9) sb >>= 1;
    const uint32_t n_cols = (c->t_nb + sb - 1) / sb;
    uint32_t per_group = (uint32_t)ncu / c->t2_groups;
    if (per_group < 1) per_group = 1;
    if (per_group > n_cols) per_group = n_cols;
    const dim3 grid(per_group * c->t2_groups);
    HIPCHK(c, hipMemsetAsync(c->tile_work2, 0, T_GROUPS_MAX * sizeof(uint32_t), c->stream));
#define LAUNCH_TILE2(E, S)                                                                                                 \
    hipLaunchKernelGGL((k_tile_ll<E, S, G>), grid, dim3(T_THREADS), 0, c->stream, c->t_nb, c->t2_nj, c->t2_cpg, c->t2_groups, n_cols, \
                       c->tile_work2, (const uint16_t *)c->thdr2, (const uint16_t *)c->tiles2, (const double *)c->tab2c, c->t_npad,    \
                       part_ll, part_ell)
    if (expected) {
        if (sb == 4) LAUNCH_TILE2(true, 4); else LAUNCH_TILE2(true, 2);
    } else {
        if (sb == 4) LAUNCH_TILE2(false, 4); else LAUNCH_TILE2(false, 2);
    }
#undef LAUNCH_TILE2
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}
static cellector_status t2_tiles_pass(cellector_ctx *c, const double2 *ab, int set, bool expected)
{
    if (!c->t2_tiles) return CELLECTOR_OK;
    return c->t2_tiles == 8 ? t2_tiles_pass_g<geo_t2<8>>(c, ab, set, expected) : t2_tiles_pass_g<geo_t2<6>>(c, ab, set, expected);
}
// ... and its partial sums added to the overflow sums of the set (behind the side stream's kernels, which write those)
static cellector_status t2_tiles_add(cellector_ctx *c, int set, bool expected)
{
    if (!c->t2_tiles) return CELLECTOR_OK;
    double *part_ll = c->part2 + (uint64_t)set * 2 * c->t2_groups * c->t_npad, *part_ell = part_ll + (uint64_t)c->t2_groups * c->t_npad;
    double *o_ll = c->ovf_sum + (uint64_t)set * 2 * c->nloc, *o_ell = o_ll + c->nloc;
    const unsigned g = gcap(c->nloc, 256, 0x7fffffffu);
    if (expected)
        hipLaunchKernelGGL(k_t2_tile_add<true>, dim3(g), dim3(256), 0, c->stream, c->nloc, c->t2_groups, c->t_npad, part_ll, part_ell, o_ll, o_ell);
    else
        hipLaunchKernelGGL(k_t2_tile_add<false>, dim3(g), dim3(256), 0, c->stream, c->nloc, c->t2_groups, c->t_npad, part_ll, part_ell, o_ll, o_ell);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

cellector_status tiled_build(cellector_ctx *c)
{
    const uint64_t nloc = c->nloc, L = c->L;
    if (nloc >= (1ull << 28)) return ctx_fail(c, CELLECTOR_EINVAL, "tiled engine: more than 2^28 cells per shard");
    if (L >= (1ull << 28)) return ctx_fail(c, CELLECTOR_EINVAL, "tiled engine: more than 2^28 loci");
    c->t_nb = (uint32_t)((nloc + T_BC - 1) / T_BC);
    c->t_nj = (uint32_t)((L + T_BLU - 1) / T_BLU);
    if (c->t_nb == 0) c->t_nb = 1;
    if (c->t_nj == 0) c->t_nj = 1;
    // Chunk groups.  The tile kernel runs one persistent workgroup per CU, each bound to a group of locus chunks and fetching
    // columns of T_SB_MAX cell blocks from the group's counter; a cell gets one partial sum per group.  Every workgroup of a
    // group walks the group's chunks once per column it fetches, so the kernel takes about rounds(g) x (chunks(g) + 3)
    // chunk-steps with rounds = ceil(columns / workgroups per group) — a column costs its chunks plus a fixed part
    // (accumulators cleared and written out as partial sums), put at three chunk-steps.  Any count from 1 up is taken, the one
    // with the shortest makespan wins (ties: fewer groups), charging 3 % per 8 groups beyond 8 for the additional partial
    // sums (16 bytes more per cell and pass written by the tile kernel and read by the finalize).  Measured (ms per EM
    // iteration): 10^6 cells x 200k loci (245 columns, 313 chunks): 1 group 2.33, 2: 2.34, 4: 2.36, 7: 2.43, 8: 2.44, 32:
    // 2.59 — with one group every workgroup does one column over all chunks: no ragged last round, no partial sums to add
    // up; 200k cells x 100k loci (49 columns, 157 chunks): 5 groups 0.394 (245 workgroups, one round), 8: 0.430 (two rounds,
    // the second half empty), 2: 0.48, 1: 0.73 (49 CUs busy).  Groups used to be multiples of 8 so that a group's workgroups
    // shared an XCD's L2 for the table reads (workgroup i runs on XCD i mod 8): the figures above show no such need — the
    // workgroups of a group walk the chunks in step, a table chunk is fetched once per XCD either way.
    c->t_groups = tile_groups_for(c, c->t_nb, c->t_nj);
    c->t_cpg = (c->t_nj + c->t_groups - 1) / c->t_groups;
    c->t_groups = (c->t_nj + c->t_cpg - 1) / c->t_cpg;
    c->t_npad = (uint64_t)c->t_nb * T_BC;
    const uint64_t nt = (uint64_t)c->t_nb * c->t_nj;

    // ---- tiles
    CHK(dev_alloc(c, &c->tile_ptr, nt + 1));
    HIPCHK(c, hipMemsetAsync(c->tile_ptr + nt, 0, 8, c->stream));
    // every (cell, chunk) pair's first entry, once per row: the two builder passes searched each row per tile (two binary
    // searches of ~11 scattered probes per cell and tile: 1.4 TB through the L2 at 1M x 200k, 0.2 s)
    uint32_t *toff = nullptr;
    if (nloc && dev_alloc(c, &toff, nloc * ((uint64_t)c->t_nj + 1)) == CELLECTOR_OK)
        hipLaunchKernelGGL(k_range_offsets, dim3(gcap(nloc, 4)), dim3(256), 0, c->stream, nloc, c->t_nj, (uint32_t)T_BLU, c->csr_ptr,
                           c->csr_ent, toff);
    else
        toff = nullptr;  // (no room for the table: the builder searches)
    // the compact by-cell entries (also the minority-driven locus pass' input): with the offsets table the builder reads these
    CHK(dev_alloc(c, &c->c4r, c->nnz));
    if (c->nnz)
        hipLaunchKernelGGL(k_cell_compact, dim3(gcap(c->nnz, 256, 0x7fffffffu)), dim3(256), 0, c->stream, c->nnz, c->csr_ent, c->c4r);
    int ncu_b = 256;
    (void)hipDeviceGetAttribute(&ncu_b, hipDeviceAttributeMultiprocessorCount, c->device);
    const unsigned bgrid = (unsigned)std::min<uint64_t>(c->t_nb, (uint64_t)ncu_b * 8);  // (one block per CU at a time: LDS)
    const uint64_t maxg = 1ull << 30;
    if (toff) {
        hipLaunchKernelGGL(k_tile_build2<false>, dim3(bgrid), dim3(T_BC), 0, c->stream, nloc, c->t_nb, c->t_nj, c->csr_ptr, c->c4r, toff,
                           c->tile_ptr, (uint16_t *)nullptr, (uint16_t *)nullptr);
    } else {
        for (uint64_t t0 = 0; t0 < nt; t0 += maxg) {
            const uint64_t g = nt - t0 < maxg ? nt - t0 : maxg;
            hipLaunchKernelGGL(k_tile_build<false>, dim3((unsigned)g), dim3(T_BC), 0, c->stream, nloc, c->t_nj, t0, c->csr_ptr,
                               c->csr_ent, c->tile_ptr, (uint16_t *)nullptr, (uint16_t *)nullptr, toff);
        }
    }
    HIPCHK(c, hipGetLastError());
    uint64_t elems = 0;
    CHK(dev_exclusive_scan_u64(c, c->tile_ptr, nt + 1, &elems));
    CHK(dev_alloc(c, &c->tiles, elems + 64));  // tail pad: the 16-byte load of the last row runs past its end
    CHK(dev_alloc(c, &c->thdr, nt * T_HDR));
    c->t_elems = elems;
    if (toff) {
        if (c->bank_order)
            hipLaunchKernelGGL((k_tile_build2<true, true>), dim3(bgrid), dim3(T_BC), 0, c->stream, nloc, c->t_nb, c->t_nj, c->csr_ptr, c->c4r,
                               toff, c->tile_ptr, c->tiles, c->thdr);
        else
            hipLaunchKernelGGL((k_tile_build2<true, false>), dim3(bgrid), dim3(T_BC), 0, c->stream, nloc, c->t_nb, c->t_nj, c->csr_ptr, c->c4r,
                               toff, c->tile_ptr, c->tiles, c->thdr);
    } else {
        for (uint64_t t0 = 0; t0 < nt; t0 += maxg) {
            const uint64_t g = nt - t0 < maxg ? nt - t0 : maxg;
            hipLaunchKernelGGL(k_tile_build<true>, dim3((unsigned)g), dim3(T_BC), 0, c->stream, nloc, c->t_nj, t0, c->csr_ptr,
                               c->csr_ent, c->tile_ptr, c->tiles, c->thdr, toff);
        }
    }
    if (toff) {
        HIPCHK(c, hipStreamSynchronize(c->stream));  // (the block goes back to the allocation cache: no kernel may still read it)
        dev_free(toff);
    }
    HIPCHK(c, hipGetLastError());

    // ---- overflow CSR
    CHK(dev_alloc(c, &c->ovf_ptr, nloc + 1));
    HIPCHK(c, hipMemsetAsync(c->ovf_ptr + nloc, 0, 8, c->stream));
    if (nloc)
        hipLaunchKernelGGL(k_ovf_build<false>, dim3(gcap(nloc, 4)), dim3(256), 0, c->stream, nloc, c->csr_ptr, c->csr_ent,
                           c->ovf_ptr, (uint64_t *)nullptr);
    CHK(dev_exclusive_scan_u64(c, c->ovf_ptr, nloc + 1, &c->ovf_n));
    CHK(dev_alloc(c, &c->ovf_ent, c->ovf_n));
    if (nloc)
        hipLaunchKernelGGL(k_ovf_build<true>, dim3(gcap(nloc, 4)), dim3(256), 0, c->stream, nloc, c->csr_ptr, c->csr_ent,
                           c->ovf_ptr, c->ovf_ent);
    HIPCHK(c, hipGetLastError());

    // ---- compact CSC + histogram, overflow CSC
    uint64_t n4 = 0, novc = 0;
    CHK(dev_alloc(c, &c->c4_ptr, L + 1));
    CHK(dev_alloc(c, &c->ovc_ptr, L + 1));
    CHK(dev_alloc(c, &c->hist_all, L * T_NCODE));
    HIPCHK(c, hipMemsetAsync(c->c4_ptr + L, 0, 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->ovc_ptr + L, 0, 8, c->stream));
    if (L) {
        hipLaunchKernelGGL((k_c4_build<false, 32>), dim3(gcap(L, 4)), dim3(256), 0, c->stream, L, c->csc_ptr, c->csc_ent,
                           c->c4_ptr, (uint32_t *)nullptr, c->hist_all);
        hipLaunchKernelGGL(k_ovf_build<false>, dim3(gcap(L, 4)), dim3(256), 0, c->stream, L, c->csc_ptr, c->csc_ent,
                           c->ovc_ptr, (uint64_t *)nullptr);
    }
    HIPCHK(c, hipGetLastError());
    CHK(dev_exclusive_scan_u64(c, c->c4_ptr, L + 1, &n4));
    CHK(dev_exclusive_scan_u64(c, c->ovc_ptr, L + 1, &novc));
    if (novc != c->ovf_n || n4 < c->nnz - novc)
        return ctx_fail(c, CELLECTOR_EDEVICE, "internal: tiled build entry counts inconsistent (%llu + %llu vs %llu, ovf %llu)",
                        (unsigned long long)n4, (unsigned long long)novc, (unsigned long long)c->nnz,
                        (unsigned long long)c->ovf_n);
    c->c4_bits = (nloc <= (1ull << 20) && c->c4_bits_opt != 32) ? 24 : 32;
    CHK(dev_alloc(c, &c->c4_ent, c->c4_bits == 32 ? n4 : (n4 * 3 + 3) / 4 + 4));
    CHK(dev_alloc(c, &c->ovc_ent, novc));
    if (L) {
        if (c->c4_bits == 32)
            hipLaunchKernelGGL((k_c4_build<true, 32>), dim3(gcap(L, 4)), dim3(256), 0, c->stream, L, c->csc_ptr, c->csc_ent,
                               c->c4_ptr, c->c4_ent, c->hist_all);
        else
            hipLaunchKernelGGL((k_c4_build<true, 24>), dim3(gcap(L, 4)), dim3(256), 0, c->stream, L, c->csc_ptr, c->csc_ent,
                               c->c4_ptr, c->c4_ent, c->hist_all);
        hipLaunchKernelGGL(k_ovf_build<true>, dim3(gcap(L, 4)), dim3(256), 0, c->stream, L, c->csc_ptr, c->csc_ent,
                           c->ovc_ptr, c->ovc_ent);
    }
    HIPCHK(c, hipGetLastError());

    // ---- overflow entries: which paths they take
    if (c->ovf_n >= (1ull << 32)) return ctx_fail(c, CELLECTOR_EINVAL, "tiled engine: more than 2^32 overflow entries per shard");
    // deep coverage: more than 3 % of the entries outside the tables (0.8 % with vartrix-like totals 1 + Geometric(0.7),
    // 13 % with 1 + Geometric(0.4)) — the side-stream arrangement built for "a few entries per row" no longer hides them
    c->ovf_deep = c->ovf_deep_opt >= 0 ? c->ovf_deep_opt != 0 : (c->ovf_n * 100 > c->nnz * 3);
    // tier 2 (k_t2_tables): on by default.  A deep matrix takes it on the LOCUS side only (counts instead of 1300 per-entry
    // evaluations per locus: locus pass 1.6 -> 1.0 ms at 10^6 cells x 200k loci deep); its cell side stays with the arithmetic
    // kernel — 2.3e8 lookups of a line each out of a 77 MB table cost more than evaluating the entries (measured: 9.8 vs 6.1 ms).
    c->t2 = c->ovf_n != 0 && L != 0 && L < (1ull << 27) /* the pair list's keys */ && (c->t2_opt >= 0 ? c->t2_opt != 0 : true);
    // tier-2 tiles: the cell side of the totals 5..8 (or 5..6) of a deep matrix walks tiles of its own (t2_tiles_build)
    c->t2_tiles = 0;
    if (c->ovf_deep && c->ovf_deep_wide && c->ovf_n && nloc && L) c->t2_tiles = c->t2_tiles_opt < 0 ? 8 : c->t2_tiles_opt;
    if (c->t2_tiles == 8) CHK(t2_tiles_build<geo_t2<8>>(c));
    else if (c->t2_tiles == 6) CHK(t2_tiles_build<geo_t2<6>>(c));
    CHK(dev_alloc(c, &c->ovf_sum, 3 * 2 * nloc));
    CHK(dev_alloc(c, &c->ovf_tab, L * OV_ROW));
    CHK(dev_alloc(c, &c->ovc_locus, c->ovf_n));
    if (L && c->ovf_n)
        hipLaunchKernelGGL(k_ovf_locus_ids, dim3(gcap(L, 4)), dim3(256), 0, c->stream, L, c->ovc_ptr, c->ovc_locus);
    if (!(c->t2 && !c->ovf_deep)) {  // (the per-entry paths of a shard without tier 2, and the deep forms)
        CHK(dev_alloc(c, &c->ovf_lp, c->ovf_n));
        CHK(dev_alloc(c, &c->ovf_etab, L * OV_REC));
    }
    CHK(dev_alloc(c, &c->ovf_nmask, L));
    c->ovx_n = 0;
    if (c->t2) {
        // tier 2: static pair histogram, per-iteration counters and table; the by-locus CSC of the entries outside it
        CHK(dev_alloc(c, &c->hist_all2, L * T2_CSTRIDE));
        CHK(dev_alloc(c, &c->cnt2, L * T2_CSTRIDE));
        CHK(dev_alloc(c, &c->tab2, L * T2_ROW));
        HIPCHK(c, hipMemsetAsync(c->hist_all2, 0, L * T2_CSTRIDE * sizeof(uint32_t), c->stream));
        HIPCHK(c, hipMemsetAsync(c->cnt2, 0, L * T2_CSTRIDE * sizeof(uint32_t), c->stream));
        hipLaunchKernelGGL(k_t2_hist, dim3(gcap(c->ovf_n, 256, 0x7fffffffu)), dim3(256), 0, c->stream, c->ovf_n, c->ovc_locus, c->ovc_ent,
                           c->hist_all2);
        {
            uint64_t *np = nullptr, *ns = nullptr, tot_p = 0, tot_s = 0;
            CHK(dev_alloc(c, &np, L + 1));
            CHK(dev_alloc(c, &ns, L + 1));
            HIPCHK(c, hipMemsetAsync(np + L, 0, 8, c->stream));
            HIPCHK(c, hipMemsetAsync(ns + L, 0, 8, c->stream));
            hipLaunchKernelGGL(k_t2_lists<false>, dim3(gcap(L, 256)), dim3(256), 0, c->stream, L, c->hist_all2, np, ns, (uint32_t *)nullptr,
                               (uint32_t *)nullptr, (uint32_t *)nullptr);
            cellector_status st = dev_exclusive_scan_u64(c, np, L + 1, &tot_p);
            if (st == CELLECTOR_OK) st = dev_exclusive_scan_u64(c, ns, L + 1, &tot_s);
            if (st == CELLECTOR_OK) st = dev_alloc(c, &c->t2_plist, tot_p);
            if (st == CELLECTOR_OK) st = dev_alloc(c, &c->t2_slist, tot_s);
            if (st == CELLECTOR_OK) st = dev_alloc(c, &c->t2_pmask, L);
            if (st == CELLECTOR_OK) {
                hipLaunchKernelGGL(k_t2_lists<true>, dim3(gcap(L, 256)), dim3(256), 0, c->stream, L, c->hist_all2, np, ns, c->t2_plist, c->t2_slist,
                                   c->t2_pmask);
                if (hipStreamSynchronize(c->stream) != hipSuccess) st = ctx_fail(c, CELLECTOR_EDEVICE, "tier-2 list build failed");
            }
            dev_free(np); dev_free(ns);
            CHK(st);
            c->t2_np = (uint32_t)tot_p; c->t2_ns = (uint32_t)tot_s;
        }
        CHK(dev_alloc(c, &c->ovx_ptr, L + 1));
        HIPCHK(c, hipMemsetAsync(c->ovx_ptr + L, 0, 8, c->stream));
        hipLaunchKernelGGL((k_ovf_build<false, 8>), dim3(gcap(L, 4)), dim3(256), 0, c->stream, L, c->ovc_ptr, c->ovc_ent, c->ovx_ptr,
                           (uint64_t *)nullptr);
        CHK(dev_exclusive_scan_u64(c, c->ovx_ptr, L + 1, &c->ovx_n));
        CHK(dev_alloc(c, &c->ovx_ent, c->ovx_n));
        CHK(dev_alloc(c, &c->ovx_locus, c->ovx_n));
        CHK(dev_alloc(c, &c->ovx_lp, c->ovx_n));
        hipLaunchKernelGGL((k_ovf_build<true, 8>), dim3(gcap(L, 4)), dim3(256), 0, c->stream, L, c->ovc_ptr, c->ovc_ent, c->ovx_ptr,
                           c->ovx_ent);
        if (c->ovx_n)
            hipLaunchKernelGGL(k_ovf_locus_ids, dim3(gcap(L, 4)), dim3(256), 0, c->stream, L, c->ovx_ptr, c->ovx_locus);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dev_free(c->ovc_locus);  // (only the histogram needed it)
    }
    {
        const uint64_t n_grp = (nloc + 63) / 64;
        uint64_t slots = 0;
        CHK(dev_alloc(c, &c->ovf_ell_ptr, n_grp + 1));
        HIPCHK(c, hipMemsetAsync(c->ovf_ell_ptr + n_grp, 0, 8, c->stream));
        if (nloc)
            hipLaunchKernelGGL(k_ovf_ell_build<false>, dim3(gcap(n_grp * 64, 256, 0x7fffffffu)), dim3(256), 0, c->stream, nloc,
                               c->ovf_ptr, c->ovf_ent, c->ovf_ell_ptr, (uint64_t *)nullptr);
        CHK(dev_exclusive_scan_u64(c, c->ovf_ell_ptr, n_grp + 1, &slots));
        CHK(dev_alloc(c, &c->ovf_ell, slots));
        if (nloc)
            hipLaunchKernelGGL(k_ovf_ell_build<true>, dim3(gcap(n_grp * 64, 256, 0x7fffffffu)), dim3(256), 0, c->stream, nloc,
                               c->ovf_ptr, c->ovf_ent, c->ovf_ell_ptr, c->ovf_ell);
        HIPCHK(c, hipGetLastError());
    }
    // the tier lists of the entries the fast cell-side kernel leaves out
    c->ovf_n_tier[0] = c->ovf_n_tier[1] = 0;
    if (nloc && c->ovf_n) {
        uint64_t *cnt0 = nullptr, *cnt1 = nullptr;
        CHK(dev_alloc(c, &cnt0, nloc + 1));
        CHK(dev_alloc(c, &cnt1, nloc + 1));
        HIPCHK(c, hipMemsetAsync(cnt0 + nloc, 0, 8, c->stream));
        HIPCHK(c, hipMemsetAsync(cnt1 + nloc, 0, 8, c->stream));
        const unsigned g = gcap(nloc, 256, 0x7fffffffu);
        hipLaunchKernelGGL(k_ovf_tier_lists<false>, dim3(g), dim3(256), 0, c->stream, nloc, c->ovf_ptr, c->ovf_ent, cnt0, cnt1,
                           (uint32_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr);
        cellector_status st = dev_exclusive_scan_u64(c, cnt0, nloc + 1, &c->ovf_n_tier[0]);
        if (st == CELLECTOR_OK) st = dev_exclusive_scan_u64(c, cnt1, nloc + 1, &c->ovf_n_tier[1]);
        for (int t = 0; t < 2 && st == CELLECTOR_OK; t++) {
            st = dev_alloc(c, &c->ovf_tier_row[t], c->ovf_n_tier[t]);
            if (st == CELLECTOR_OK) st = dev_alloc(c, &c->ovf_tier_ent[t], c->ovf_n_tier[t]);
        }
        if (st == CELLECTOR_OK) {
            hipLaunchKernelGGL(k_ovf_tier_lists<true>, dim3(g), dim3(256), 0, c->stream, nloc, c->ovf_ptr, c->ovf_ent, cnt0, cnt1,
                               c->ovf_tier_row[0], c->ovf_tier_ent[0], c->ovf_tier_row[1], c->ovf_tier_ent[1]);
            if (hipStreamSynchronize(c->stream) != hipSuccess) st = ctx_fail(c, CELLECTOR_EDEVICE, "tier list build failed");
        }
        dev_free(cnt0); dev_free(cnt1);
        CHK(st);
        CHK(dev_alloc(c, &c->ovf_tier_val, 2 * c->ovf_n_tier[1]));  // (log-pmf, expected term) of the tier-1 entries, per pass
    }
    // which totals the per-entry tables (k_ovf_tables, k_ovf_tables_e) must cover at every locus: those of the entries that
    // take these paths
    if (L && c->ovf_n) {
        if (c->t2 && !c->ovf_deep)  // (a deep matrix' cell side evaluates every overflow entry: its tables cover all totals)
            hipLaunchKernelGGL(k_ovf_nmask, dim3(gcap(L, 4)), dim3(256), 0, c->stream, L, c->ovx_ptr, c->ovx_ent, c->ovf_nmask);
        else
            hipLaunchKernelGGL(k_ovf_nmask, dim3(gcap(L, 4)), dim3(256), 0, c->stream, L, c->ovc_ptr, c->ovc_ent, c->ovf_nmask);
    }
    HIPCHK(c, hipGetLastError());

    // ---- per-iteration workspaces
    // tables: three log-pmf-only sets (posterior passes; set 0 also serves an EM pass without the expected column),
    // then one set of (log-pmf, expected) pairs; tail pad for the unconditional partial last table load
    const uint64_t tab_elems = (uint64_t)c->t_nj * TAB_ELEMS;
    CHK(dev_alloc(c, &c->tab, 4 * tab_elems + 4 * T_THREADS));
    c->tab_em = c->tab;
    c->tab_em_stride = 1;
    CHK(dev_alloc(c, &c->part, 3ull * 2 * c->t_groups * c->t_npad));
    CHK(dev_alloc(c, &c->ab3, 3 * L));
    CHK(dev_alloc(c, &c->masked_cnt, nloc));
    CHK(dev_alloc(c, &c->flag_bits, (nloc + 31) / 32 + 1));
    CHK(dev_alloc(c, &c->tile_work, 3 * T_GROUPS_MAX));
    CHK(dev_alloc(c, &c->minlist, nloc));
    {
        // subsets of the exclusion set: enough (range, subset) workgroups to fill the chip once
        int ncu = 256;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
        const uint32_t R = (uint32_t)((L + LR_LOCI - 1) / LR_LOCI);
        uint32_t sub = R ? (uint32_t)ncu / R : 1;  // one workgroup per CU (LDS): at most one round of them
        if (sub < 1) sub = 1;
        if (sub > LR_SUB_MAX) sub = LR_SUB_MAX;
        c->lr_sub = sub;
        CHK(dev_alloc(c, &c->hist_min, (uint64_t)sub * L * 16));
        CHK(dev_alloc(c, &c->roff, nloc * (R + 1)));
        if (nloc)
            hipLaunchKernelGGL(k_range_offsets, dim3(gcap(nloc, 4)), dim3(256), 0, c->stream, nloc, R, (uint32_t)LR_LOCI, c->csr_ptr, c->csr_ent,
                               c->roff);
        HIPCHK(c, hipGetLastError());
        // the transposed offsets of the excluded cells, sized for the largest exclusion set the automatic choice hands to the
        // minority-driven form (allocated here, not in the first iteration's locus pass: that cost the first iteration a
        // stream synchronisation and two allocations — and a run has few iterations)
        if (nloc && c->locus_mode != 1) {
            c->mroff_cap = ((nloc * LM_NUM / LM_DEN + LT_CELLS) + 63) & ~63ull;
            CHK(dev_alloc(c, &c->mroff, (uint64_t)(R + 1) * c->mroff_cap));
            CHK(dev_alloc(c, &c->mbeg, c->mroff_cap));
        }
    }
    HIPCHK(c, hipMemsetAsync(c->masked_cnt, 0, (nloc ? nloc : 1) * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->flag_bits, 0, ((nloc + 31) / 32 + 1) * 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->tiled_ready = true;
    return CELLECTOR_OK;
}

// Overflow side of one pass: tables -> per-entry values (locus-major) -> per-cell sums.  Launched on the side stream so
// that these small, latency-bound kernels run next to the tile kernel instead of in front of it.
// cell side of the overflow entries of one pass, on stream `st`
static void launch_overflow_cell(cellector_ctx *c, hipStream_t st, const double2 *ab, int set, bool expected)
{
    double *o_ll = c->ovf_sum + (uint64_t)set * 2 * c->nloc, *o_ell = o_ll + c->nloc;
    const unsigned g = gcap(c->nloc, 256, 0x7fffffffu), eg = gcap(c->L * 16, 256, 0x7fffffffu);
    const bool deep = c->ovf_deep;  // the overflow share is large: full form, never throttled, no tier-0 list
    if (c->t2 && !deep) {
        // tier 2: the pairs' table of this pass, then the rows' lookups (writes the sums); the other totals ADD to them
        const unsigned gp = gcap(c->t2_np, 256, 0x7fffffffu), gs = expected ? gcap(c->t2_ns, 256, 0x7fffffffu) : 0u;
        // (side_lds > 0: a request for dynamic LDS the lookup kernel does not use, to limit its blocks per CU in A/B runs)
        const size_t lds_req = c->side_lds > 0 ? (size_t)c->side_lds : 0;
        // waves of the lookup kernel: all groups at once when it has the machine to itself, option t2_waves (default 512) beside the tile kernel
        const uint64_t n_grp = (c->nloc + 63) / 64;
        const unsigned cg = (unsigned)std::min<uint64_t>(n_grp ? n_grp : 1, st == c->side ? (uint64_t)c->t2_waves : 0x7fffffffull);
        const bool need_e = expected && c->ovf_n_tier[0];  // E(9..17) of the tier-0 entries
#define T2_CELL(E)                                                                                                         \
        do {                                                                                                               \
            hipLaunchKernelGGL(k_t2_tables<E>, dim3(gp + gs), dim3(256), 0, st, c->t2_np, c->t2_plist, c->t2_ns, c->t2_slist, gp, ab, c->lf, \
                               c->tab2);                                                                                   \
            hipLaunchKernelGGL(k_t2_cell<E>, dim3(cg), dim3(64), lds_req, st, c->nloc, c->ovf_ell_ptr, c->ovf_ell, c->tab2, o_ll, o_ell); \
            if (need_e)                                                                                                    \
                hipLaunchKernelGGL(k_ovf_tables_e, dim3(eg), dim3(256), 0, st, c->L, ab, c->ovf_nmask, c->ovf_tab, c->ovf_etab); \
            if (c->ovf_n_tier[0])                                                                                          \
                hipLaunchKernelGGL((k_ovf_cell_listed<E, false>), dim3(gcap(c->ovf_n_tier[0], 256, 0x7fffffffu)), dim3(256), 0, st, \
                                   c->ovf_n_tier[0], c->ovf_tier_row[0], c->ovf_tier_ent[0], ab, c->lf, c->ovf_tab, o_ll, o_ell); \
            if (c->ovf_n_tier[1]) {                                                                                        \
                hipLaunchKernelGGL(k_ovf_listed_values<E>, dim3(gcap(c->ovf_n_tier[1], 256, 0x7fffffffu)), dim3(256), 0, st, \
                                   c->ovf_n_tier[1], c->ovf_tier_ent[1], ab, c->lf, c->ovf_tier_val, c->ovf_tier_val + c->ovf_n_tier[1]); \
                hipLaunchKernelGGL(k_ovf_listed_add<E>, dim3(gcap(c->ovf_n_tier[1], 256, 0x7fffffffu)), dim3(256), 0, st, c->ovf_n_tier[1], \
                                   c->ovf_tier_row[1], c->ovf_tier_val, c->ovf_tier_val + c->ovf_n_tier[1], o_ll, o_ell);  \
            }                                                                                                              \
        } while (0)
        if (expected) T2_CELL(true); else T2_CELL(false);
#undef T2_CELL
        return;
    }
    if (expected) {
        hipLaunchKernelGGL(k_ovf_tables_e, dim3(eg), dim3(256), 0, st, c->L, ab, c->ovf_nmask, c->ovf_tab, c->ovf_etab);
        // Residency throttle: a request for dynamic LDS it does not use leaves room for only ONE block of this kernel beside
        // a tile workgroup (one wave per SIMD instead of two).  On a big shard the kernel still ends well inside the tile
        // kernel and disturbs it less (cfg4: 2.83 -> 2.78 ms per iteration); a small shard's tile kernel is too short for that.
        const size_t lds_req = deep ? 0 : c->side_lds >= 0 ? (size_t)c->side_lds : (st == c->side && c->nloc >= (1ull << 19) ? 5000 : 0);
        if (deep && c->ovf_deep_wide)
            hipLaunchKernelGGL(k_ovf_cell_wide<true>, dim3(gcap(c->nloc * LF_LANES, 256, 0x7fffffffu)), dim3(256), 0, st, c->nloc, c->t2_tiles ? c->ovr_ptr : c->ovf_ptr,
                               c->t2_tiles ? c->ovr_ent : c->ovf_ent, ab, c->lf, c->ovf_etab, c->ovf_tab, o_ll, o_ell);
        else if (deep)
            hipLaunchKernelGGL((k_ovf_cell_direct<true, false, true>), dim3(g), dim3(256), 0, st, c->nloc, c->ovf_ell_ptr, c->ovf_ell, ab,
                               c->lf, c->ovf_etab, c->ovf_tab, o_ll, o_ell);
        else if (lds_req)  // one block per CU: the 64-VGPR form with the packed per-locus record
            hipLaunchKernelGGL((k_ovf_cell_direct<true, true, false>), dim3(g), dim3(256), lds_req, st, c->nloc, c->ovf_ell_ptr, c->ovf_ell,
                               ab, c->lf, c->ovf_etab, c->ovf_tab, o_ll, o_ell);
        else
            hipLaunchKernelGGL((k_ovf_cell_direct<true, false, false>), dim3(g), dim3(256), 0, st, c->nloc, c->ovf_ell_ptr, c->ovf_ell, ab,
                               c->lf, c->ovf_etab, c->ovf_tab, o_ll, o_ell);
        if (c->ovf_n_tier[0] && !deep)
            hipLaunchKernelGGL((k_ovf_cell_listed<true, false>), dim3(gcap(c->ovf_n_tier[0], 256, 0x7fffffffu)), dim3(256), 0, st,
                               c->ovf_n_tier[0], c->ovf_tier_row[0], c->ovf_tier_ent[0], ab, c->lf, c->ovf_tab, o_ll, o_ell);
        if (c->ovf_n_tier[1]) {
            hipLaunchKernelGGL(k_ovf_listed_values<true>, dim3(gcap(c->ovf_n_tier[1], 256, 0x7fffffffu)), dim3(256), 0, st,
                               c->ovf_n_tier[1], c->ovf_tier_ent[1], ab, c->lf, c->ovf_tier_val, c->ovf_tier_val + c->ovf_n_tier[1]);
            hipLaunchKernelGGL(k_ovf_listed_add<true>, dim3(gcap(c->ovf_n_tier[1], 256, 0x7fffffffu)), dim3(256), 0, st, c->ovf_n_tier[1],
                               c->ovf_tier_row[1], c->ovf_tier_val, c->ovf_tier_val + c->ovf_n_tier[1], o_ll, o_ell);
        }
    } else {
        if (deep && c->ovf_deep_wide)
            hipLaunchKernelGGL(k_ovf_cell_wide<false>, dim3(gcap(c->nloc * LF_LANES, 256, 0x7fffffffu)), dim3(256), 0, st, c->nloc, c->t2_tiles ? c->ovr_ptr : c->ovf_ptr,
                               c->t2_tiles ? c->ovr_ent : c->ovf_ent, ab, c->lf, c->ovf_etab, c->ovf_tab, o_ll, o_ell);
        else if (deep)
            hipLaunchKernelGGL((k_ovf_cell_direct<false, false, true>), dim3(g), dim3(256), 0, st, c->nloc, c->ovf_ell_ptr, c->ovf_ell, ab, c->lf,
                               c->ovf_etab, c->ovf_tab, o_ll, o_ell);
        else
            hipLaunchKernelGGL((k_ovf_cell_direct<false, false, false>), dim3(g), dim3(256), 0, st, c->nloc, c->ovf_ell_ptr, c->ovf_ell, ab, c->lf,
                               c->ovf_etab, c->ovf_tab, o_ll, o_ell);
        if (c->ovf_n_tier[0] && !deep)
            hipLaunchKernelGGL((k_ovf_cell_listed<false, false>), dim3(gcap(c->ovf_n_tier[0], 256, 0x7fffffffu)), dim3(256), 0, st,
                               c->ovf_n_tier[0], c->ovf_tier_row[0], c->ovf_tier_ent[0], ab, c->lf, c->ovf_tab, o_ll, o_ell);
        if (c->ovf_n_tier[1]) {
            hipLaunchKernelGGL(k_ovf_listed_values<false>, dim3(gcap(c->ovf_n_tier[1], 256, 0x7fffffffu)), dim3(256), 0, st,
                               c->ovf_n_tier[1], c->ovf_tier_ent[1], ab, c->lf, c->ovf_tier_val, c->ovf_tier_val + c->ovf_n_tier[1]);
            hipLaunchKernelGGL(k_ovf_listed_add<false>, dim3(gcap(c->ovf_n_tier[1], 256, 0x7fffffffu)), dim3(256), 0, st, c->ovf_n_tier[1],
                               c->ovf_tier_row[1], c->ovf_tier_val, c->ovf_tier_val + c->ovf_n_tier[1], o_ll, o_ell);
        }
    }
}
// locus side: the per-locus cumulative-log tables that k_locus_finalize evaluates the overflow entries' log-pmfs from, on stream `st`
static void launch_overflow_locus_values(cellector_ctx *c, hipStream_t st, const double2 *ab)
{
    if (c->t2 && !c->ovf_deep) {  // tier 2 needs nothing here (k_t2_tables is the cell side's first kernel); the few other entries:
        if (c->ovx_n)
            hipLaunchKernelGGL(k_ovx_values, dim3(gcap(c->ovx_n, 256, 0x7fffffffu)), dim3(256), 0, st, c->ovx_n, c->ovx_locus, c->ovx_ent, ab,
                               c->lf, c->ovx_lp);
        return;
    }
    if (c->t2)  // deep: the pairs' log-pmfs for the locus finalize's counts (the cell side does not use the table)
        hipLaunchKernelGGL(k_t2_tables<false>, dim3(gcap(c->t2_np, 256, 0x7fffffffu)), dim3(256), 0, st, c->t2_np, c->t2_plist, c->t2_ns,
                           c->t2_slist, gcap(c->t2_np, 256, 0x7fffffffu), ab, c->lf, c->tab2);
    hipLaunchKernelGGL(k_ovf_tables, dim3(gcap(c->L * 3, 256, 0x7fffffffu)), dim3(256), 0, st, c->L, ab, c->ovf_nmask, c->ovf_tab);
    if (!c->ovf_deep)  // shallow coverage: the values are stored here, beside the tile kernel, and the finalize reads them
        hipLaunchKernelGGL(k_ovf_values, dim3(gcap(c->ovf_n, 256, 0x7fffffffu)), dim3(256), 0, st, c->ovf_n, c->ovc_locus, c->ovc_ent, ab,
                           c->lf, c->ovf_tab, c->ovf_lp);
}

static bool have_overflow(const cellector_ctx *c) { return c->ovf_n != 0 && c->L != 0 && c->nloc != 0; }

// fork: the side stream starts after everything already queued on the main stream (alpha/beta are ready there)
static cellector_status side_fork(cellector_ctx *c)
{
    if (c->tab_event_valid) {  // the table kernel is the last thing queued on the main stream and its completion has an event
        c->tab_event_valid = false;
        HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_tab, 0));
        return CELLECTOR_OK;
    }
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    return CELLECTOR_OK;
}
// join: the main stream continues after everything queued on the side stream
static cellector_status side_join(cellector_ctx *c)
{
    HIPCHK(c, hipEventRecord(c->ev_join, c->side));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    return CELLECTOR_OK;
}

// the chunk tables of one pass
static cellector_status build_tile_tables(cellector_ctx *c, const double2 *ab, int set, bool expected, bool form_ab = false,
                                          const uint8_t *mask = nullptr, hipEvent_t done = nullptr /*rides on the dispatch*/)
{
    c->tab_event_valid = false;
    const uint64_t tab_elems = (uint64_t)c->t_nj * TAB_ELEMS;
    double *tab = expected ? c->tab + 3 * tab_elems : c->tab + (uint64_t)set * tab_elems;
    const unsigned tgrid = gcap((uint64_t)c->t_nj * T_BL, 64);
    ab_src_t src = {};
    if (form_ab) {  // first kernel of an EM iteration: alpha/beta from the exchanged tallies, counters reset
        const uint64_t L = c->L;
        src.s_alt = c->s_alt; src.s_ref = c->s_ref;
        src.alt_min = c->x_locus + LB_ALT_MIN * L; src.ref_min = c->x_locus + LB_REF_MIN * L;
        src.mask = mask ? mask : c->mask; src.ab_out = c->ab;
        src.xl_counters = c->x_locus + (uint64_t)LB_PLANES * L; src.d_counters = c->d_counters;
        src.tile_work = c->tile_work; src.n_work = 3u * T_GROUPS_MAX;
        c->work_zeroed = true;
    }
    if (done) {  // (an event RECORDED behind the kernel is a barrier packet of its own: ~6 us of idle queue in front of the next kernel)
        if (expected)
            hipExtLaunchKernelGGL(k_build_tables<true>, dim3(tgrid), dim3(64 * TB_PARTS), 0, c->stream, nullptr, done, 0, c->L, c->t_nj, ab,
                                  (const double *)c->lf, tab, src);
        else
            hipExtLaunchKernelGGL(k_build_tables<false>, dim3(tgrid), dim3(64 * TB_PARTS), 0, c->stream, nullptr, done, 0, c->L, c->t_nj, ab,
                                  (const double *)c->lf, tab, src);
        c->tab_event_valid = true;
    } else if (expected)
        hipLaunchKernelGGL(k_build_tables<true>, dim3(tgrid), dim3(64 * TB_PARTS), 0, c->stream, c->L, c->t_nj, ab, c->lf, tab, src);
    else
        hipLaunchKernelGGL(k_build_tables<false>, dim3(tgrid), dim3(64 * TB_PARTS), 0, c->stream, c->L, c->t_nj, ab, c->lf, tab, src);
    if (set == 0) {  // the locus pass of this iteration reads the log-pmfs of the EM pass' table
        c->tab_em = tab;
        c->tab_em_stride = expected ? 2 : 1;
    }
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

// tile kernel of one pass on the main stream (its tables are built)
static cellector_status run_tile_pass(cellector_ctx *c, int set, bool expected)
{
    const uint64_t tab_elems = (uint64_t)c->t_nj * TAB_ELEMS;
    double *tab = expected ? c->tab + 3 * tab_elems : c->tab + (uint64_t)set * tab_elems;
    double *part_ll = c->part + (uint64_t)set * 2 * c->t_groups * c->t_npad;
    double *part_ell = part_ll + (uint64_t)c->t_groups * c->t_npad;
    // several cell blocks per column amortise the table staging; with few blocks (small shard) prefer more columns
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
    int sb = T_SB_MAX;
    while (sb > 2 && (uint64_t)((c->t_nb + sb - 1) / sb) * c->t_groups * 10 < (uint64_t)ncu * 9) sb >>= 1;  // < 90 % of the CUs busy
    const uint32_t n_cols = (c->t_nb + sb - 1) / sb;
    // persistent workgroups: one per CU, an equal number for every chunk group, never more than there are columns
    uint32_t per_group = (uint32_t)ncu / c->t_groups;
    if (per_group < 1) per_group = 1;
    if (per_group > n_cols) per_group = n_cols;
    const dim3 grid(per_group * c->t_groups);
    uint32_t *work = c->tile_work + (size_t)set * T_GROUPS_MAX;
    static_assert(T_GROUPS_MAX == CELLECTOR_TILE_WORK_STRIDE, "k_alpha_beta resets the counters with this stride");
    if (!(set == 0 && c->work_zeroed))  // else: reset by this iteration's k_alpha_beta
        HIPCHK(c, hipMemsetAsync(work, 0, T_GROUPS_MAX * sizeof(uint32_t), c->stream));
    hipEvent_t t_a = nullptr, t_b = nullptr;  // the timer's events ride on the dispatch: no barrier packets around the kernel
    const bool timed = timer_take(c, CELLECTOR_K_TILE_LL, &t_a, &t_b);
#define LAUNCH_TILE(E, S)                                                                                                  \
    do {                                                                                                                   \
        if (timed)                                                                                                         \
            hipExtLaunchKernelGGL((k_tile_ll<E, S>), grid, dim3(T_THREADS), 0, c->stream, t_a, t_b, 0, c->t_nb, c->t_nj, c->t_cpg, \
                                  c->t_groups, n_cols, work, (const uint16_t *)c->thdr, (const uint16_t *)c->tiles,        \
                                  (const double *)tab, c->t_npad, part_ll, part_ell);                                      \
        else                                                                                                               \
            hipLaunchKernelGGL((k_tile_ll<E, S>), grid, dim3(T_THREADS), 0, c->stream, c->t_nb, c->t_nj, c->t_cpg, c->t_groups, \
                               n_cols, work, c->thdr, c->tiles, tab, c->t_npad, part_ll, part_ell);                        \
    } while (0)
    if (expected) {
        if (sb == 4) LAUNCH_TILE(true, 4); else LAUNCH_TILE(true, 2);
    } else {
        if (sb == 4) LAUNCH_TILE(false, 4); else LAUNCH_TILE(false, 2);
    }
#undef LAUNCH_TILE
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

// first half of the cell pass: tables, the tile kernel and the overflow kernels beside it (results in scratch arrays)
static cellector_status cell_pass_launch(cellector_ctx *c, const double2 *ab, bool for_em)
{
    timer_begin(c, CELLECTOR_K_CELL_LL);
    const bool ovf = have_overflow(c);
    // The overflow entries' cell side runs on the side stream BESIDE the tile kernel: in the few wave slots the persistent
    // tile workgroups leave it is several times slower than alone, but it ends well inside the tile kernel's time.  In an
    // EM iteration the locus side's values follow on the side stream once the tile kernel is done; the locus finalize
    // waits for them (tiled_locus_pass).  (overlap 0: everything in the main stream.)
    if (for_em && c->tables_prebuilt) c->tables_prebuilt = false;  // built ahead by the previous iteration's em_finish
    else CHK(build_tile_tables(c, ab, 0, c->compute_expected, for_em));
    // (deep coverage, ovf_deep: the same arrangement with the unthrottled 16-lanes-per-row kernel — it fills the tile kernel's
    //  idle issue slots while that runs and has the machine to itself afterwards: 7.8 ms per iteration at 1M x 200k deep
    //  against 8.3 with the two one after the other)
    if (ovf && c->overlap) {
        // the tile kernel is launched FIRST: with the tables built ahead the queue is empty when the host gets here, and
        // every launch ahead of it (five on the side stream) would be ~10 us of idle GPU
        CHK(side_fork(c));
        CHK(run_tile_pass(c, 0, c->compute_expected));
        CHK(t2_tiles_pass(c, ab, 0, c->compute_expected));
        launch_overflow_cell(c, c->side, ab, 0, c->compute_expected);
        HIPCHK(c, hipEventRecord(c->ev_join, c->side));
        if (for_em && c->overlap == 1) {  // the locus side's values right behind the cell side, beside the tile kernel
            launch_overflow_locus_values(c, c->side, ab);  // (they may end after it: only the locus finalize waits for them)
            HIPCHK(c, hipEventRecord(c->ev_join2, c->side));
            c->ovf_locus_pending = true;
        }
        if (for_em && c->overlap == 2) {  // ... or once the tile kernel is done, beside the order statistics
            HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
            launch_overflow_locus_values(c, c->side, ab);
            HIPCHK(c, hipEventRecord(c->ev_join2, c->side));
            c->ovf_locus_pending = true;
        }
        c->cell_join_pending = true;
    } else {
        if (ovf) {
            launch_overflow_cell(c, c->stream, ab, 0, c->compute_expected);
            if (for_em) launch_overflow_locus_values(c, c->stream, ab);
        }
        CHK(run_tile_pass(c, 0, c->compute_expected));
        CHK(t2_tiles_pass(c, ab, 0, c->compute_expected));
    }
    return CELLECTOR_OK;
}

cellector_status tiled_cell_pass(cellector_ctx *c, const double2 *ab, double *norm_out, bool for_em)
{
    if (c->nloc == 0) return CELLECTOR_OK;
    const bool ovf = have_overflow(c);
    // (Measured and dropped: em_finish queueing this first half for the NEXT iteration right behind its table kernel, so that the
    //  tile kernel starts without waiting for the host to read the summary and launch — no gain at 10^6 cells, none on a 125k-cell
    //  shard: 0.50 vs 0.49 ms.  The ~15 us between the table kernel and the tile kernel are not the host's.)
    CHK(cell_pass_launch(c, ab, for_em));
    if (c->cell_join_pending) {  // the overflow entries' cell-side sums (side stream)
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        c->cell_join_pending = false;
    }
    CHK(t2_tiles_add(c, 0, c->compute_expected));
    double *part_ll = c->part, *part_ell = c->part + (uint64_t)c->t_groups * c->t_npad;
    const double *o_ll = ovf ? c->ovf_sum : nullptr, *o_ell = ovf ? c->ovf_sum + c->nloc : nullptr;
    const unsigned grid = gcap((c->nloc + 1) / 2, 256, 0x7fffffffu);
    if (c->compute_expected)
        hipLaunchKernelGGL(k_cell_finalize<true>, dim3(grid), dim3(256), 0, c->stream, c->nloc, c->t_groups, c->t_npad, part_ll,
                           part_ell, o_ll, o_ell, c->csr_ptr, c->masked_cnt, c->ll, c->ell, c->nloci, norm_out);
    else
        hipLaunchKernelGGL(k_cell_finalize<false>, dim3(grid), dim3(256), 0, c->stream, c->nloc, c->t_groups, c->t_npad, part_ll,
                           part_ell, o_ll, o_ell, c->csr_ptr, c->masked_cnt, c->ll, c->ell, c->nloci, norm_out);
    timer_end(c, CELLECTOR_K_CELL_LL);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

cellector_status tiled_locus_pass(cellector_ctx *c)
{
    if (c->L == 0) return CELLECTOR_OK;
    timer_begin(c, CELLECTOR_K_LOCUS_STATS);
    // (k_locus_finalize's loop over the overflow entries, a third of its time at 10^6 cells, was tried as a kernel of its
    //  own on the side stream beside k_minority_ranges: both stream scattered lines, the pair took as long as one after
    //  the other)
    const uint32_t words = (uint32_t)((c->nloc + 31) / 32);
    // (k_flag wrote the exclusion bitmask flag_bits along with the flags)
    const size_t lds = (size_t)words * 4;
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
    unsigned grid = (unsigned)ncu;
    const uint64_t need = (c->L + LS_THREADS / 64 - 1) / (LS_THREADS / 64);
    if (grid > need) grid = (unsigned)(need ? need : 1);
#define LAUNCH_LS(INLDS, EBV, GRID, LDSB)                                                                              \
    hipLaunchKernelGGL((k_locus_stats2<INLDS, EBV>), dim3(GRID), dim3(LS_THREADS), LDSB, c->stream, c->L, words, c->c4_ptr, \
                       c->c4_ent, c->flag_bits, c->hist_min, c->locus_mode, c->nloc, c->d_counters + DC_N_MIN, c->lr_sub)
    // (a forced minority-driven form, locus_mode 2, still launches the streamed kernel: it returns at once unless the
    //  exclusion set is too large for that form's 16-bit counters, the one case the device predicate overrides the option)
    if (lds <= 128 * 1024) {
        if (c->c4_bits == 24) {
            // the whole 2^20-cell bitmask: a padding entry (cell = all ones) then reads inside the allocation
            const int lb = 128 * 1024;
            HIPCHK(c, hipFuncSetAttribute((const void *)k_locus_stats2<true, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
            LAUNCH_LS(true, 24, grid, lb);
        } else {
            const int lb = (int)(lds ? lds : 4);
            HIPCHK(c, hipFuncSetAttribute((const void *)k_locus_stats2<true, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
            LAUNCH_LS(true, 32, grid, lb);
        }
    } else {
        LAUNCH_LS(false, 32, grid * 2, 4);  // more than 2^20 cells per shard: 32-bit entries, bitmask read from L2
    }
#undef LAUNCH_LS
    if (c->nloc == 0)  // an empty shard: no kernel fills the planes
        HIPCHK(c, hipMemsetAsync(c->hist_min, 0, (uint64_t)c->lr_sub * c->L * 16 * sizeof(uint32_t), c->stream));
    if (c->locus_mode != 1 && c->nloc) {
        const uint32_t R = (uint32_t)((c->L + LR_LOCI - 1) / LR_LOCI);
        // capacity of the transposed offsets: the largest exclusion set the automatic choice hands to this form; the
        // forced form (tests, ablations) may need all cells
        const uint64_t want = c->locus_mode == 2 ? c->nloc : c->nloc * LM_NUM / LM_DEN + LT_CELLS;
        if (c->mroff_cap < want) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            dev_free(c->mroff); dev_free(c->mbeg);
            c->mroff_cap = (want + 63) & ~63ull;
            CHK(dev_alloc(c, &c->mroff, (uint64_t)(R + 1) * c->mroff_cap));
            CHK(dev_alloc(c, &c->mbeg, c->mroff_cap));
        }
        const size_t lds_t = (size_t)LT_CELLS * (R + 2) * sizeof(uint32_t);
        if (lds_t > 64 * 1024)
            HIPCHK(c, hipFuncSetAttribute((const void *)k_minority_offsets, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
        hipLaunchKernelGGL(k_minority_offsets, dim3(gcap(want, LT_CELLS, 0x7fffffffu)), dim3(256), lds_t, c->stream, c->locus_mode,
                           c->nloc, c->lr_sub, R, c->mroff_cap, c->d_counters + DC_N_MIN, c->minlist, c->csr_ptr, c->roff, c->mroff, c->mbeg);
        hipLaunchKernelGGL(k_minority_ranges, dim3(R * c->lr_sub), dim3(LR_THREADS), 0, c->stream, c->locus_mode, c->nloc, c->L, R,
                           c->lr_sub, c->mroff_cap, c->d_counters + DC_N_MIN, c->mroff, c->mbeg, c->c4r, c->hist_min);
    }
    // the excluded cells' tier-2 entries per (locus, pair).  (On the side stream it does NOT run beside k_minority_ranges, whose
    // sixteen 128-register waves per CU leave no room: it started when that kernel ended, two event gaps later.)
    if (c->t2 && c->nloc)
        hipLaunchKernelGGL(k_t2_minority, dim3(gcap(c->nloc / LM_DEN + 1, 256 / LF_LANES, 8192)), dim3(256), 0, c->stream, c->d_counters + DC_N_MIN,
                           c->minlist, c->ovf_ptr, c->ovf_ent, c->cnt2);
    if (c->ovf_locus_pending) {  // the overflow entries' log-pmfs of this iteration (side stream)
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join2, 0));
        c->ovf_locus_pending = false;
    }
    // the entries the finalize walks one by one: all overflow entries, or (tier 2) those outside tier 2
    const uint64_t *w_ptr = c->t2 ? (c->ovx_n ? c->ovx_ptr : nullptr) : (c->ovf_n ? c->ovc_ptr : nullptr);
    const uint64_t *w_ent = c->t2 ? c->ovx_ent : c->ovc_ent;
    const double *w_lp = c->t2 ? c->ovx_lp : c->ovf_lp;
#define LAUNCH_LF(INL)                                                                                                             \
    hipLaunchKernelGGL(k_locus_finalize<INL>, dim3(gcap(c->L * LF_LANES, 256, 0x7fffffffu)), dim3(256), 0, c->stream, c->L, c->locus_mode, \
                       c->nloc, c->lr_sub, c->d_counters + DC_N_MIN, c->hist_min, c->flag_bits, c->hist_all, c->tab_em,            \
                       (uint32_t)c->tab_em_stride, c->mask, w_ptr, w_ent, w_lp, c->ovf_tab, c->lf, c->ab, c->x_locus,              \
                       c->t2 ? c->cnt2 : (uint32_t *)nullptr, c->hist_all2, c->t2_pmask, c->tab2,                                    \
                       c->filter_fused ? c->mask_next : (uint8_t *)nullptr, c->d_counters)
    // (a ctx that holds all cells and has no communicator: nothing is exchanged between the finalize and the filter)
    c->filter_fused = !comm_active(c->comm) && c->nloc == c->total_cells && c->fuse_filter;
    if (c->ovf_deep) LAUNCH_LF(true); else LAUNCH_LF(false);
#undef LAUNCH_LF
    timer_end(c, CELLECTOR_K_LOCUS_STATS);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

// The NEXT iteration's first kernel (alpha/beta + tables + counter reset), queued by em_finish behind the locus filter and
// the summary kernel, i.e. before the host waits for the summary: it then runs while the host wakes up and decides.  It
// reads what the next em_begin would read: the exchanged tallies and the filtered mask (mask_next, about to become mask).
cellector_status tiled_prebuild_tables(cellector_ctx *c)
{
    if (c->nloc == 0 || c->L == 0) return CELLECTOR_OK;
    CHK(build_tile_tables(c, c->ab, 0, c->compute_expected, true, c->mask_next, c->ev_tab));
    c->tables_prebuilt = true;
    c->prebuilt_expected = c->compute_expected;
    return CELLECTOR_OK;
}

// called after the locus filter with mask = this iteration's mask, mask_next = filtered mask
cellector_status tiled_masked_update(cellector_ctx *c)
{
    if (c->L == 0) return CELLECTOR_OK;
    if (c->c4_bits == 24)
        hipLaunchKernelGGL(k_masked_update<24>, dim3(gcap(c->L, 4)), dim3(256), 0, c->stream, c->L, c->mask, c->mask_next,
                           c->c4_ptr, c->c4_ent, c->ovc_ptr, c->ovc_ent, c->masked_cnt);
    else
        hipLaunchKernelGGL(k_masked_update<32>, dim3(gcap(c->L, 4)), dim3(256), 0, c->stream, c->L, c->mask, c->mask_next,
                           c->c4_ptr, c->c4_ent, c->ovc_ptr, c->ovc_ent, c->masked_cnt);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

cellector_status tiled_posteriors(cellector_ctx *c, double mf0, double lp_min, double lp_maj, double lp_dbl)
{
    c->tables_prebuilt = false;  // the posterior passes rebuild table set 0 and use its column counters
    if (c->cell_join_pending) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        c->cell_join_pending = false;
    }
    c->work_zeroed = false;
    const uint64_t L = c->L;
    if (L)
        hipLaunchKernelGGL(k_ab_posterior3, dim3(gcap(L, 256)), dim3(256), 0, c->stream, L, c->s_alt, c->s_ref,
                           c->x_locus + LB_ALT_MIN * L, c->x_locus + LB_REF_MIN * L, mf0, c->ab3, c->ab6);
    HIPCHK(c, hipGetLastError());
    if (c->nloc == 0) return CELLECTOR_OK;
    timer_begin(c, CELLECTOR_K_POSTERIOR);
    const bool ovf = have_overflow(c);
    for (int set = 0; set < 3; set++) CHK(build_tile_tables(c, c->ab3 + (uint64_t)set * L, set, false));
    if (ovf && c->overlap) {
        CHK(side_fork(c));
        for (int set = 0; set < 3; set++) launch_overflow_cell(c, c->side, c->ab3 + (uint64_t)set * L, set, false);
        for (int set = 0; set < 3; set++) CHK(run_tile_pass(c, set, false));
        for (int set = 0; set < 3; set++) CHK(t2_tiles_pass(c, c->ab3 + (uint64_t)set * L, set, false));
        CHK(side_join(c));
    } else {
        if (ovf)
            for (int set = 0; set < 3; set++) launch_overflow_cell(c, c->stream, c->ab3 + (uint64_t)set * L, set, false);
        for (int set = 0; set < 3; set++) CHK(run_tile_pass(c, set, false));
        for (int set = 0; set < 3; set++) CHK(t2_tiles_pass(c, c->ab3 + (uint64_t)set * L, set, false));
    }
    for (int set = 0; set < 3; set++) CHK(t2_tiles_add(c, set, false));
    hipLaunchKernelGGL(k_posterior_finalize, dim3(gcap(c->nloc, 256, 0x7fffffffu)), dim3(256), 0, c->stream, c->nloc,
                       ovf ? c->ovf_sum : (const double *)nullptr, c->t_groups, c->t_npad, c->part, lp_min, lp_maj, lp_dbl, c->post);
    timer_end(c, CELLECTOR_K_POSTERIOR);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}
