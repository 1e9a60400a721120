// Deterministic synthetic vartrix-like (locus x cell) ref/alt matrix generated on the device, in the
// file order vartrix/combiner use (sorted by locus, then cell).  All-integer definition so that
// cellector_amd/synth.py reproduces it bit for bit on the host:
//
//   mix(z)        = splitmix64 finaliser
//   present(l,c)  = (mix(S_lc ^ (l<<32 | c)) >> 40) < round(density * 2^24)             c = GLOBAL cell id
//   n(l,c)        = 1 + #{k>=1 : low32(h2) < T_k},  h2 = mix(h1 + GOLD)   (1 + Geometric(1 - q)); default q = 0.3:
//                   T_k = floor(0.3^k * 2^32) (the table below); option "synth_continue_pct" = p != 30 (deeper coverage,
//                   benchmarks of the count distribution): T_0 = 2^32, T_k = floor(T_{k-1} * p / 100), k < 64
//   class(c)      = minority if (mix(S_min ^ c) >> 40) < round(minority_fraction * 2^24)
//                   doublet  if (mix(S_dbl ^ c) >> 40) < round(doublet_fraction * 2^24)  (takes precedence)
//   genotype(l,S) = v = mix(S ^ l) >> 48 : v < 32768 -> 0/0, v < 52428 -> 0/1, else 1/1  (p = .5/.3/.2)
//   af16          = {655, 32768, 64881}[genotype]  (= .01, .5, .99 in 1/65536); doublet: mean of both
//   alt(l,c)      = #{i < n : 16-bit field i of h3 < af16}, h3 = mix(h2 + GOLD), rehashed every 4 fields
//   ref           = n - alt
#include "ctx.h"

#define SY_BLOCK 256
#define SY_CELLS_PER_THREAD 64
#define SY_CHUNK (SY_BLOCK * SY_CELLS_PER_THREAD)  // 16384 cells per tile
#define GOLD 0x9E3779B97F4A7C15ull

#define SY_GEOM_MAX 64
struct SynthParams {
    uint64_t s_lc, s_min, s_dbl, s_gmaj, s_gmin;
    uint32_t thr_density, thr_min, thr_dbl;
    uint32_t nchunks;
    uint64_t cell_begin, nloc;
    uint32_t n_geom;       // thresholds in geom[1..n_geom-1]
    uint32_t geom[SY_GEOM_MAX];
};

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

static const uint32_t SY_GEOM[19] = {0u, 1288490188u, 386547056u, 115964116u, 34789235u, 10436770u, 3131031u,
                                     939309u, 281792u, 84537u, 25361u, 7608u, 2282u, 684u, 205u, 61u, 18u, 5u, 1u};

__device__ __forceinline__ uint32_t genotype_af16(uint64_t s, uint32_t l)
{
    const uint32_t v = (uint32_t)(mix64(s ^ (uint64_t)l) >> 48);
    return v < 32768u ? 655u : (v < 52428u ? 32768u : 64881u);
}

__device__ __forceinline__ uint64_t presence_mask(const SynthParams &p, uint32_t l, uint64_t c0_local, uint64_t *h_first)
{
    uint64_t m = 0;
    for (int b = 0; b < SY_CELLS_PER_THREAD; b++) {
        const uint64_t cl = c0_local + b;
        if (cl >= p.nloc) break;
        const uint64_t h = mix64(p.s_lc ^ (((uint64_t)l << 32) | (p.cell_begin + cl)));
        if ((uint32_t)(h >> 40) < p.thr_density) m |= 1ull << b;
    }
    (void)h_first;
    return m;
}

__global__ __launch_bounds__(SY_BLOCK) void k_synth_count(SynthParams p, uint64_t tile_base,
                                                          uint64_t *__restrict__ tile_cnt)
{
    const uint64_t tile = tile_base + blockIdx.x;
    const uint32_t l = (uint32_t)(tile / p.nchunks), chunk = (uint32_t)(tile % p.nchunks);
    const uint64_t c0 = (uint64_t)chunk * SY_CHUNK + (uint64_t)threadIdx.x * SY_CELLS_PER_THREAD;
    const uint64_t m = c0 < p.nloc ? presence_mask(p, l, c0, nullptr) : 0;
    uint32_t cnt = (uint32_t)__popcll(m);
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    __shared__ uint32_t w[SY_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[tile] = (uint64_t)w[0] + w[1] + w[2] + w[3];
}

__global__ __launch_bounds__(SY_BLOCK) void k_synth_fill(SynthParams p, uint64_t tile_base,
                                                         const uint64_t *__restrict__ tile_off,
                                                         uint32_t *__restrict__ o_locus, uint32_t *__restrict__ o_cell,
                                                         uint16_t *__restrict__ o_alt, uint16_t *__restrict__ o_ref)
{
    const uint64_t tile = tile_base + blockIdx.x;
    const uint32_t l = (uint32_t)(tile / p.nchunks), chunk = (uint32_t)(tile % p.nchunks);
    const uint64_t c0 = (uint64_t)chunk * SY_CHUNK + (uint64_t)threadIdx.x * SY_CELLS_PER_THREAD;
    uint64_t m = c0 < p.nloc ? presence_mask(p, l, c0, nullptr) : 0;
    const uint32_t cnt = (uint32_t)__popcll(m);
    // exclusive prefix of cnt over the block (cell order == thread order)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = cnt;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    __shared__ uint32_t w[SY_BLOCK / 64];
    if (lane == 63) w[wv] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < wv; k++) woff += w[k];
    uint64_t pos = tile_off[tile] + woff + inc - cnt;
    if (!m) return;
    const uint32_t af_maj = genotype_af16(p.s_gmaj, l), af_min = genotype_af16(p.s_gmin, l);
    while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint64_t cl = c0 + b, cg = p.cell_begin + cl;
        const uint64_t h1 = mix64(p.s_lc ^ (((uint64_t)l << 32) | cg));
        const uint64_t h2 = mix64(h1 + GOLD);
        const uint32_t u = (uint32_t)h2;
        uint32_t n = 1;
        for (uint32_t k = 1; k < p.n_geom; k++) {
            if (u < p.geom[k]) n++; else break;
        }
        uint32_t af = ((uint32_t)(mix64(p.s_min ^ cg) >> 40) < p.thr_min) ? af_min : af_maj;
        if ((uint32_t)(mix64(p.s_dbl ^ cg) >> 40) < p.thr_dbl) af = (af_maj + af_min) >> 1;
        uint64_t h3 = mix64(h2 + GOLD);
        uint32_t alt = 0;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t field = (uint32_t)(h3 >> (16 * (i & 3))) & 0xffffu;
            alt += field < af ? 1u : 0u;
            if ((i & 3) == 3) h3 = mix64(h3 + GOLD);
        }
        o_locus[pos] = l; o_cell[pos] = (uint32_t)cl;
        o_alt[pos] = (uint16_t)alt; o_ref[pos] = (uint16_t)(n - alt);
        pos++;
    }
}

static uint64_t host_mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

cellector_status synth_generate(cellector_ctx *c, double density, uint64_t seed, double minority_fraction,
                                double doublet_fraction)
{
    if (!(density > 0.0 && density <= 1.0) || minority_fraction < 0.0 || minority_fraction > 1.0 ||
        doublet_fraction < 0.0 || doublet_fraction > 1.0)
        return ctx_fail(c, CELLECTOR_EINVAL, "synthetic: density/fractions out of range");
    if (c->total_loci == 0 || c->total_loci > 0xffffffffull || c->nloc == 0)
        return ctx_fail(c, CELLECTOR_EINVAL, "synthetic: unsupported dims");
    SynthParams p;
    const uint64_t base = host_mix64(seed * GOLD + 0x5EEDull);
    p.s_lc = host_mix64(base + 1); p.s_min = host_mix64(base + 2); p.s_dbl = host_mix64(base + 3);
    p.s_gmaj = host_mix64(base + 4); p.s_gmin = host_mix64(base + 5);
    p.thr_density = (uint32_t)(density * 16777216.0 + 0.5);
    p.thr_min = (uint32_t)(minority_fraction * 16777216.0 + 0.5);
    p.thr_dbl = (uint32_t)(doublet_fraction * 16777216.0 + 0.5);
    p.cell_begin = c->cell_begin; p.nloc = c->nloc;
    memset(p.geom, 0, sizeof p.geom);
    if (c->synth_continue_pct == 30) {
        p.n_geom = 19;
        for (int k = 0; k < 19; k++) p.geom[k] = SY_GEOM[k];
    } else {
        p.n_geom = SY_GEOM_MAX;
        uint64_t t = 1ull << 32;
        for (int k = 1; k < SY_GEOM_MAX; k++) {
            t = t * (uint64_t)c->synth_continue_pct / 100u;
            p.geom[k] = (uint32_t)t;
        }
    }
    p.nchunks = (uint32_t)((c->nloc + SY_CHUNK - 1) / SY_CHUNK);
    const uint64_t ntiles = c->total_loci * p.nchunks;
    uint64_t *tiles = nullptr;
    CHK(dev_alloc(c, &tiles, ntiles + 1));
    HIPCHK(c, hipMemsetAsync(tiles + ntiles, 0, 8, c->stream));
    const uint64_t max_grid = 1ull << 30;
    for (uint64_t t0 = 0; t0 < ntiles; t0 += max_grid) {
        const uint64_t g = ntiles - t0 < max_grid ? ntiles - t0 : max_grid;
        hipLaunchKernelGGL(k_synth_count, dim3((unsigned)g), dim3(SY_BLOCK), 0, c->stream, p, t0, tiles);
    }
    HIPCHK(c, hipGetLastError());
    uint64_t n = 0;
    CHK(dev_exclusive_scan_u64(c, tiles, ntiles + 1, &n));
    c->coo_n = n;
    c->coo_sorted = true;
    CHK(dev_alloc(c, &c->coo_locus, n)); CHK(dev_alloc(c, &c->coo_cell, n));
    CHK(dev_alloc(c, &c->coo_alt, n)); CHK(dev_alloc(c, &c->coo_ref, n));
    for (uint64_t t0 = 0; t0 < ntiles; t0 += max_grid) {
        const uint64_t g = ntiles - t0 < max_grid ? ntiles - t0 : max_grid;
        hipLaunchKernelGGL(k_synth_fill, dim3((unsigned)g), dim3(SY_BLOCK), 0, c->stream, p, t0, tiles, c->coo_locus,
                           c->coo_cell, c->coo_alt, c->coo_ref);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dev_free(tiles);
    return CELLECTOR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Benchmark utility (not a reference seam): the staged matrix as vartrix-style MatrixMarket text, `locus cell count` lines,
// 1-based, file order — so that BASELINE-sized inputs for the text path can be produced in seconds (tools/e2e_bench.py).
// Lines are formatted on the device in chunks: lengths -> exclusive scan -> characters, then copied out and written.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dec_digits(uint32_t v)
{
    uint32_t d = 1;
    while (v >= 10u) { v /= 10u; d++; }
    return d;
}
__device__ __forceinline__ uint8_t *put_dec(uint8_t *p, uint32_t v, uint32_t d)
{
    for (uint32_t k = d; k-- > 0;) { p[k] = (uint8_t)('0' + v % 10u); v /= 10u; }
    return p + d;
}
template <bool FILL>
__global__ __launch_bounds__(256) void k_mtx_lines(uint64_t n, uint64_t cell_begin, const uint32_t *__restrict__ locus,
                                                   const uint32_t *__restrict__ cell, const uint16_t *__restrict__ val,
                                                   uint64_t *__restrict__ len_or_off, uint8_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = locus[i] + 1u, b = (uint32_t)(cell_begin + cell[i]) + 1u, v = val[i];
    const uint32_t da = dec_digits(a), db = dec_digits(b), dv = dec_digits(v);
    if (!FILL) {
        len_or_off[i] = da + db + dv + 3u;
        return;
    }
    uint8_t *p = out + len_or_off[i];
    p = put_dec(p, a, da); *p++ = ' ';
    p = put_dec(p, b, db); *p++ = ' ';
    p = put_dec(p, v, dv); *p = '\n';
}

cellector_status synth_write_mtx(cellector_ctx *c, const char *alt_path, const char *ref_path)
{
    if (!c->coo_locus && c->coo_n) return ctx_fail(c, CELLECTOR_EINVAL, "write_staged_mtx: no staged matrix (option keep_coo=1)");
    const uint64_t n = c->coo_n, CH = 1ull << 25;  // 32M lines per chunk: at most 26 bytes each
    FILE *f[2] = {fopen(alt_path, "wb"), fopen(ref_path, "wb")};
    uint64_t *off = nullptr;
    uint8_t *dbuf = nullptr, *hbuf = nullptr;
    cellector_status st = CELLECTOR_OK;
    if (!f[0] || !f[1]) st = ctx_fail(c, CELLECTOR_EIO, "write_staged_mtx: cannot create %s", f[0] ? ref_path : alt_path);
    if (st == CELLECTOR_OK) st = dev_alloc(c, &off, CH + 1);
    if (st == CELLECTOR_OK) st = dev_alloc(c, &dbuf, CH * 26);
    if (st == CELLECTOR_OK && hipHostMalloc((void **)&hbuf, CH * 26) != hipSuccess) st = ctx_fail(c, CELLECTOR_ENOMEM, "pinned buffer");
    for (int k = 0; k < 2 && st == CELLECTOR_OK; k++)
        fprintf(f[k], "%%%%MatrixMarket matrix coordinate real general\n%% written by sprs\n%llu %llu %llu\n",
                (unsigned long long)c->total_loci, (unsigned long long)c->total_cells, (unsigned long long)n);
    for (uint64_t i0 = 0; i0 < n && st == CELLECTOR_OK; i0 += CH) {
        const uint64_t m = n - i0 < CH ? n - i0 : CH;
        const unsigned g = (unsigned)((m + 255) / 256);
        for (int k = 0; k < 2 && st == CELLECTOR_OK; k++) {
            const uint16_t *val = (k == 0 ? c->coo_alt : c->coo_ref) + i0;
            uint64_t bytes = 0;
            hipError_t e = hipMemsetAsync(off + m, 0, 8, c->stream);
            hipLaunchKernelGGL(k_mtx_lines<false>, dim3(g), dim3(256), 0, c->stream, m, c->cell_begin, c->coo_locus + i0,
                               c->coo_cell + i0, val, off, (uint8_t *)nullptr);
            if (e == hipSuccess) st = dev_exclusive_scan_u64(c, off, m + 1, &bytes);
            if (st != CELLECTOR_OK) break;
            hipLaunchKernelGGL(k_mtx_lines<true>, dim3(g), dim3(256), 0, c->stream, m, c->cell_begin, c->coo_locus + i0,
                               c->coo_cell + i0, val, off, dbuf);
            if (e == hipSuccess) e = hipMemcpyAsync(hbuf, dbuf, bytes, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { st = ctx_fail(c, CELLECTOR_EDEVICE, "write_staged_mtx: %s", hipGetErrorString(e)); break; }
            if (fwrite(hbuf, 1, bytes, f[k]) != bytes) st = ctx_fail(c, CELLECTOR_EIO, "write_staged_mtx: short write");
        }
    }
    for (int k = 0; k < 2; k++)
        if (f[k] && fclose(f[k]) != 0 && st == CELLECTOR_OK) st = ctx_fail(c, CELLECTOR_EIO, "write_staged_mtx: close failed");
    dev_free(off); dev_free(dbuf);
    if (hbuf) (void)hipHostFree(hbuf);
    return st;
}
