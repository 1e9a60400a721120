// Exact order statistics of N f64 keys on the device (statrs Data::median / Data::quantile need the
// values at up to six ranks: main.rs:324-327).
//
// MSB-first radix select in four kernels, all SEL_T target ranks refined together, no host round trip:
//   k_sel_top     histogram of the top 12 bits (sign + exponent) of every key: a handful of bins in practice, so a wave
//                 whose keys agree votes with one LDS atomic;
//   k_sel_mid     the step on that histogram (wave 0 of every block, redundantly: a kernel of its own per step costs more
//                 in launch gaps than the re-read of a 16 KB histogram), then the histogram of the next 10 bits of the
//                 keys that carry a target's 12-bit prefix (targets that share a prefix share a histogram);
//   k_sel_gather  the step on those histograms, then the keys that carry a target's 22-bit prefix are appended to a list
//                 (N / 2000 of them per target when the mantissa bits are spread; all of them in the worst case);
//   k_sel_finish  one workgroup: the remaining 42 bits over the list (11 bits, then the few survivors ranked by
//                 counting), then the quartile / threshold arithmetic.
// Few workgroups on purpose (SEL_GRID): a workgroup's histogram goes to the global one with one atomic per non-empty
// bin, and atomics to the same few cache lines run one after the other — 1024 workgroups x 256 bins took 40 us.
// The results stay in device memory: the flagging kernel reads the threshold there.  The histograms are zero when a
// context is created and k_sel_finish leaves them zero.
#include "ctx.h"

#define SEL_THREADS 1024
#define SEL_GRID 128
#define SEL_UNROLL 8  // keys per thread in flight (10^6 keys: one round)
#define SEL_GATHER_ROUND 4  // ... in the gathering pass: a round's matches wait in 32 KB of LDS
#define SEL_NB0 4096  // bins of bits 63..52
#define SEL_NB1 1024  // bins of bits 51..42
#define SEL_SH0 52
#define SEL_SH1 42

__device__ __forceinline__ uint64_t key_of(double x)
{
    uint64_t u = (uint64_t)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);  // monotone: order of u == numeric order of x
}
__device__ __forceinline__ double value_of(uint64_t k)
{
    uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

struct sel_ranks_t { uint64_t r[SEL_T]; };

// one wave, one histogram of NB bins (lane owns NB/64 consecutive ones): lane sums and their inclusive scan
template <int NB>
__device__ __forceinline__ void sel_scan(int lane, const uint32_t *hist, uint64_t &lsum, uint64_t &inc)
{
    constexpr int PER = NB / 64;
    static_assert(PER % 4 == 0, "uint4 reads");
    const uint4 *mine = reinterpret_cast<const uint4 *>(hist + lane * PER);
    uint64_t s = 0;
#pragma unroll 4
    for (int i = 0; i < PER / 4; i++) {
        const uint4 h = mine[i];
        s += (uint64_t)h.x + h.y + h.z + h.w;
    }
    lsum = s;
    uint64_t v = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = (uint64_t)__shfl_up((long long)v, off, 64);
        if (lane >= off) v += o;
    }
    inc = v;
}
// ... and the bin that holds 0-based rank r, with the number of keys below that bin (valid in every lane).  Up to 32 bins
// per lane are fetched together and walked in registers instead of one dependent load per bin.
template <int NB>
__device__ __forceinline__ void sel_pick(int lane, const uint32_t *hist, uint64_t lsum, uint64_t inc, uint64_t r, uint32_t &bin,
                                         uint64_t &below)
{
    constexpr int PER = NB / 64;
    const uint64_t exc = inc - lsum;
    // the lane whose bins contain the rank (the last lane if counts are short: defensive, cannot happen)
    const bool mine = (exc <= r && r < inc) || (lane == 63 && r >= inc);
    uint32_t d = (uint32_t)(lane * PER);
    uint64_t cum = exc;
    if constexpr (PER <= 32) {
        uint32_t hs[PER];
        const uint4 *src4 = reinterpret_cast<const uint4 *>(hist + lane * PER);
#pragma unroll
        for (int i = 0; i < PER / 4; i++) {
            const uint4 q = src4[i];
            hs[4 * i] = q.x; hs[4 * i + 1] = q.y; hs[4 * i + 2] = q.z; hs[4 * i + 3] = q.w;
        }
        bool go = mine;
#pragma unroll
        for (int i = 0; i < PER - 1; i++) {
            go = go && cum + hs[i] <= r;
            if (go) { cum += hs[i]; d++; }
        }
    } else if (mine) {  // (64 bins in registers for six targets spill: the one lane walks memory)
        for (int i = 0; i < PER - 1; i++) {
            const uint32_t h = hist[lane * PER + i];
            if (cum + h > r) break;
            cum += h;
            d++;
        }
    }
    const unsigned long long who = __ballot(mine);
    const int src = __ffsll((long long)who) - 1;
    bin = (uint32_t)__shfl((int)d, src, 64);
    below = (uint64_t)__shfl((long long)cum, src, 64);
}

// targets with equal prefixes share the histogram of the first of them
__device__ __forceinline__ void sel_leaders(const uint64_t prefix[SEL_T], int leader[SEL_T])
{
#pragma unroll
    for (int t = 0; t < SEL_T; t++) {
        int ld = t;
#pragma unroll
        for (int u = SEL_T - 1; u >= 0; u--)
            if (u < t && prefix[u] == prefix[t]) ld = u;
        leader[t] = ld;
    }
}

__global__ __launch_bounds__(SEL_THREADS) void k_sel_top(const double *__restrict__ keys, uint64_t n, uint32_t *__restrict__ hist0)
{
    __shared__ uint32_t h[SEL_NB0];
    for (int i = threadIdx.x; i < SEL_NB0; i += SEL_THREADS) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t stride = (uint64_t)gridDim.x * SEL_THREADS;
    for (uint64_t base = (uint64_t)blockIdx.x * SEL_THREADS; base < n; base += SEL_UNROLL * stride) {
        uint64_t k[SEL_UNROLL];
        bool ok[SEL_UNROLL];
#pragma unroll
        for (int u = 0; u < SEL_UNROLL; u++) {
            const uint64_t i = base + u * stride + threadIdx.x;
            ok[u] = i < n;
            k[u] = ok[u] ? key_of(keys[i]) : 0;
        }
#pragma unroll
        for (int u = 0; u < SEL_UNROLL; u++) {
            const uint32_t d = (uint32_t)(k[u] >> SEL_SH0);
            const unsigned long long act = __ballot(ok[u]);
            if (!act) continue;  // wave-uniform
            const int src = __ffsll((long long)act) - 1;
            const uint32_t d0 = (uint32_t)__shfl((int)d, src, 64);
            const unsigned long long same = __ballot(ok[u] && d == d0);
            if (same == act) {
                if (lane == src) atomicAdd(&h[d0], (uint32_t)__popcll(act));
            } else if (ok[u]) {
                atomicAdd(&h[d], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_NB0; i += SEL_THREADS) {
        const uint32_t v = h[i];
        if (v) atomicAdd(&hist0[i], v);  // agent scope
    }
}

// state = SEL_T x {prefix (resolved high bits, low bits zero), remaining rank inside that prefix}
__global__ __launch_bounds__(SEL_THREADS) void k_sel_mid(const double *__restrict__ keys, uint64_t n, sel_ranks_t ranks,
                                                         const uint32_t *__restrict__ hist0, uint64_t *__restrict__ state1,
                                                         uint32_t *__restrict__ hist1)
{
    __shared__ uint32_t h[SEL_T][SEL_NB1];
    __shared__ uint32_t top[SEL_T];
    for (int i = threadIdx.x; i < SEL_T * SEL_NB1; i += SEL_THREADS) (&h[0][0])[i] = 0;
    if (threadIdx.x < 64 * SEL_T) {  // a wave per target: the step is on every workgroup's critical path
        const int lane = (int)threadIdx.x & 63, t = (int)threadIdx.x >> 6;
        uint64_t lsum, inc, below;
        uint32_t bin;
        sel_scan<SEL_NB0>(lane, hist0, lsum, inc);
        sel_pick<SEL_NB0>(lane, hist0, lsum, inc, ranks.r[t], bin, below);
        if (lane == 0) {
            top[t] = bin;
            if (blockIdx.x == 0) { state1[2 * t] = (uint64_t)bin << SEL_SH0; state1[2 * t + 1] = ranks.r[t] - below; }
        }
    }
    __syncthreads();
    uint32_t ltop[SEL_T];
    bool lead[SEL_T];
    {
        uint64_t p[SEL_T];
        int ld[SEL_T];
#pragma unroll
        for (int t = 0; t < SEL_T; t++) { ltop[t] = top[t]; p[t] = ltop[t]; }
        sel_leaders(p, ld);
#pragma unroll
        for (int t = 0; t < SEL_T; t++) lead[t] = ld[t] == t;
    }
    const uint64_t stride = (uint64_t)gridDim.x * SEL_THREADS;
    for (uint64_t base = (uint64_t)blockIdx.x * SEL_THREADS; base < n; base += SEL_UNROLL * stride) {
        uint64_t k[SEL_UNROLL];
        bool ok[SEL_UNROLL];
#pragma unroll
        for (int u = 0; u < SEL_UNROLL; u++) {
            const uint64_t i = base + u * stride + threadIdx.x;
            ok[u] = i < n;
            k[u] = ok[u] ? key_of(keys[i]) : 0;
        }
#pragma unroll
        for (int u = 0; u < SEL_UNROLL; u++) {
            if (!ok[u]) continue;
            const uint32_t hi = (uint32_t)(k[u] >> SEL_SH0), d = (uint32_t)(k[u] >> SEL_SH1) & (SEL_NB1 - 1);
#pragma unroll
            for (int t = 0; t < SEL_T; t++)
                if (lead[t] && hi == ltop[t]) atomicAdd(&h[t][d], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_T * SEL_NB1; i += SEL_THREADS) {
        const uint32_t v = (&h[0][0])[i];
        if (v) atomicAdd(&hist1[i], v);
    }
}

__global__ __launch_bounds__(SEL_THREADS) void k_sel_gather(const double *__restrict__ keys, uint64_t n,
                                                            const uint64_t *__restrict__ state1, const uint32_t *__restrict__ hist1,
                                                            uint64_t *__restrict__ state2, uint64_t *__restrict__ list,
                                                            uint32_t *__restrict__ count)
{
    __shared__ uint32_t pre[SEL_T];  // the 22 resolved bits
    __shared__ uint64_t buf[SEL_GATHER_ROUND * SEL_THREADS];
    __shared__ uint32_t lcount, gbase;
    if (threadIdx.x == 0) lcount = 0;
    if (threadIdx.x < 64 * SEL_T) {  // a wave per target
        const int lane = (int)threadIdx.x & 63, t = (int)threadIdx.x >> 6;
        uint64_t p1[SEL_T];
        int ld[SEL_T];
#pragma unroll
        for (int u = 0; u < SEL_T; u++) p1[u] = state1[2 * u];
        sel_leaders(p1, ld);
        int mine = 0;
        uint64_t pt = 0;
#pragma unroll
        for (int u = 0; u < SEL_T; u++)
            if (u == t) { mine = ld[u]; pt = p1[u]; }
        const uint64_t r1 = state1[2 * t + 1];
        const uint32_t *hist = hist1 + mine * SEL_NB1;
        uint64_t lsum, inc, below;
        uint32_t bin;
        sel_scan<SEL_NB1>(lane, hist, lsum, inc);
        sel_pick<SEL_NB1>(lane, hist, lsum, inc, r1, bin, below);
        if (lane == 0) {
            const uint64_t np = pt | ((uint64_t)bin << SEL_SH1);
            pre[t] = (uint32_t)(np >> SEL_SH1);
            if (blockIdx.x == 0) { state2[2 * t] = np; state2[2 * t + 1] = r1 - below; }
        }
    }
    __syncthreads();
    uint32_t lp[SEL_T];
#pragma unroll
    for (int t = 0; t < SEL_T; t++) lp[t] = pre[t];
    // matches of a round (SEL_GATHER_ROUND keys per thread) go to LDS first, then to the list with ONE global atomic per workgroup:
    // returning atomics to one address run at ~14 ns each (one per matching wave took 140 us at 10^4 matches)
    const int lane = threadIdx.x & 63;
    const uint64_t stride = (uint64_t)gridDim.x * SEL_THREADS;
    for (uint64_t base = (uint64_t)blockIdx.x * SEL_THREADS; base < n; base += SEL_GATHER_ROUND * stride) {
        uint64_t k[SEL_GATHER_ROUND];
        bool ok[SEL_GATHER_ROUND];
#pragma unroll
        for (int u = 0; u < SEL_GATHER_ROUND; u++) {
            const uint64_t i = base + u * stride + threadIdx.x;
            ok[u] = i < n;
            k[u] = ok[u] ? key_of(keys[i]) : 0;
        }
#pragma unroll
        for (int u = 0; u < SEL_GATHER_ROUND; u++) {
            const uint32_t hi = (uint32_t)(k[u] >> SEL_SH1);
            bool m = false;
#pragma unroll
            for (int t = 0; t < SEL_T; t++) m |= hi == lp[t];
            m &= ok[u];
            const unsigned long long mm = __ballot(m);
            if (!mm) continue;  // wave-uniform
            const int src = __ffsll((long long)mm) - 1;
            uint32_t at = 0;
            if (lane == src) at = atomicAdd(&lcount, (uint32_t)__popcll(mm));
            at = (uint32_t)__shfl((int)at, src, 64);
            if (m) buf[at + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = k[u];
        }
        __syncthreads();
        const uint32_t got = lcount;
        if (threadIdx.x == 0 && got) gbase = atomicAdd(count, got);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < got; i += SEL_THREADS) list[gbase + i] = buf[i];
        if (threadIdx.x == 0) lcount = 0;
        __syncthreads();
    }
}

// statrs Data::median / quantile and the threshold of main.rs:328-329 from the six order statistics,
// in the reference's operation order: out = {median, iqr, threshold}
struct sel_quart_t { double h1, h3; int64_t hf1, hf3; uint64_t n; double iqr_multiple; };

// one workgroup: the low 42 bits over the gathered keys (the six order statistics -> v[0..5]), then the threshold
// arithmetic -> out[0..2].  The first step takes 11 bits (2048-bin histograms): of ~10^4 gathered keys a handful per
// target survive it, they move to LDS and are ranked by counting.  Keys that do not spread (ties) go on in 7-bit steps,
// from LDS when they fit, from the list otherwise.
#define SEL_FIN_CAP 1024     // survivors kept in LDS
#define SEL_FIN_DIRECT 256   // ... and ranked by counting instead of further steps
#define SEL_FIN_NB 2048      // bins of the first step
#define SEL_FIN_STEPS 6      // bits 41..31, then 7, 7, 7, 7, 3
__device__ __forceinline__ void sel_fin_vote(uint64_t k, int shift, int bits, const uint64_t hi[SEL_T], const bool lead[SEL_T],
                                             uint32_t (*h)[SEL_FIN_NB])
{
    const uint32_t d = (uint32_t)(k >> shift) & ((1u << bits) - 1u);
    const uint64_t kh = k >> (shift + bits);
#pragma unroll
    for (int t = 0; t < SEL_T; t++)
        if (lead[t] && kh == hi[t]) atomicAdd(&h[t][d], 1u);
}
__global__ __launch_bounds__(SEL_THREADS) void k_sel_finish(sel_quart_t q, const uint64_t *__restrict__ state2,
                                                            const uint64_t *__restrict__ list, const uint32_t *count,
                                                            uint32_t *scratch /* holds *count */, double *__restrict__ v,
                                                            double *__restrict__ out)
{
    __shared__ uint32_t h[SEL_T][SEL_FIN_NB];
    __shared__ uint64_t st[2 * SEL_T];
    __shared__ int leader[SEL_T];
    __shared__ uint64_t cache[SEL_FIN_CAP + SEL_T];  // (+ the direct ranking's results)
    __shared__ uint32_t n_cache;
    const uint32_t m = *count;
    // the histograms and the list length are zero again for the next call (the earlier kernels are done with them: this
    // one is the last of the chain; `count` is inside the scratch block and was read above)
    __syncthreads();
    for (int i = threadIdx.x; i < CELLECTOR_SEL_HIST_WORDS; i += SEL_THREADS) scratch[i] = 0;
    bool cached = false;  // workgroup-uniform: the keys still in play are cache[0 .. n_cache)
    // a list of up to 16 keys per thread is read ONCE, all loads in flight together (one workgroup reading the list four
    // keys at a time, twice, was 16 us of load latency); a longer one is streamed in both passes of the first step
    constexpr int FIN_REG = 16;
    const bool in_regs = m <= FIN_REG * SEL_THREADS;
    uint64_t kr[FIN_REG];
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < FIN_REG; u++) {
            const uint32_t i = u * SEL_THREADS + threadIdx.x;
            kr[u] = i < m ? list[i] : 0;
        }
    }
    if (threadIdx.x < 2 * SEL_T) st[threadIdx.x] = state2[threadIdx.x];
    __syncthreads();
    int top = SEL_SH1;  // bits [top, 64) are resolved
    for (int step = 0; step < SEL_FIN_STEPS; step++) {
        const int bits = step == 0 ? 11 : (top >= 7 ? 7 : top);
        const int shift = top - bits;
        const int nbs = step == 0 ? 11 : 8;  // log2 of the bins zeroed and scanned (a later step's digits stay below 128)
        for (int i = threadIdx.x; i < (SEL_T << nbs); i += SEL_THREADS) h[i >> nbs][i & ((1 << nbs) - 1)] = 0;
        if (threadIdx.x == 0) {
            uint64_t p[SEL_T];
            int ld[SEL_T];
#pragma unroll
            for (int t = 0; t < SEL_T; t++) p[t] = st[2 * t];
            sel_leaders(p, ld);
#pragma unroll
            for (int t = 0; t < SEL_T; t++) leader[t] = ld[t];
        }
        __syncthreads();
        uint64_t hi[SEL_T];
        bool lead[SEL_T];
#pragma unroll
        for (int t = 0; t < SEL_T; t++) { hi[t] = st[2 * t] >> top; lead[t] = leader[t] == t; }
        if (cached) {
            const uint32_t nc = n_cache;
            for (uint32_t i = threadIdx.x; i < nc; i += SEL_THREADS) sel_fin_vote(cache[i], shift, bits, hi, lead, h);
        } else if (in_regs) {
#pragma unroll
            for (int u = 0; u < FIN_REG; u++)
                if (u * SEL_THREADS + threadIdx.x < m) sel_fin_vote(kr[u], shift, bits, hi, lead, h);
        } else {
            for (uint32_t i0 = 0; i0 < m; i0 += 4 * SEL_THREADS) {  // four loads in flight
                uint64_t k[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t i = i0 + u * SEL_THREADS + threadIdx.x;
                    k[u] = i < m ? list[i] : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (i0 + u * SEL_THREADS + threadIdx.x < m) sel_fin_vote(k[u], shift, bits, hi, lead, h);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && !cached) n_cache = 0;
        if (threadIdx.x < 64 * SEL_T) {  // a wave per target (it touches only its own two words of st)
            const int lane = (int)threadIdx.x & 63, t = (int)threadIdx.x >> 6;
            const uint32_t *hist = &h[leader[t]][0];
            const uint64_t r = st[2 * t + 1];
            uint64_t lsum, inc, below;
            uint32_t bin;
            if (step == 0) {
                sel_scan<SEL_FIN_NB>(lane, hist, lsum, inc);
                sel_pick<SEL_FIN_NB>(lane, hist, lsum, inc, r, bin, below);
            } else {
                sel_scan<256>(lane, hist, lsum, inc);
                sel_pick<256>(lane, hist, lsum, inc, r, bin, below);
            }
            if (lane == 0) {
                st[2 * t] |= (uint64_t)bin << shift;
                st[2 * t + 1] = r - below;
            }
        }
        __syncthreads();
        top = shift;
        if (top == 0) break;
        if (!cached) {
            // the survivors of this step: how many is known from the histograms, so the decision is uniform
            uint64_t keep = 0;
#pragma unroll
            for (int t = 0; t < SEL_T; t++)
                if (leader[t] == t || (st[2 * t] != st[2 * leader[t]]))  // one count per distinct new prefix
                    keep += h[leader[t]][(uint32_t)(st[2 * t] >> shift) & ((1u << bits) - 1u)];
            // (targets that shared a histogram and still share the prefix were counted once; targets that split were
            //  counted each: keep >= the number of distinct survivors)
            if (keep <= SEL_FIN_CAP) {
                uint64_t pre[SEL_T];
#pragma unroll
                for (int t = 0; t < SEL_T; t++) pre[t] = st[2 * t] >> shift;
                if (in_regs) {
#pragma unroll
                    for (int u = 0; u < FIN_REG; u++) {
                        bool mt = false;
#pragma unroll
                        for (int t = 0; t < SEL_T; t++) mt |= (kr[u] >> shift) == pre[t];
                        if (mt && u * SEL_THREADS + threadIdx.x < m) cache[atomicAdd(&n_cache, 1u)] = kr[u];
                    }
                } else {
                    for (uint32_t i0 = 0; i0 < m; i0 += 4 * SEL_THREADS) {
                        uint64_t k[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const uint32_t i = i0 + u * SEL_THREADS + threadIdx.x;
                            k[u] = i < m ? list[i] : 0;
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            bool mt = false;
#pragma unroll
                            for (int t = 0; t < SEL_T; t++) mt |= (k[u] >> shift) == pre[t];
                            if (mt && i0 + u * SEL_THREADS + threadIdx.x < m) cache[atomicAdd(&n_cache, 1u)] = k[u];
                        }
                    }
                }
                cached = true;
            }
            __syncthreads();
        }
        if (cached && n_cache <= SEL_FIN_DIRECT) {
            // a key per thread: the number of survivors with the same prefix below it (ties by index), by counting; the
            // thread whose count equals a target's remaining rank holds that order statistic
            const uint32_t nc = n_cache;
            if (threadIdx.x < nc) {
                const uint64_t mine = cache[threadIdx.x], mp = mine >> shift;
                uint32_t below = 0;
                for (uint32_t j = 0; j < nc; j++) {
                    const uint64_t kj = cache[j];
                    below += ((kj >> shift) == mp && (kj < mine || (kj == mine && j < threadIdx.x))) ? 1u : 0u;
                }
#pragma unroll
                for (int t = 0; t < SEL_T; t++)
                    if (mp == (st[2 * t] >> shift) && (uint64_t)below == st[2 * t + 1]) cache[SEL_FIN_CAP + t] = mine;
            }
            __syncthreads();
            if (threadIdx.x < SEL_T) st[2 * threadIdx.x] = cache[SEL_FIN_CAP + threadIdx.x];
            __syncthreads();
            break;
        }
    }
    if (threadIdx.x != 0) return;
    double w[SEL_T];
#pragma unroll
    for (int t = 0; t < SEL_T; t++) { w[t] = value_of(st[2 * t]); v[t] = w[t]; }
    const double median = (q.n % 2 != 0) ? w[1] : (w[0] + w[1]) / 2.0;
    // hf <= 0 / hf >= n only for n <= 2; the clamped ranks then already are min / max
    double q1, q3;
    if (q.hf1 <= 0) q1 = w[3]; else if (q.hf1 >= (int64_t)q.n) q1 = w[2]; else q1 = w[2] + (q.h1 - (double)q.hf1) * (w[3] - w[2]);
    if (q.hf3 <= 0) q3 = w[5]; else if (q.hf3 >= (int64_t)q.n) q3 = w[4]; else q3 = w[4] + (q.h3 - (double)q.hf3) * (w[5] - w[4]);
    const double iqr = q3 - q1;
    out[0] = median;
    out[1] = iqr;
    out[2] = q1 - q.iqr_multiple * iqr;  // main.rs:328-329
}

// median / quartiles / threshold of n keys into c->sel_out[8..10] = {median, iqr, threshold}, the six order statistics into
// c->sel_out[0..5] (device memory).  No host synchronisation.
cellector_status select_threshold(cellector_ctx *c, const double *keys, uint64_t n, double iqr_multiple)
{
    if (n == 0) return ctx_fail(c, CELLECTOR_EINVAL, "order statistics of an empty array");
    if (n > 0xffffffffull) return ctx_fail(c, CELLECTOR_EINVAL, "order statistics: more than 2^32 - 1 keys");
    // statrs Data::median / quantile: ranks of the order statistics needed
    const uint64_t k = n / 2;
    sel_quart_t q;
    q.h1 = ((double)n + 1.0 / 3.0) * 0.25 + 1.0 / 3.0;
    q.h3 = ((double)n + 1.0 / 3.0) * 0.75 + 1.0 / 3.0;
    q.hf1 = (int64_t)q.h1;
    q.hf3 = (int64_t)q.h3;
    q.n = n;
    q.iqr_multiple = iqr_multiple;
    auto clampr = [n](int64_t r) -> uint64_t { return r < 0 ? 0 : ((uint64_t)r >= n ? n - 1 : (uint64_t)r); };
    sel_ranks_t r;
    const uint64_t ranks[SEL_T] = {k ? k - 1 : 0, k, clampr(q.hf1 - 1), clampr(q.hf1), clampr(q.hf3 - 1), clampr(q.hf3)};
    for (int t = 0; t < SEL_T; t++) r.r[t] = ranks[t];
    if (c->sel_list_cap < n) {
        dev_free(c->sel_list);
        c->sel_list = nullptr;
        c->sel_list_cap = 0;
        CHK(dev_alloc(c, &c->sel_list, n));
        c->sel_list_cap = n;
    }
    uint32_t *hist0 = c->sel_hist, *hist1 = hist0 + SEL_NB0, *count = hist1 + SEL_T * SEL_NB1;
    uint64_t *state1 = c->sel_state, *state2 = c->sel_state + 2 * SEL_T;
    timer_begin(c, CELLECTOR_K_SELECT);
    uint64_t g = (n + SEL_THREADS * SEL_UNROLL - 1) / (SEL_THREADS * SEL_UNROLL);
    if (g > SEL_GRID) g = SEL_GRID;
    hipLaunchKernelGGL(k_sel_top, dim3((unsigned)g), dim3(SEL_THREADS), 0, c->stream, keys, n, hist0);
    hipLaunchKernelGGL(k_sel_mid, dim3((unsigned)g), dim3(SEL_THREADS), 0, c->stream, keys, n, r, hist0, state1, hist1);
    hipLaunchKernelGGL(k_sel_gather, dim3((unsigned)g), dim3(SEL_THREADS), 0, c->stream, keys, n, state1, hist1, state2,
                       c->sel_list, count);
    hipLaunchKernelGGL(k_sel_finish, dim3(1), dim3(SEL_THREADS), 0, c->stream, q, state2, c->sel_list, count, hist0, c->sel_out,
                       c->sel_out + 8);
    timer_end(c, CELLECTOR_K_SELECT);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// The same order statistics over the keys of ALL shards of a sharded run, every shard holding only its own cells' keys
// (select_threshold_sharded; the exchange layer of comm.h).  MSB-first radix select again, but nothing is gathered: the 64 key
// bits are resolved in six levels of 11, 11, 11, 11, 10 and 10 bits; a level is one kernel over the shard's keys (the step on
// the previous level's histograms, summed over the shards, then the histogram of the next digit of the keys that carry a
// target's resolved prefix) and one all-reduce of 48 KB of counts.  After the sixth level a target's key is known bit for bit:
// every shard computes the same threshold, with a fixed number of small collectives instead of an all-gather of all keys and
// a select over all of them on every shard (8 B per cell of the whole run per iteration, and work that does not shrink with
// the shard).
// ---------------------------------------------------------------------------------------------------------
#define SELD_LEVELS 6
#define SELD_NB 2048
__device__ __constant__ int SELD_SHIFT[SELD_LEVELS] = {53, 42, 31, 20, 10, 0};
__device__ __constant__ int SELD_BITS[SELD_LEVELS] = {11, 11, 11, 11, 10, 10};

// the step on level (L - 1)'s summed histograms: state_in (prefix, remaining rank per target) -> st[] in LDS; a wave per target
__device__ __forceinline__ void seld_step(int level_prev, const uint32_t *__restrict__ hist_prev, const uint64_t *__restrict__ state_in,
                                          uint64_t *st /*[2 * SEL_T] in LDS*/)
{
    if (threadIdx.x < 64 * SEL_T) {
        const int lane = (int)threadIdx.x & 63, t = (int)threadIdx.x >> 6;
        uint64_t p[SEL_T];
        int ld[SEL_T];
#pragma unroll
        for (int u = 0; u < SEL_T; u++) p[u] = state_in[2 * u];
        sel_leaders(p, ld);
        int mine = 0;
        uint64_t pt = 0;
#pragma unroll
        for (int u = 0; u < SEL_T; u++)
            if (u == t) { mine = ld[u]; pt = p[u]; }
        const uint64_t r = state_in[2 * t + 1];
        const uint32_t *hist = hist_prev + mine * SELD_NB;
        uint64_t lsum, inc, below;
        uint32_t bin;
        sel_scan<SELD_NB>(lane, hist, lsum, inc);
        sel_pick<SELD_NB>(lane, hist, lsum, inc, r, bin, below);
        if (lane == 0) {
            st[2 * t] = pt | ((uint64_t)bin << SELD_SHIFT[level_prev]);
            st[2 * t + 1] = r - below;
        }
    }
}

template <bool FIRST>
__global__ __launch_bounds__(SEL_THREADS) void k_seld_level(const double *__restrict__ keys, uint64_t n, int level, sel_ranks_t ranks,
                                                            const uint32_t *__restrict__ hist_prev, const uint64_t *__restrict__ state_in,
                                                            uint64_t *__restrict__ state_out, uint32_t *__restrict__ hist_out)
{
    __shared__ uint32_t h[FIRST ? 1 : SEL_T][SELD_NB];
    __shared__ uint64_t st[2 * SEL_T];
    for (int i = threadIdx.x; i < (FIRST ? 1 : SEL_T) * SELD_NB; i += SEL_THREADS) (&h[0][0])[i] = 0;
    if constexpr (!FIRST) {
        seld_step(level - 1, hist_prev, state_in, st);
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x < 2 * SEL_T) state_out[threadIdx.x] = st[threadIdx.x];
    } else {
        // nothing is resolved yet: all targets share histogram 0; the step on it is taken by level 1
        if (blockIdx.x == 0 && threadIdx.x < SEL_T) { state_out[2 * threadIdx.x] = 0; state_out[2 * threadIdx.x + 1] = ranks.r[threadIdx.x]; }
        __syncthreads();
    }
    const int shift = SELD_SHIFT[level], bits = SELD_BITS[level];
    const uint32_t dmask = (1u << bits) - 1u;
    uint64_t hi[SEL_T];
    bool lead[SEL_T];
    if constexpr (!FIRST) {
        uint64_t p[SEL_T];
        int ld[SEL_T];
#pragma unroll
        for (int t = 0; t < SEL_T; t++) p[t] = st[2 * t];
        sel_leaders(p, ld);
#pragma unroll
        for (int t = 0; t < SEL_T; t++) { hi[t] = p[t] >> (shift + bits); lead[t] = ld[t] == t; }
    }
    const int lane = threadIdx.x & 63;
    const uint64_t stride = (uint64_t)gridDim.x * SEL_THREADS;
    for (uint64_t base = (uint64_t)blockIdx.x * SEL_THREADS; base < n; base += SEL_UNROLL * stride) {
        uint64_t k[SEL_UNROLL];
        bool ok[SEL_UNROLL];
#pragma unroll
        for (int u = 0; u < SEL_UNROLL; u++) {
            const uint64_t i = base + u * stride + threadIdx.x;
            ok[u] = i < n;
            k[u] = ok[u] ? key_of(keys[i]) : 0;
        }
#pragma unroll
        for (int u = 0; u < SEL_UNROLL; u++) {
            const uint32_t d = (uint32_t)(k[u] >> shift) & dmask;
            if constexpr (FIRST) {  // sign + exponent bits: a handful of bins, a wave whose keys agree votes with one atomic
                const unsigned long long act = __ballot(ok[u]);
                if (!act) continue;  // wave-uniform
                const int src = __ffsll((long long)act) - 1;
                const uint32_t d0 = (uint32_t)__shfl((int)d, src, 64);
                const unsigned long long same = __ballot(ok[u] && d == d0);
                if (same == act) {
                    if (lane == src) atomicAdd(&h[0][d0], (uint32_t)__popcll(act));
                } else if (ok[u]) {
                    atomicAdd(&h[0][d], 1u);
                }
            } else {
                if (!ok[u]) continue;
                const uint64_t kh = k[u] >> (shift + bits);
#pragma unroll
                for (int t = 0; t < SEL_T; t++)
                    if (lead[t] && kh == hi[t]) atomicAdd(&h[t][d], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (FIRST ? 1 : SEL_T) * SELD_NB; i += SEL_THREADS) {
        const uint32_t v = (&h[0][0])[i];
        if (v) atomicAdd(&hist_out[i], v);
    }
}

// the step on the last level's histograms: the six keys, bit for bit; then the threshold arithmetic of k_sel_finish
__global__ __launch_bounds__(64 * SEL_T) void k_seld_finish(sel_quart_t q, const uint32_t *__restrict__ hist_prev,
                                                            const uint64_t *__restrict__ state_in, double *__restrict__ v,
                                                            double *__restrict__ out)
{
    __shared__ uint64_t st[2 * SEL_T];
    seld_step(SELD_LEVELS - 1, hist_prev, state_in, st);
    __syncthreads();
    if (threadIdx.x != 0) return;
    double w[SEL_T];
#pragma unroll
    for (int t = 0; t < SEL_T; t++) { w[t] = value_of(st[2 * t]); v[t] = w[t]; }
    const double median = (q.n % 2 != 0) ? w[1] : (w[0] + w[1]) / 2.0;
    double q1, q3;
    if (q.hf1 <= 0) q1 = w[3]; else if (q.hf1 >= (int64_t)q.n) q1 = w[2]; else q1 = w[2] + (q.h1 - (double)q.hf1) * (w[3] - w[2]);
    if (q.hf3 <= 0) q3 = w[5]; else if (q.hf3 >= (int64_t)q.n) q3 = w[4]; else q3 = w[4] + (q.h3 - (double)q.hf3) * (w[5] - w[4]);
    const double iqr = q3 - q1;
    out[0] = median;
    out[1] = iqr;
    out[2] = q1 - q.iqr_multiple * iqr;  // main.rs:328-329
}

// keys = this shard's n_local keys, n_total = the keys of all shards; results as select_threshold (identical on every shard)
cellector_status select_threshold_sharded(cellector_ctx *c, const double *keys, uint64_t n_local, uint64_t n_total, double iqr_multiple)
{
    const uint64_t n = n_total;
    if (n == 0) return ctx_fail(c, CELLECTOR_EINVAL, "order statistics of an empty array");
    if (n > 0xffffffffull) return ctx_fail(c, CELLECTOR_EINVAL, "order statistics: more than 2^32 - 1 keys");
    const uint64_t k = n / 2;
    sel_quart_t q;
    q.h1 = ((double)n + 1.0 / 3.0) * 0.25 + 1.0 / 3.0;
    q.h3 = ((double)n + 1.0 / 3.0) * 0.75 + 1.0 / 3.0;
    q.hf1 = (int64_t)q.h1;
    q.hf3 = (int64_t)q.h3;
    q.n = n;
    q.iqr_multiple = iqr_multiple;
    auto clampr = [n](int64_t r) -> uint64_t { return r < 0 ? 0 : ((uint64_t)r >= n ? n - 1 : (uint64_t)r); };
    const uint64_t ranks[SEL_T] = {k ? k - 1 : 0, k, clampr(q.hf1 - 1), clampr(q.hf1), clampr(q.hf3 - 1), clampr(q.hf3)};
    constexpr uint64_t HW = (uint64_t)SEL_T * SELD_NB;  // histogram words of a level
    if (!c->seld_hist) CHK(dev_alloc(c, &c->seld_hist, SELD_LEVELS * HW));
    if (!c->seld_state) CHK(dev_alloc(c, &c->seld_state, (uint64_t)(SELD_LEVELS + 1) * 2 * SEL_T));
    sel_ranks_t r;
    for (int t = 0; t < SEL_T; t++) r.r[t] = ranks[t];
    timer_begin(c, CELLECTOR_K_SELECT);
    HIPCHK(c, hipMemsetAsync(c->seld_hist, 0, SELD_LEVELS * HW * sizeof(uint32_t), c->stream));
    uint64_t g = (n_local + SEL_THREADS * SEL_UNROLL - 1) / (SEL_THREADS * SEL_UNROLL);
    if (g > SEL_GRID) g = SEL_GRID;
    if (g == 0) g = 1;  // (a shard without cells still takes the steps: it writes no state anybody reads, but keeps the kernels uniform)
    for (int level = 0; level < SELD_LEVELS; level++) {
        uint32_t *hist = c->seld_hist + (uint64_t)level * HW;
        const uint32_t *prev = level ? c->seld_hist + (uint64_t)(level - 1) * HW : nullptr;
        const uint64_t *s_in = c->seld_state + (uint64_t)level * 2 * SEL_T;
        uint64_t *s_out = c->seld_state + (uint64_t)(level + 1) * 2 * SEL_T;
        if (level == 0)
            hipLaunchKernelGGL(k_seld_level<true>, dim3((unsigned)g), dim3(SEL_THREADS), 0, c->stream, keys, n_local, level, r, prev, s_in, s_out, hist);
        else
            hipLaunchKernelGGL(k_seld_level<false>, dim3((unsigned)g), dim3(SEL_THREADS), 0, c->stream, keys, n_local, level, r, prev, s_in, s_out, hist);
        HIPCHK(c, hipGetLastError());
        CHK((cellector_status)comm_allreduce_sum_u32(c, hist, level == 0 ? SELD_NB : HW));
    }
    hipLaunchKernelGGL(k_seld_finish, dim3(1), dim3(64 * SEL_T), 0, c->stream, q, c->seld_hist + (uint64_t)(SELD_LEVELS - 1) * HW,
                       c->seld_state + (uint64_t)SELD_LEVELS * 2 * SEL_T, c->sel_out, c->sel_out + 8);
    timer_end(c, CELLECTOR_K_SELECT);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}
