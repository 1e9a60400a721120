// Exact order statistics of N f64 keys on the device (statrs Data::median / Data::quantile need the
// values at up to six ranks: main.rs:324-327).
//
// MSB-first radix select, 8 bits per pass, all SEL_T target ranks refined together: one kernel per pass.  Keys that match
// a target's resolved prefix vote into that target's 256-bin LDS histogram (votes aggregated per wave before the LDS
// atomic, then per block before the global atomic); targets that still share a prefix share one histogram.  The
// selection step between two passes (walk each histogram to the bin holding the target rank) is done redundantly by
// wave 0 of every block at the start of the next pass — a separate single-wave kernel per pass cost more in launch gaps
// than the redundant 6 KB histogram read — with three rotating histogram buffers (read the previous pass', fill this
// pass', block 0 clears the next pass') and two state buffers.  The last step runs inside the threshold kernel.  The
// results stay in device memory: the flagging kernel reads the threshold there, so the phase needs no host round trip.
#include "ctx.h"

#define SEL_BLOCK 256
#define SEL_PASSES 8

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

// state[t] = {prefix (resolved high bits, low bits zero), remaining rank inside that prefix}
struct sel_ranks_t { uint64_t r[SEL_T]; };

// one wave: for each target find the bin of its (leader's) histogram that holds the remaining rank.
// Lane i owns bins 4i..4i+3; a shuffle scan over the lane sums locates the lane, then the bin inside it.
// `pass` = the pass that filled `hist`; newp / newr are valid in every lane.
__device__ __forceinline__ void sel_step(int lane, int pass, const uint64_t *__restrict__ state, const uint32_t *__restrict__ hist,
                                         uint64_t newp[SEL_T], uint64_t newr[SEL_T])
{
    uint64_t prefix[SEL_T], rank[SEL_T];
#pragma unroll
    for (int t = 0; t < SEL_T; t++) {
        prefix[t] = state[2 * t];
        rank[t] = state[2 * t + 1];
    }
    const int shift = 56 - 8 * pass;
#pragma unroll
    for (int t = 0; t < SEL_T; t++) {
        int ld = t;
#pragma unroll
        for (int u = SEL_T - 1; u >= 0; u--)
            if (u < t && prefix[u] == prefix[t]) ld = u;
        const uint4 h4 = reinterpret_cast<const uint4 *>(hist + ld * 256)[lane];
        const uint32_t hs[4] = {h4.x, h4.y, h4.z, h4.w};
        const uint32_t lsum = hs[0] + hs[1] + hs[2] + hs[3];
        uint32_t inc = lsum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        const uint64_t exc = inc - lsum;
        // the lane whose bins contain the rank (the last lane if counts are short: defensive, cannot happen)
        const bool mine = (exc <= rank[t] && rank[t] < (uint64_t)inc) || (lane == 63 && rank[t] >= (uint64_t)inc);
        uint32_t d = 0;
        uint64_t cum = exc;
        if (mine) {
            d = 4 * lane;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                if (cum + hs[q] <= rank[t] && d == (uint32_t)(4 * lane + q)) { cum += hs[q]; d++; }
            }
        }
        const unsigned long long who = __ballot(mine);
        const int src = __ffsll((long long)who) - 1;
        const uint32_t dsel = (uint32_t)__shfl((int)d, src, 64);
        const uint64_t cumsel = (uint64_t)__shfl((long long)cum, src, 64);
        newp[t] = prefix[t] | ((uint64_t)dsel << shift);
        newr[t] = rank[t] - cumsel;
    }
}

// One pass: (1) the selection step on the previous pass' histogram (wave 0 of every block, redundantly; pass 0 starts from
// the ranks), (2) the histogram of this pass' digit of the keys that match a target's resolved prefix.
__global__ __launch_bounds__(SEL_BLOCK) void k_sel_pass(const double *__restrict__ keys, uint64_t n, int pass, sel_ranks_t ranks,
                                                        const uint64_t *__restrict__ state_prev, uint64_t *__restrict__ state_cur,
                                                        const uint32_t *__restrict__ hist_prev, uint32_t *__restrict__ hist_cur,
                                                        uint32_t *__restrict__ hist_next)
{
    __shared__ uint32_t h[SEL_T][256];
    __shared__ uint64_t prefix[SEL_T];
    __shared__ int leader[SEL_T];
    for (int i = threadIdx.x; i < SEL_T * 256; i += SEL_BLOCK) (&h[0][0])[i] = 0;
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < SEL_T * 256; i += SEL_BLOCK) hist_next[i] = 0;  // nobody touches it during this pass
    if (threadIdx.x < 64) {
        uint64_t np[SEL_T], nr[SEL_T];
        if (pass == 0) {
#pragma unroll
            for (int t = 0; t < SEL_T; t++) { np[t] = 0; nr[t] = ranks.r[t]; }
        } else {
            sel_step((int)threadIdx.x, pass - 1, state_prev, hist_prev, np, nr);
        }
        if (threadIdx.x == 0) {
#pragma unroll
            for (int t = 0; t < SEL_T; t++) {
                prefix[t] = np[t];
                if (blockIdx.x == 0) { state_cur[2 * t] = np[t]; state_cur[2 * t + 1] = nr[t]; }
            }
            for (int t = 0; t < SEL_T; t++) {
                int ld = t;
                for (int u = 0; u < t; u++)
                    if (np[u] == np[t]) { ld = u; break; }
                leader[t] = ld;
            }
        }
    }
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const uint64_t himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    for (uint64_t i = (uint64_t)blockIdx.x * SEL_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * SEL_BLOCK) {
        const uint64_t k = key_of(keys[i]);
        const uint32_t digit = (uint32_t)(k >> shift) & 0xffu;
        for (int t = 0; t < SEL_T; t++) {
            if (leader[t] != t) continue;  // block-uniform
            bool match = (k & himask) == prefix[t];
            // wave-aggregated vote: one LDS atomic per distinct digit per wave
            unsigned long long todo = __ballot(match);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                const uint32_t d0 = (uint32_t)__shfl((int)digit, src, 64);
                const unsigned long long same = __ballot(match && digit == d0) & todo;
                if ((threadIdx.x & 63) == src) atomicAdd(&h[t][d0], (uint32_t)__popcll(same));
                todo &= ~same;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_T * 256; i += SEL_BLOCK) {
        const uint32_t v = (&h[0][0])[i];
        if (v) atomicAdd(&hist_cur[i], v);  // agent scope
    }
}

// statrs Data::median / quantile and the threshold of main.rs:328-329 from the six order statistics,
// in the reference's operation order: out = {median, iqr, threshold}
struct sel_quart_t { double h1, h3; int64_t hf1, hf3; uint64_t n; double iqr_multiple; };
// one wave: the last selection step (the six order statistics -> v[0..5]), then the threshold arithmetic -> out[0..2]
__global__ void k_threshold(sel_quart_t q, const uint64_t *__restrict__ state_last, const uint32_t *__restrict__ hist_last,
                            double *__restrict__ v, double *__restrict__ out)
{
    uint64_t np[SEL_T], nr[SEL_T];
    sel_step((int)threadIdx.x, SEL_PASSES - 1, state_last, hist_last, np, nr);
    if (threadIdx.x != 0) return;
#pragma unroll
    for (int t = 0; t < SEL_T; t++) v[t] = value_of(np[t]);
    const double median = (q.n % 2 != 0) ? v[1] : (v[0] + v[1]) / 2.0;
    // hf <= 0 / hf >= n only for n <= 2; the clamped ranks then already are min / max
    double q1, q3;
    if (q.hf1 <= 0) q1 = v[3]; else if (q.hf1 >= (int64_t)q.n) q1 = v[2]; else q1 = v[2] + (q.h1 - (double)q.hf1) * (v[3] - v[2]);
    if (q.hf3 <= 0) q3 = v[5]; else if (q.hf3 >= (int64_t)q.n) q3 = v[4]; else q3 = v[4] + (q.h3 - (double)q.hf3) * (v[5] - v[4]);
    const double iqr = q3 - q1;
    out[0] = median;
    out[1] = iqr;
    out[2] = q1 - q.iqr_multiple * iqr;  // main.rs:328-329
}

// The passes of the radix select for SEL_T 0-based ranks of n keys; the caller's final kernel runs the last step on
// (*state_last, *hist_last).  No host synchronisation.
static cellector_status select_passes(cellector_ctx *c, const double *keys, uint64_t n, const uint64_t ranks[SEL_T],
                                      const uint64_t **state_last, const uint32_t **hist_last)
{
    if (n == 0) return ctx_fail(c, CELLECTOR_EINVAL, "order statistics of an empty array");
    sel_ranks_t r;
    for (int t = 0; t < SEL_T; t++) {
        if (ranks[t] >= n) return ctx_fail(c, CELLECTOR_EINVAL, "rank %llu out of range", (unsigned long long)ranks[t]);
        r.r[t] = ranks[t];
    }
    // pass p fills buffer p % 3 and clears buffer (p + 1) % 3; buffer 0 is cleared here for pass 0
    uint32_t *hist[3] = {c->sel_hist, c->sel_hist + SEL_T * 256, c->sel_hist + 2 * SEL_T * 256};
    uint64_t *state[2] = {c->sel_state, c->sel_state + 2 * SEL_T};
    HIPCHK(c, hipMemsetAsync(hist[0], 0, SEL_T * 256 * sizeof(uint32_t), c->stream));
    uint64_t g = (n + SEL_BLOCK * 4 - 1) / (SEL_BLOCK * 4);
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    for (int pass = 0; pass < SEL_PASSES; pass++)
        hipLaunchKernelGGL(k_sel_pass, dim3((unsigned)g), dim3(SEL_BLOCK), 0, c->stream, keys, n, pass, r, state[(pass + 1) & 1],
                           state[pass & 1], hist[(pass + 2) % 3], hist[pass % 3], hist[(pass + 1) % 3]);
    *state_last = state[(SEL_PASSES - 1) & 1];
    *hist_last = hist[(SEL_PASSES - 1) % 3];
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}

// median / quartiles / threshold of n keys into c->sel_out[8..10] = {median, iqr, threshold}, the six order statistics into
// c->sel_out[0..5] (device memory)
cellector_status select_threshold(cellector_ctx *c, const double *keys, uint64_t n, double iqr_multiple)
{
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
    const uint64_t ranks[SEL_T] = {k ? k - 1 : 0, k, clampr(q.hf1 - 1), clampr(q.hf1), clampr(q.hf3 - 1), clampr(q.hf3)};
    const uint64_t *state_last = nullptr;
    const uint32_t *hist_last = nullptr;
    timer_begin(c, CELLECTOR_K_SELECT);
    CHK(select_passes(c, keys, n, ranks, &state_last, &hist_last));
    hipLaunchKernelGGL(k_threshold, dim3(1), dim3(64), 0, c->stream, q, state_last, hist_last, c->sel_out, c->sel_out + 8);
    timer_end(c, CELLECTOR_K_SELECT);
    HIPCHK(c, hipGetLastError());
    return CELLECTOR_OK;
}
