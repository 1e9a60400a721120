// f64 device arithmetic of the scoring path.
//
// The reference evaluates log_beta_binomial_pmf (stats.rs:41-46) with six Lanczos ln_gamma calls per
// entry.  With integer counts a, r (n = a + r) the same quantity is the exact finite product
//
//   B(a+α, r+β) / B(α, β) = Π_{i<a}(α+i) · Π_{j<r}(β+j) / Π_{k<n}(α+β+k)            (α, β > 0 real)
//
// so one log of a ratio of short products replaces the ln_gamma differences (which cancel ~1e5-sized
// terms and carry ~1e-11 absolute error).  Results agree with the reference formula to its own rounding
// noise; the parity tests state the tolerance.  ln C(n,a) follows the reference bit for bit: statrs
// ln_factorial = ln(FCACHE[x]) for x <= 170 (table uploaded from the host libm), ln_gamma(x+1) beyond.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#define DM_CHUNK 8  // factors multiplied between logs: (1e9)^8 and (1e-9)^8 stay far inside f64 range

// statrs ln_gamma (Lanczos) — only reached for ln_factorial(x > 170)
__device__ inline double dm_ln_gamma(double x)
{
    const double dk[11] = {2.48574089138753565546e-5,  1.05142378581721974210,    -3.45687097222016235469,
                           4.51227709466894823700,     -2.98285225323576655721,   1.05639711577126713077,
                           -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
                           4.63399473359905636708e-6,  -2.71994908488607703910e-9};
    double s = dk[0];
#pragma unroll
    for (int i = 1; i <= 10; i++) s += dk[i] / (x + (double)i - 1.0);
    return log(s) + 0.6207822376352452223455184457816472122518527279025978 +
           (x - 0.5) * log((x - 0.5 + 10.900511) / 2.71828182845904523536028747135266250);
}

__device__ __forceinline__ double dm_ln_factorial(const double *lf, uint32_t x)
{
    return x <= 170u ? lf[x] : dm_ln_gamma((double)x + 1.0);
}

// ln C(a+r, a) == statrs ln_binomial(n, k) = ln_factorial(n) - ln_factorial(k) - ln_factorial(n-k)
__device__ __forceinline__ double dm_ln_choose(const double *lf, uint32_t a, uint32_t r)
{
    return dm_ln_factorial(lf, a + r) - dm_ln_factorial(lf, a) - dm_ln_factorial(lf, r);
}

// ln[ B(a+α, r+β) / B(α, β) ]
__device__ __forceinline__ double dm_log_beta_ratio(double alpha, double beta, uint32_t a, uint32_t r)
{
    const uint32_t n = a + r;
    const double ab = alpha + beta;
    double num = 1.0, den = 1.0, acc = 0.0;
    uint32_t k = 0;
    while (k < n) {
        const uint32_t stop = min(n, k + (uint32_t)DM_CHUNK);
        for (; k < stop; ++k) {
            const double f = (k < a) ? (alpha + (double)k) : (beta + (double)(k - a));
            num *= f;
            den *= ab + (double)k;
        }
        if (k < n) {
            acc += log(num / den);
            num = 1.0;
            den = 1.0;
        }
    }
    return acc + log(num / den);
}

// log_beta_binomial_pmf(alt, ref, α, β, ln C) — stats.rs:41-46
__device__ __forceinline__ double dm_log_bb_pmf(const double *lf, double alpha, double beta, uint32_t a,
                                                uint32_t r)
{
    return dm_ln_choose(lf, a, r) + dm_log_beta_ratio(alpha, beta, a, r);
}

// ---- option ref_arith (engine 1): the reference's OWN arithmetic, operation for operation ---------------------------------
// log_beta_calc (stats.rs:48-53) and log_beta_binomial_pmf (stats.rs:41-46) with statrs' Lanczos ln_gamma in the reference's
// association order: (lnGamma(a) + lnGamma(b)) - lnGamma(a + b), then (lnC + numerator) - denominator.  Reproduces the
// reference's cancellation error (~eps * lnGamma(alpha + beta), i.e. 1e-11 .. 1e-9 per entry at alpha + beta of 1e4 .. 1e6)
// instead of avoiding it; what is left between this and the reference's bits is the last ulp of log() in the two maths
// libraries (the product form above is the default because it is more accurate and an order of magnitude cheaper).
// Arguments below 0.5 (statrs' reflection branch) do not occur: alpha, beta >= 1 by construction (main.rs:598-611).
__device__ inline double dm_log_beta_calc_ref(double a, double b)
{
    const double lga = dm_ln_gamma(a);
    const double lgb = dm_ln_gamma(b);
    const double lgab = dm_ln_gamma(a + b);
    return lga + lgb - lgab;
}
__device__ inline double dm_log_bb_pmf_ref(const double *lf, double alpha, double beta, uint32_t a, uint32_t r)
{
    const double lnc = dm_ln_choose(lf, a, r);
    const double num = dm_log_beta_calc_ref((double)a + alpha, (double)r + beta);
    const double den = dm_log_beta_calc_ref(alpha, beta);
    return lnc + num - den;
}

// stats.rs:35-39
__device__ __forceinline__ double dm_logsumexp(double x, double y)
{
    const double m = fmax(x, y);
    return m + log(exp(x - m) + exp(y - m));
}

// expected_log_beta_binomial_pmf(...).expected_log_likelihood = ln Σ_k pmf(k)^2 — stats.rs:8-22.
// Small n: pmf by the ratio recurrence pmf(k+1)/pmf(k) = (n-k)(α+k) / ((k+1)(β+n-k-1)), no logs until
// the end.  Large n (rare): the reference's log-space fold.
__device__ inline double dm_expected_log_pmf(const double *lf, double alpha, double beta, uint32_t n)
{
    if (n <= (uint32_t)DM_CHUNK) {
        const double ab = alpha + beta;
        double num = 1.0, den = 1.0;
        for (uint32_t j = 0; j < n; ++j) {
            num *= beta + (double)j;
            den *= ab + (double)j;
        }
        double p = num / den;
        double s = p * p;
        for (uint32_t k = 0; k < n; ++k) {
            p *= ((double)(n - k) * (alpha + (double)k)) / ((double)(k + 1) * (beta + (double)(n - k - 1)));
            s += p * p;
        }
        return log(s);
    }
    double e = 2.0 * dm_log_bb_pmf(lf, alpha, beta, 0u, n);
    for (uint32_t k = 1; k <= n; ++k) e = dm_logsumexp(e, 2.0 * dm_log_bb_pmf(lf, alpha, beta, k, n - k));
    return e;
}

// ---- wave64 reductions ------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// expected_log_beta_binomial_pmf with the reference's arithmetic (stats.rs:8-22): the log-space fold over all n + 1 pmfs
__device__ inline double dm_expected_log_pmf_ref(const double *lf, double alpha, double beta, uint32_t n)
{
    double e = 2.0 * dm_log_bb_pmf_ref(lf, alpha, beta, 0u, n);
    for (uint32_t k = 1; k <= n; ++k) e = dm_logsumexp(e, 2.0 * dm_log_bb_pmf_ref(lf, alpha, beta, k, n - k));
    return e;
}
