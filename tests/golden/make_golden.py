#!/usr/bin/env python3
"""Generates tests/golden/*.json — builder-made known answers for the oracle (PARITY UNPINNED: the reference
holds no fixtures and cannot be run here; see oracle/cellector_oracle.h).

Two kinds of values:
  * "truth": 50-digit mpmath evaluations of the mathematical quantities (ln Gamma, ln C(n,k), log
    beta-binomial pmf, ln sum pmf^2) and scipy's betabinom.logpmf — independent of the oracle's code;
  * "reference_style": the values SURVEY.md Appendix B.5 lists for the reference-style formula
    (Lanczos ln_gamma differences) — regression anchors for the oracle's operation order.
Run: python tests/golden/make_golden.py   (needs mpmath + scipy; only this script needs them).
"""
import json
import os

import mpmath as mp
import numpy as np
from scipy import stats

mp.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))


def log_bb(a, r, al, be):
    n = a + r
    return mp.log(mp.binomial(n, a)) + mp.log(mp.beta(a + al, r + be)) - mp.log(mp.beta(al, be))


def main():
    rng = np.random.default_rng(1234)
    out = {}
    xs = [1.0, 1.5, 2.0, 3.0, 7.25, 10.0, 101.0, 171.5, 1000.0, 12345.678, 2.0e5, 3.0e7]
    out["ln_gamma"] = [[x, float(mp.loggamma(x))] for x in xs]
    nk = [(5, 2), (1, 0), (1, 1), (2, 1), (100, 50), (130, 60), (170, 85), (171, 3), (200, 3), (1000, 500)]
    out["ln_binomial"] = [[n, k, float(mp.log(mp.binomial(n, k)))] for n, k in nk]
    cases = [(1, 0, 5, 7), (0, 1, 5, 7), (2, 1, 101, 4001), (0, 3, 12001, 8001), (3, 0, 1, 20001), (60, 70, 501, 499),
             (0, 0, 3, 9), (1, 1, 2.5, 7.75), (4, 9, 0.75, 1.25), (150, 90, 1234.5, 987.25)]
    for _ in range(30):
        cases.append((int(rng.integers(0, 6)), int(rng.integers(0, 6)), float(rng.integers(1, 30000)),
                      float(rng.integers(1, 30000))))
    out["log_beta_binomial_pmf"] = [
        {"alt": a, "ref": r, "alpha": al, "beta": be, "truth": float(log_bb(a, r, al, be)),
         "scipy": float(stats.betabinom.logpmf(a, a + r, al, be))} for a, r, al, be in cases]
    exp_cases = [(2, 5, 7), (1, 101, 4001), (0, 4, 4), (3, 12001, 8001), (8, 3.5, 2.25), (9, 40, 60), (30, 501, 499)]
    out["expected_log_pmf"] = [
        {"n": n, "alpha": al, "beta": be,
         "truth": float(mp.log(sum(mp.e ** (2 * log_bb(k, n - k, al, be)) for k in range(n + 1))))}
        for n, al, be in exp_cases]
    out["reference_style"] = {
        "ln_gamma": [[1.0, -4.440892098500626e-16], [3.0, 0.6931471805599485], [101.0, 363.7393755555637]],
        "ln_binomial": [[5, 2, 2.3025850929940455], [100, 50, 66.78384165201743], [130, 60, 87.06560835258631],
                        [200, 3, 14.08812975273895]],
        "logsumexp": [[-1.0, -2.0, -0.6867383124817772]],
        "log_beta_binomial_pmf": [[1, 0, 5, 7, -0.8754687373539269], [0, 1, 5, 7, -0.5389965007327397],
                                  [2, 1, 101, 4001, -6.3254157902809425], [0, 3, 12001, 8001, -2.7485722687270027],
                                  [3, 0, 1, 20001, -27.919153152150102], [60, 70, 501, 499, -3.0783208035428515]],
        "expected_log_pmf": [[2, 5, 7, -1.0018686737499964], [1, 101, 4001, -0.04922361868011212]],
        "quantiles_1_to_10": {"q1": 2.916666666666667, "median": 5.5, "q3": 8.083333333333334},
    }
    # R-8 quantiles / median against numpy's independent implementation (method 'median_unbiased' == R-8)
    qs = []
    for n in (1, 2, 3, 4, 5, 10, 11, 100, 101, 1000):
        x = rng.normal(size=n).round(6)
        qs.append({"x": x.tolist(), "median": float(np.median(x)),
                   "q1": float(np.quantile(x, 0.25, method="median_unbiased")),
                   "q3": float(np.quantile(x, 0.75, method="median_unbiased"))})
    out["quantiles"] = qs
    # Binomial pmf (VCF genotype rule)
    out["binomial_pmf"] = [[p, n, k, float(stats.binom.pmf(k, n, p))]
                           for p, n, k in [(0.9618, 10, 9), (0.5, 20, 7), (0.0112, 15, 0), (0.3, 0, 0), (0.97, 200, 190)]]
    with open(os.path.join(HERE, "math_kat.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(HERE, "math_kat.json"))


if __name__ == "__main__":
    main()
