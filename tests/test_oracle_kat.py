"""CPU tests of the oracle (test infrastructure) against builder-generated known answers.

PARITY UNPINNED: the reference has no tests/fixtures and cannot be built here; these anchors are (a) the
SURVEY Appendix B.5 reference-style values (bit-exact regression of the oracle's operation order) and (b)
independent mpmath / scipy / numpy truths with the tolerance the reference formula's own cancellation allows.
"""
import json
import math
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "math_kat.json")))


@pytest.fixture(scope="module")
def ob(oracle_lib):
    return oracle_lib


def test_reference_style_anchors_bit_exact(ob):
    L = ob.lib()
    rs = GOLD["reference_style"]
    for x, v in rs["ln_gamma"]:
        assert L.orc_ln_gamma(x) == v
    for n, k, v in rs["ln_binomial"]:
        assert L.orc_ln_binomial(n, k) == v
    for a, b, v in rs["logsumexp"]:
        assert L.orc_logsumexp(a, b) == v
    for a, r, al, be, v in rs["log_beta_binomial_pmf"]:
        assert L.orc_log_beta_binomial_pmf(a, r, al, be, L.orc_ln_binomial(a + r, a)) == v
    for n, al, be, v in rs["expected_log_pmf"]:
        assert ob.expected_log_pmf(n, al, be)[0] == v
    x = np.arange(1, 11, dtype=float)
    q = rs["quantiles_1_to_10"]
    assert (ob.quantile(x, 0.25), ob.median(x), ob.quantile(x, 0.75)) == (q["q1"], q["median"], q["q3"])


def test_ln_gamma_and_binomial_vs_mpmath(ob):
    L = ob.lib()
    for x, truth in GOLD["ln_gamma"]:
        assert abs(L.orc_ln_gamma(x) - truth) <= 2e-15 * max(1.0, abs(truth)) + 5e-15
    for n, k, truth in GOLD["ln_binomial"]:
        assert abs(L.orc_ln_binomial(n, k) - truth) <= 1e-12 * max(1.0, abs(truth))
    assert L.orc_ln_binomial(3, 5) == -math.inf


def test_log_beta_binomial_pmf_vs_mpmath_and_scipy(ob):
    L = ob.lib()
    for c in GOLD["log_beta_binomial_pmf"]:
        a, r, al, be = c["alt"], c["ref"], c["alpha"], c["beta"]
        v = L.orc_log_beta_binomial_pmf(a, r, al, be, L.orc_ln_binomial(a + r, a))
        # cancellation of ~|lnGamma(alpha+beta)|-sized terms: abs error ~ eps * lnGamma(alpha+beta+n)
        tol = 4e-16 * abs(math.lgamma(al + be + a + r)) * 4 + 1e-13
        assert abs(v - c["truth"]) <= tol, (c, v)
        assert abs(c["scipy"] - c["truth"]) <= 1e-9
    assert L.orc_log_beta_binomial_pmf(0, 0, 3.0, 9.0, 0.0) == 0.0  # quirk Q14


def test_expected_log_pmf_vs_mpmath(ob):
    for c in GOLD["expected_log_pmf"]:
        e, var = ob.expected_log_pmf(c["n"], c["alpha"], c["beta"])
        assert abs(e - c["truth"]) <= 1e-9, c
        assert var >= 0.0


def test_order_statistics_vs_numpy(ob):
    for c in GOLD["quantiles"]:
        x = np.array(c["x"])
        assert ob.median(x) == pytest.approx(c["median"], abs=1e-15)
        assert ob.quantile(x, 0.25) == pytest.approx(c["q1"], abs=1e-12)
        assert ob.quantile(x, 0.75) == pytest.approx(c["q3"], abs=1e-12)
    assert math.isnan(ob.median(np.zeros(0)))
    assert math.isnan(ob.quantile(np.zeros(0), 0.25))


def test_binomial_pmf_and_vcf_genotype(ob):
    L = ob.lib()
    for p, n, k, v in GOLD["binomial_pmf"]:
        assert L.orc_binomial_pmf(p, n, k) == pytest.approx(v, rel=1e-11, abs=1e-300)
    assert L.orc_binomial_pmf(0.0, 5, 0) == 1.0 and L.orc_binomial_pmf(1.0, 5, 5) == 1.0
    assert L.orc_binomial_pmf(0.3, 5, 6) == 0.0
    # genotype rule: deep homozygous-alt majority, heterozygous minority, no reads -> ./. with 1/3
    gmaj, pmaj, gmin, pmin = ob.vcf_genotype(20, 22, 400, 3)
    assert (gmaj, gmin) == (1, 2) and pmaj > 0.99 and pmin > 0.99
    gmaj, pmaj, gmin, pmin = ob.vcf_genotype(0, 0, 0, 0)
    assert (gmaj, gmin) == (0, 0) and pmaj == pytest.approx(1 / 3) and pmin == pytest.approx(1 / 3)
