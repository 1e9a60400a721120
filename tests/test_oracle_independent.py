"""CPU: the C oracle against an INDEPENDENT numpy / scipy restatement of the same path, written from the reference's
formulas with library arithmetic: scipy.stats.betabinom for the log-pmf (what the reference's deprecated
Python script used for the same quantity, cellector_depricated.py:149), numpy sorts for the order statistics, np.logaddexp.
It shares no code with oracle/cellector_oracle.c, so it pins the oracle's FLOW — which alpha/beta an iteration uses, which
exclusion set the locus statistics see (quirk Q9), when the -80 filter takes effect (Q8), zero-locus cells in the quantiles
(Q4), the all-loci posterior pass (Q1), the three alpha/beta sets and priors (Q10) — not statrs' last bits: the oracle is
still "parity unpinned" against the Rust binary (it cannot be built here)."""
import numpy as np
import pytest
from scipy.stats import betabinom

from cellector_amd import synth


def _quantile_r8(srt, tau):
    n = len(srt)
    h = (n + 1.0 / 3.0) * tau + 1.0 / 3.0
    hf = int(h)
    if hf <= 0:
        return srt[0]
    if hf >= n:
        return srt[-1]
    return srt[hf - 1] + (h - hf) * (srt[hf] - srt[hf - 1])


class NumpyCellector:
    """main.rs:36-50,228-347,368-451,541-611 + load_data.rs:134-181,254-280 in numpy"""

    def __init__(self, L, N, lo, ce, al, re, min_alt=4, min_ref=4):
        lo, ce, al, re = (np.asarray(x, np.int64) for x in (lo, ce, al, re))
        c_ref = np.bincount(lo[re > 0], minlength=L)
        c_alt = np.bincount(lo[al > 0], minlength=L)
        used = (c_ref >= min_ref) & (c_alt >= min_alt)           # cells with a read, not reads (quirk Q3)
        self.to_used = np.cumsum(used) - 1
        keep = used[lo]
        self.l = self.to_used[lo[keep]]
        self.c, self.a, self.r = ce[keep], al[keep], re[keep]
        self.n = self.a + self.r
        self.L, self.N = int(used.sum()), N
        self.s_alt = np.bincount(self.l, weights=self.a, minlength=self.L)
        self.s_ref = np.bincount(self.l, weights=self.r, minlength=self.L)
        self.mask = np.ones(self.L, bool)
        self.excluded = np.zeros(N, bool)
        self.entries_per_cell = np.bincount(self.c, minlength=N)

    def alpha_beta(self, excl):
        ex = excl[self.c]
        a = (self.s_alt + 1.0) - np.bincount(self.l[ex], weights=self.a[ex], minlength=self.L)
        b = (self.s_ref + 1.0) - np.bincount(self.l[ex], weights=self.r[ex], minlength=self.L)
        return a, b

    def cell_ll(self, alpha, beta, mask):
        live = mask[self.l]
        lp = betabinom.logpmf(self.a, self.n, alpha[self.l], beta[self.l])
        lp = np.where(self.n == 0, 0.0, lp)                       # quirk Q14
        ll = np.bincount(self.c[live], weights=lp[live], minlength=self.N)
        nl = np.bincount(self.c[live], minlength=self.N).astype(float)
        return ll, nl, lp

    def expected_ll(self, alpha, beta, mask):
        """ln sum_k pmf(k)^2 per entry (stats.rs:8-22), summed per cell"""
        e = np.zeros(len(self.l))
        for n in np.unique(self.n):
            m = self.n == n
            ks = np.arange(n + 1)
            p = betabinom.pmf(ks[None, :], n, alpha[self.l[m]][:, None], beta[self.l[m]][:, None])
            e[m] = np.log((p * p).sum(axis=1))
        live = mask[self.l]
        return np.bincount(self.c[live], weights=e[live], minlength=self.N)

    def iteration(self, mult=5.0):
        alpha, beta = self.alpha_beta(self.excluded)               # the PREVIOUS exclusion set's alpha/beta
        ll, nl, lp = self.cell_ll(alpha, beta, self.mask)
        norm = np.where(nl > 0, ll / np.where(nl > 0, nl, 1.0), 0.0)   # zero-locus cells enter as 0.0 (Q4)
        srt = np.sort(norm)
        k = self.N // 2
        med = srt[k] if self.N % 2 else (srt[k - 1] + srt[k]) / 2.0
        q1, q3 = _quantile_r8(srt, 0.25), _quantile_r8(srt, 0.75)
        thr = q1 - mult * (q3 - q1)
        new = norm < thr                                           # strict
        change = bool((new != self.excluded).any())
        # locus statistics: this iteration's log-pmfs (old alpha/beta), NEW exclusion set (Q9), unmasked entries only
        live = self.mask[self.l]
        mn = live & new[self.c]
        cmin = np.bincount(self.l[mn], weights=lp[mn], minlength=self.L)
        nmin = np.bincount(self.l[mn], minlength=self.L)
        per_cell = np.where(nmin > 0, cmin / np.where(nmin > 0, nmin, 1), 0.0)
        out = dict(ll=ll, nl=nl, norm=norm, median=med, iqr=q3 - q1, threshold=thr, excluded=new.copy(), any_change=change,
                   cells_min=nmin, contrib_min=cmin, mask_used=self.mask.copy(), alpha=alpha, beta=beta)
        self.mask = self.mask & ~(per_cell < -80.0)                # takes effect from the next iteration on (Q8)
        self.excluded = new
        return out

    def posteriors(self):
        N, ex = self.N, self.excluded
        a_maj, b_maj = self.alpha_beta(ex)
        a_min, b_min = self.alpha_beta(~ex)
        mf0 = (ex.sum() + 1.0) / (N + 1.0)
        a_dbl, b_dbl = (a_maj - 1.0) * mf0 + (a_min - 1.0) + 1.0, (b_maj - 1.0) * mf0 + (b_min - 1.0) + 1.0   # unclamped (Q10)
        mf = max(mf0, 0.01)
        a_maj, b_maj = (a_maj - 1.0) * mf + 1.0, (b_maj - 1.0) * mf + 1.0
        allm = np.ones(self.L, bool)                               # the posterior pass uses ALL loci (Q1)
        l_min, l_maj, l_dbl = (self.cell_ll(a, b, allm)[0] for a, b in ((a_min, b_min), (a_maj, b_maj), (a_dbl, b_dbl)))
        lpd = np.log(N / 1000.0 / 100.0 * max(mf, 0.1))
        num = np.log(mf) + l_min
        den = np.logaddexp(np.logaddexp(num, np.log(1.0 - mf) + l_maj), lpd + l_dbl)
        return np.exp(num - den), np.exp(lpd + l_dbl - den), l_maj, l_min


def _compare(oracle_lib, L, N, coo, min_alt=4, min_ref=4, check_expected=False):
    o = oracle_lib.Oracle.from_coo(L, N, *coo, min_alt, min_ref)
    p = NumpyCellector(L, N, *coo, min_alt, min_ref)
    assert (o.loci_used, o.nnz) == (p.L, len(p.l))
    assert np.array_equal(o.entries_per_cell(), p.entries_per_cell)
    n_filtered = 0
    for it in range(30):
        so, sp = o.em_iteration(5.0), p.iteration(5.0)
        co = o.cell_outputs()
        assert np.array_equal(co["loci_used"], sp["nl"])
        np.testing.assert_allclose(co["ll"], sp["ll"], rtol=0, atol=1e-7)
        np.testing.assert_allclose([so.median, so.iqr, so.threshold], [sp["median"], sp["iqr"], sp["threshold"]], rtol=0, atol=1e-9)
        assert not (np.abs(sp["norm"] - sp["threshold"]) < 1e-9).any()   # nobody sits on the threshold: flags are decidable
        assert np.array_equal(o.excluded().astype(bool), sp["excluded"]) and bool(so.any_change) == sp["any_change"]
        lo_ = o.locus_outputs()
        assert np.array_equal(lo_["cells_min"], sp["cells_min"])
        np.testing.assert_allclose(lo_["contrib_min"], sp["contrib_min"], rtol=1e-9, atol=1e-7)
        assert np.array_equal(o.loci_mask().astype(bool), p.mask)
        n_filtered += int(so.n_loci_filtered)
        if check_expected and it == 0:
            np.testing.assert_allclose(co["expected_ll"], p.expected_ll(sp["alpha"], sp["beta"], sp["mask_used"]), rtol=0, atol=1e-7)
        if not so.any_change:
            break
    assert it >= 1
    po = o.posteriors()
    post, dbl, l_maj, l_min = p.posteriors()
    np.testing.assert_allclose(po["ll_majority"], l_maj, rtol=0, atol=1e-7)
    np.testing.assert_allclose(po["ll_minority"], l_min, rtol=0, atol=1e-7)
    np.testing.assert_allclose(po["posterior"], post, rtol=0, atol=1e-6)
    np.testing.assert_allclose(po["doublet_posterior"], dbl, rtol=0, atol=1e-6)
    o.close()
    return n_filtered, p


def test_oracle_flow_matches_numpy_restatement_cfg1(oracle_lib):
    L, N = 2000, 1000   # BASELINE configs[0]
    coo = synth.generate_coo(L, N, 0.10, seed=4, minority_fraction=0.05, doublet_fraction=0.01)
    _, p = _compare(oracle_lib, L, N, coo, check_expected=True)
    assert 0 < p.excluded.sum() < 0.1 * N


def test_oracle_flow_with_locus_filter_and_empty_cells(oracle_lib):
    """a pathological locus that the -80 filter removes after the first iteration, and cells without any entry"""
    L, N = 400, 600
    lo, ce, al, re = synth.generate_coo(L, N, 0.25, seed=9, minority_fraction=0.1)
    cls = synth.cell_classes(N, seed=9, minority_fraction=0.1)
    keep = ce % 11 != 5                                        # cells 5, 16, 27, ... lose every entry (Q4)
    lo, ce, al, re = lo[keep], ce[keep], al[keep], re[keep]
    cells = np.arange(N, dtype=np.uint32)[np.arange(N) % 11 != 5]
    lo = np.concatenate([lo, np.full(len(cells), L, np.uint32)])
    ce = np.concatenate([ce, cells])
    al = np.concatenate([al, np.where(cls[cells] == 1, 60, 0).astype(np.uint32)])
    re = np.concatenate([re, np.where(cls[cells] == 1, 0, 60).astype(np.uint32)])
    n_filtered, p = _compare(oracle_lib, L + 1, N, (lo, ce, al, re))
    assert n_filtered >= 1 and not p.mask.all()
    assert (p.entries_per_cell == 0).sum() >= N // 11


def test_vcf_genotype_rule_matches_scipy_binomial(oracle_lib):
    """output_final_vcf's 3-genotype posterior (main.rs:88-118): p = 0.97 * {0.99, 0.5, 0.01} + 0.03 * soup fraction, uniform
    prior, call at posterior > 0.99 — against scipy.stats.binom"""
    from scipy.stats import binom
    rng = np.random.default_rng(5)
    names = {1: "1/1", 2: "0/1", 3: "0/0", 0: "./."}
    seen = set()
    cases = [(0, 0, 0, 0), (30, 0, 0, 40), (5, 5, 100, 3), (0, 1, 1, 0)] + \
        [tuple(int(x) for x in rng.integers(0, rng.choice([3, 30, 400]), 4)) for _ in range(300)]
    for amin, rmin, amaj, rmaj in cases:
        tot_a, tot_r = amin + amaj, rmin + rmaj
        soup = tot_a / (tot_a + tot_r) if tot_a + tot_r else 0.5
        ps = [0.97 * g + 0.03 * soup for g in (0.99, 0.5, 0.01)]
        gmaj, pmaj, gmin, pmin = oracle_lib.vcf_genotype(amin, rmin, amaj, rmaj)
        for (a, r), g, pmax in (((amaj, rmaj), gmaj, pmaj), ((amin, rmin), gmin, pmin)):
            lik = np.array([binom.pmf(a, a + r, p) for p in ps])
            post = lik / 3.0 / (lik / 3.0).sum()
            assert abs(post.max() - pmax) <= 1e-9 * max(pmax, 1e-300), (amin, rmin, amaj, rmaj)
            want = 0
            for code, q in ((1, post[0]), (2, post[1]), (3, post[2])):
                if q > 0.99:
                    want = code
                    break
            if abs(post.max() - 0.99) > 1e-9:
                assert g == want, (a, r, post, g)
            seen.add(names[g])
    assert seen == {"1/1", "0/1", "0/0", "./."}
