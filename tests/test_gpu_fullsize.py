"""GPU parity at BASELINE.json's sizes.

  * configs[1] (50k cells x 50k loci, 1%): the WHOLE run — every EM iteration and the posterior phase — against the
    CPU oracle on the same matrix (the oracle's per-cell loop uses the host's cores; its results are bit-identical
    for any thread count);
  * configs[2] / configs[3] (200k x 100k and 1M x 200k, 1%): the oracle cannot finish these in seconds, so parity is
    checked (a) on a sample of cells with the oracle evaluating the reference formulas under the device's own
    per-locus alpha/beta, (b) between the two independent device engines on every cell, and (c) through the
    size-independent properties the path offers (conservation between the cell pass and the locus pass, exactness of
    the integer tallies, idempotence at the fixed point, recovery of the planted minority population);
  * a shard of more than 2^20 cells (the locus pass then reads the exclusion bitmask from L2 instead of LDS and
    stores 32-bit entries).

Tolerances as in test_gpu_parity.py: integers / flags / assignments exact, log-likelihood sums abs 1e-7, posteriors
abs 1e-6 (north_star).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LL_ATOL = 1e-7
POST_ATOL = 1e-6


def _lse(a, b):
    m = np.maximum(a, b)
    return m + np.log(np.exp(a - m) + np.exp(b - m))


def _posterior_alpha_betas(lc, alt_min, ref_min, n_excluded, n_cells):
    """The three alpha/beta sets of calculate_posteriors (main.rs:239-254) from the per-locus totals and the
    minority tallies, in the reference's operation order."""
    s_ref, s_alt = lc[:, 0], lc[:, 1]
    a_maj, b_maj = (s_alt + 1.0) - alt_min, (s_ref + 1.0) - ref_min
    a_min, b_min = (s_alt + 1.0) - (s_alt - alt_min), (s_ref + 1.0) - (s_ref - ref_min)
    mf0 = (n_excluded + 1.0) / (n_cells + 1.0)
    a_dbl = (a_maj - 1.0) * mf0 + (a_min - 1.0) + 1.0
    b_dbl = (b_maj - 1.0) * mf0 + (b_min - 1.0) + 1.0
    mf = max(mf0, 0.01)
    a_maj, b_maj = (a_maj - 1.0) * mf + 1.0, (b_maj - 1.0) * mf + 1.0
    lp_dbl = np.log(n_cells / 1000.0 / 100.0 * max(mf, 0.1))
    return (a_min, b_min), (a_maj, b_maj), (a_dbl, b_dbl), (np.log(mf), np.log(1.0 - mf), lp_dbl)


def dm_nnz(g):
    return g.dims().nnz_used


def _load(engine, L, N, d, seed=4, minority=0.05, doublet=0.0, deep=False):
    from cellector_amd import Cellector
    g = Cellector(0)
    g.set_option("engine", engine)
    g.set_option("keep_coo", 0)
    if deep:  # entry totals 1 + Geometric(0.4): 13 % of the entries outside the regular tables (bench.py's *-deep workloads)
        g.set_option("synth_continue_pct", 60)
    g.load_synthetic(L, N, d, seed=seed, minority_fraction=minority, doublet_fraction=doublet)
    return g


# ---- BASELINE configs[1]: the whole run against the oracle ------------------------------------------------------
_CFG2 = {}


def _cfg2_oracle_run(ob, g):
    """Oracle trajectory on the cfg2 matrix, computed once and shared by both engine parametrisations."""
    if "run" in _CFG2:
        return _CFG2["run"]
    L, N = 50_000, 50_000
    dm = g.dims()
    rp, ent = g.csr_rows(0, N)
    o = ob.Oracle.from_csr(dm.loci_used, rp, ent, g.locus_counts())
    ob.set_threads(ob.host_threads())
    try:
        iters = []
        while True:
            s = o.em_iteration(5.0)
            iters.append(dict(summary=(s.any_change, s.n_new_excluded, s.n_rescued, s.n_loci_filtered),
                              stats=(s.median, s.iqr, s.threshold), cells=o.cell_outputs(), loci=o.locus_outputs(),
                              excluded=o.excluded(), mask=o.loci_mask()))
            if not s.any_change or len(iters) >= 30:
                break
        post = o.posteriors()
        assign = o.assignments(post["posterior"], post["doublet_posterior"])
    finally:
        ob.set_threads(1)
    o.close()
    _CFG2["run"] = (iters, post, assign)
    return _CFG2["run"]


@pytest.mark.parametrize("engine", [2, 1], ids=["tiled", "csr"])
def test_cfg2_whole_run_matches_oracle(engine, oracle_lib, hip_lib_path):
    from cellector_amd import ffi
    g = _load(engine, 50_000, 50_000, 0.01, doublet=0.01)
    iters, post, assign = _cfg2_oracle_run(oracle_lib, g)
    assert len(iters) >= 2 and iters[-1]["summary"][0] == 0
    for it in iters:
        s = g.em_iteration(5.0)
        cg, co = g.cell_outputs(), it["cells"]
        assert np.array_equal(cg["loci_used"], co["loci_used"])
        np.testing.assert_allclose(cg["ll"], co["ll"], rtol=0, atol=LL_ATOL)
        np.testing.assert_allclose(cg["expected_ll"], co["expected_ll"], rtol=0, atol=LL_ATOL)
        np.testing.assert_allclose([s.median, s.iqr, s.threshold], it["stats"], rtol=0, atol=1e-9)
        near = np.abs(co["normalized"] - it["stats"][2]) < 1e-9
        assert not near.any(), f"near-tie cells at the threshold: {np.nonzero(near)[0]}"
        assert np.array_equal(g.excluded(), it["excluded"])
        assert (s.any_change, s.n_new_excluded, s.n_rescued, s.n_loci_filtered) == it["summary"]
        lg = g.locus_outputs()
        for k in ("cells_min", "cells_maj", "alt_min", "ref_min", "alt_maj", "ref_maj"):
            assert np.array_equal(lg[k], it["loci"][k]), k
        np.testing.assert_allclose(lg["contrib_min"], it["loci"]["contrib_min"], rtol=0, atol=LL_ATOL)
        np.testing.assert_allclose(lg["contrib_maj"], it["loci"]["contrib_maj"], rtol=0, atol=1e-5)
        assert np.array_equal(g.loci_mask(), it["mask"])
    pg = g.posteriors()
    for k in ("ll_majority", "ll_minority"):
        np.testing.assert_allclose(pg[k], post[k], rtol=0, atol=LL_ATOL)
    np.testing.assert_allclose(pg["posterior"], post["posterior"], rtol=0, atol=POST_ATOL)
    np.testing.assert_allclose(pg["doublet_posterior"], post["doublet_posterior"], rtol=0, atol=POST_ATOL)
    ga = ffi.assignments(pg["posterior"], pg["doublet_posterior"], g.entries_per_cell(), g.excluded())
    assert np.array_equal(ga[0], assign[0]) and np.array_equal(ga[1], assign[1])  # assignments bit-identical
    assert np.max(np.abs(ga[2].astype(np.int64) - assign[2].astype(np.int64))) <= 1
    assert {0, 1} <= set(ga[0].tolist())  # minority and majority labels both occur
    g.close()


# ---- BASELINE configs[2] and [3]: sampled oracle parity, engine cross-check, properties ----------------------------
@pytest.mark.parametrize("name,N,L", [("cfg3", 200_000, 100_000), ("cfg4", 1_000_000, 200_000), ("cfg3-deep", 200_000, 100_000)])
def test_full_size_sampled_parity_and_properties(name, N, L, oracle_lib, hip_lib_path):
    """(cfg3-deep: the same shape with deep coverage — the cell side of the totals 5..8 then walks the tier-2 tiles, five chunk
    groups of them, and the per-entry kernel the compact rest.)"""
    from cellector_amd import ffi, synth
    ob = oracle_lib
    deep = name.endswith("-deep")
    g = _load(2, L, N, 0.01, deep=deep)
    if deep:
        assert g.engine_info().nnz_overflow > 0.1 * dm_nnz(g)
    dm = g.dims()
    assert (dm.total_cells, dm.total_loci) == (N, L) and dm.loci_used > 0.9 * L
    lc = g.locus_counts()
    # the oracle's sample: row ranges spread over the whole matrix — the first 1024-cell block, ranges that straddle
    # block and 4-block column boundaries at several depths (different chunk-group work items), the middle, and the
    # ragged last block
    last_block = (N - 1) // 1024 * 1024
    ranges = [(0, 700), (1000, 1400), (4090, 4500), (N // 5 + 333, N // 5 + 333 + 500), (N // 2 - 300, N // 2 + 300),
              (2 * N // 3 + 17, 2 * N // 3 + 17 + 400), (last_block - 200, last_block + 100), (max(last_block + 100, N - 300), N)]
    sel = np.concatenate([np.arange(a, b) for a, b in ranges])
    assert len(sel) == len(np.unique(sel)) and sel[-1] == N - 1
    rps, ents = zip(*(g.csr_rows(a, b) for a, b in ranges))
    rp = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(r) for r in rps]))]).astype(np.uint64)
    ent = np.concatenate(ents)
    o = ob.Oracle.from_csr(dm.loci_used, rp, ent, lc)
    ob.set_threads(ob.host_threads())
    try:
        # -- EM loop: every iteration checked on the sample under the device's alpha/beta
        prev_excluded = np.zeros(N, np.uint8)
        for it in range(30):
            alpha, beta = g.alpha_betas()      # what this iteration will use: init_alpha_betas(current excluded)
            mask = g.loci_mask()
            s = g.em_iteration(5.0)
            cg = g.cell_outputs()
            ll_o, ell_o, nl_o = o.cell_log_likelihoods(alpha, beta, mask)
            np.testing.assert_allclose(cg["ll"][sel], ll_o, rtol=0, atol=LL_ATOL)
            np.testing.assert_allclose(cg["expected_ll"][sel], ell_o, rtol=0, atol=LL_ATOL)
            assert np.array_equal(cg["loci_used"][sel], nl_o)
            # threshold from exact order statistics of ALL cells (numpy sort as the independent check)
            srt = np.sort(cg["normalized"])
            k = N // 2
            med = srt[k] if N % 2 else (srt[k - 1] + srt[k]) / 2.0
            h1, h3 = (N + 1.0 / 3.0) * 0.25 + 1.0 / 3.0, (N + 1.0 / 3.0) * 0.75 + 1.0 / 3.0
            q1 = srt[int(h1) - 1] + (h1 - int(h1)) * (srt[int(h1)] - srt[int(h1) - 1])
            q3 = srt[int(h3) - 1] + (h3 - int(h3)) * (srt[int(h3)] - srt[int(h3) - 1])
            assert (s.median, s.iqr, s.threshold) == (med, q3 - q1, q1 - 5.0 * (q3 - q1))
            ex = g.excluded()
            assert np.array_equal(ex, (cg["normalized"] < s.threshold).astype(np.uint8))
            assert s.n_new_excluded == int(((ex == 1) & (prev_excluded == 0)).sum())
            assert s.n_rescued == int(((ex == 0) & (prev_excluded == 1)).sum())
            prev_excluded = ex
            # conservation: the cell pass and the locus pass saw the same log-pmfs and the same entries
            lg = g.locus_outputs()
            tot = cg["ll"].sum()
            assert abs(tot - (lg["contrib_min"].sum() + lg["contrib_maj"].sum())) < 1e-9 * abs(tot)
            assert int(lg["cells_min"].sum() + lg["cells_maj"].sum()) == int(cg["loci_used"].sum())
            if not s.any_change:
                break
        assert not s.any_change and it >= 1
        # -- idempotence at the fixed point
        s2 = g.em_iteration(5.0)
        assert not s2.any_change and np.array_equal(g.excluded(), ex) and s2.threshold == s.threshold
        cls = synth.cell_classes(N, seed=4, minority_fraction=0.05)
        assert (ex[cls == 1] == 1).mean() > 0.95 and (ex[cls == 0] == 1).mean() < 0.01
        # -- posteriors of the sample: reference formulas on the oracle's log-likelihoods
        lg = g.locus_outputs()
        pg = g.posteriors()
        n_exc = int(ex.sum())
        # the minority tallies that feed the posterior alpha/betas cover masked loci too (main.rs:598-611 ignores
        # the mask): take them from alpha_betas(), which is init_alpha_betas(excluded) on the device
        a_next, b_next = g.alpha_betas()
        alt_min, ref_min = (lc[:, 1] + 1.0) - a_next, (lc[:, 0] + 1.0) - b_next
        (a1, b1), (a2, b2), (a3, b3), (lp_min, lp_maj, lp_dbl) = _posterior_alpha_betas(lc, alt_min, ref_min, n_exc, N)
        l_min = o.cell_log_likelihoods(a1, b1)[0]
        l_maj = o.cell_log_likelihoods(a2, b2)[0]
        l_dbl = o.cell_log_likelihoods(a3, b3)[0]
        np.testing.assert_allclose(pg["ll_minority"][sel], l_min, rtol=0, atol=LL_ATOL)
        np.testing.assert_allclose(pg["ll_majority"][sel], l_maj, rtol=0, atol=LL_ATOL)
        num = lp_min + l_min
        den = _lse(_lse(num, lp_maj + l_maj), lp_dbl + l_dbl)
        np.testing.assert_allclose(pg["posterior"][sel], np.exp(num - den), rtol=0, atol=POST_ATOL)
        np.testing.assert_allclose(pg["doublet_posterior"][sel], np.exp(lp_dbl + l_dbl - den), rtol=0, atol=POST_ATOL)
    finally:
        ob.set_threads(1)
        o.close()
    epc = g.entries_per_cell()
    pa2 = ffi.assignments(pg["posterior"], pg["doublet_posterior"], epc, ex)[0]
    g.close()

    # -- the second, independent engine (every entry evaluated by the CSR/CSC kernels) on ALL cells
    g1 = _load(1, L, N, 0.01, deep=deep)
    s1 = g1.run(5.0, 30)
    assert len(s1) == it + 1 and not s1[-1].any_change
    assert abs(s1[-1].threshold - s.threshold) < 1e-9
    assert np.array_equal(g1.excluded(), ex)
    np.testing.assert_allclose(g1.cell_outputs()["ll"], cg["ll"], rtol=0, atol=LL_ATOL)
    p1 = g1.posteriors()
    np.testing.assert_allclose(p1["posterior"], pg["posterior"], rtol=0, atol=POST_ATOL)
    np.testing.assert_allclose(p1["doublet_posterior"], pg["doublet_posterior"], rtol=0, atol=POST_ATOL)
    pa1 = ffi.assignments(p1["posterior"], p1["doublet_posterior"], epc, ex)[0]
    assert np.array_equal(pa1, pa2)
    g1.close()


# ---- a shard above 2^20 cells ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("engine", [2, 1], ids=["tiled", "csr"])
def test_shard_above_2p20_cells(engine, oracle_lib, hip_lib_path):
    from cellector_amd import ffi
    ob = oracle_lib
    N, L = (1 << 20) + 70_001, 40
    g = _load(engine, L, N, 0.06, seed=8, minority=0.06)
    dm = g.dims()
    rp, ent = g.csr_rows(0, N)
    o = ob.Oracle.from_csr(dm.loci_used, rp, ent, g.locus_counts())
    ob.set_threads(ob.host_threads())
    try:
        for _ in range(12):
            sg, so = g.em_iteration(5.0), o.em_iteration(5.0)
            cg, co = g.cell_outputs(), o.cell_outputs()
            np.testing.assert_allclose(cg["ll"], co["ll"], rtol=0, atol=LL_ATOL)
            assert np.array_equal(cg["loci_used"], co["loci_used"])
            near = np.abs(co["normalized"] - so.threshold) < 1e-9
            # the threshold itself sits on a value of this very discrete distribution: cells AT the threshold are not
            # excluded (strict <) on either side, so only cells strictly within 1e-9 but not equal could flip
            assert not (near & (co["normalized"] != so.threshold)).any()
            np.testing.assert_allclose([sg.median, sg.iqr, sg.threshold], [so.median, so.iqr, so.threshold], rtol=0, atol=1e-9)
            assert np.array_equal(g.excluded(), o.excluded())
            lg, lo_ = g.locus_outputs(), o.locus_outputs()
            for k in ("cells_min", "cells_maj", "alt_min", "ref_min", "alt_maj", "ref_maj"):
                assert np.array_equal(lg[k], lo_[k]), k
            np.testing.assert_allclose(lg["contrib_min"], lo_["contrib_min"], rtol=0, atol=1e-6)
            if not so.any_change:
                break
        pg, po = g.posteriors(), o.posteriors()
        np.testing.assert_allclose(pg["posterior"], po["posterior"], rtol=0, atol=POST_ATOL)
    finally:
        ob.set_threads(1)
        o.close()
    g.close()
