"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (north_star: assignments bit-exact, posteriors within 1e-6):
  * integer / index / flag outputs: exact;
  * per-cell log-likelihood sums: abs 1e-7 (the oracle's own ln_gamma-difference cancellation noise is
    ~1e-11 per entry; the device uses the exact product form, see csrc/device_math.h);
  * posteriors: abs 1e-6.
Cells whose normalised LL lies within 1e-9 of the threshold would be reported as near-ties (none occur on
these inputs; the test asserts that so a flip can never hide).
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LL_ATOL = 1e-7
POST_ATOL = 1e-6


@pytest.fixture(scope="module", params=[2, 1], ids=["tiled", "csr"])
def mods(request, oracle_lib, hip_lib_path):
    """Every parity test runs against both engines: 2 = table-driven tiled passes (default), 1 = CSR/CSC kernels."""
    from cellector_amd import Cellector, ffi, synth
    engine = request.param

    def make(device=0):
        g = Cellector(device)
        g.set_option("engine", engine)
        return g

    return dict(Cellector=make, ffi=ffi, synth=synth, ob=oracle_lib, engine=engine)


def _case(mods, L, N, d, seed=4, minority=0.05, doublet=0.0, min_alt=4, min_ref=4):
    lo, ce, al, re = mods["synth"].generate_coo(L, N, d, seed=seed, minority_fraction=minority,
                                                 doublet_fraction=doublet)
    g = mods["Cellector"](0)
    g.load_coo(L, N, lo, ce, al, re, min_alt, min_ref)
    o = mods["ob"].Oracle.from_coo(L, N, lo, ce, al, re, min_alt, min_ref)
    return g, o, (lo, ce, al, re)


def _check_matrix(g, o):
    d = g.dims()
    assert (d.total_cells, d.total_loci, d.loci_used, d.nnz_used) == (o.total_cells, o.total_loci, o.loci_used, o.nnz)
    assert np.array_equal(g.locus_ids(), o.locus_ids())
    assert np.array_equal(g.locus_counts(), o.locus_counts())
    assert np.array_equal(g.entries_per_cell(), o.entries_per_cell())
    rp, ent = g.csr_rows(0, o.total_cells)
    assert np.array_equal(rp, o.row_ptr())
    li, a, r, _ = o.entries()
    assert np.array_equal(ent & np.uint64(0xFFFFFFFF), li.astype(np.uint64))
    assert np.array_equal((ent >> np.uint64(32)) & np.uint64(0xFFFF), a.astype(np.uint64))
    assert np.array_equal(ent >> np.uint64(48), r.astype(np.uint64))


def _check_iteration(g, o, sg, so):
    cg, co = g.cell_outputs(), o.cell_outputs()
    assert np.array_equal(cg["loci_used"], co["loci_used"])
    np.testing.assert_allclose(cg["ll"], co["ll"], rtol=0, atol=LL_ATOL)
    np.testing.assert_allclose(cg["expected_ll"], co["expected_ll"], rtol=0, atol=LL_ATOL)
    np.testing.assert_allclose(cg["normalized"], co["normalized"], rtol=0, atol=1e-9)
    np.testing.assert_allclose([sg.median, sg.iqr, sg.threshold], [so.median, so.iqr, so.threshold], rtol=0, atol=1e-9)
    # the product's own near-tie report (cellector_iter_summary.n_near_threshold): the same IEEE expression on the
    # device's values must give the same count, and any cell the oracle sees on the threshold must be reported
    tol = 1e-9 * max(1.0, abs(sg.threshold))
    assert sg.n_near_threshold == int((np.abs(cg["normalized"] - sg.threshold) <= tol).sum())
    if (np.abs(co["normalized"] - so.threshold) < 0.9e-9).any():
        assert sg.n_near_threshold > 0, "the oracle sees cells on the threshold, the product reported none"
    near = np.abs(co["normalized"] - so.threshold) < 1e-9
    assert not near.any(), f"near-tie cells at the threshold: {np.nonzero(near)[0]}"
    assert sg.n_near_threshold == 0
    assert np.array_equal(g.excluded(), o.excluded())
    assert (sg.any_change, sg.n_new_excluded, sg.n_rescued) == (so.any_change, so.n_new_excluded, so.n_rescued)
    assert sg.n_excluded == int(o.excluded().sum())
    lg, lo_ = g.locus_outputs(), o.locus_outputs()
    for k in ("cells_min", "cells_maj", "alt_min", "ref_min", "alt_maj", "ref_maj"):
        assert np.array_equal(lg[k], lo_[k]), k
    # (sums over up to all cells of a locus: the oracle's own ln_gamma-difference error — ~1e-10 per entry at alpha+beta ~ 1e5, one-signed — grows with the sum: rtol)
    np.testing.assert_allclose(lg["contrib_min"], lo_["contrib_min"], rtol=1e-8, atol=LL_ATOL)
    np.testing.assert_allclose(lg["contrib_maj"], lo_["contrib_maj"], rtol=1e-8, atol=1e-6)
    assert np.array_equal(g.loci_mask(), o.loci_mask())
    assert sg.n_loci_filtered == so.n_loci_filtered
    ag, bg = g.alpha_betas()
    ao, bo = o.alpha_betas()
    assert np.array_equal(ag, ao) and np.array_equal(bg, bo)


def _run_both(g, o, max_iter=30):
    n = 0
    while True:
        sg, so = g.em_iteration(5.0), o.em_iteration(5.0)
        _check_iteration(g, o, sg, so)
        n += 1
        if not so.any_change or n >= max_iter:
            return n


def _check_posteriors(mods, g, o):
    pg, po = g.posteriors(), o.posteriors()
    for k in ("ll_majority", "ll_minority"):
        np.testing.assert_allclose(pg[k], po[k], rtol=0, atol=LL_ATOL)
    np.testing.assert_allclose(pg["posterior"], po["posterior"], rtol=0, atol=POST_ATOL)
    np.testing.assert_allclose(pg["doublet_posterior"], po["doublet_posterior"], rtol=0, atol=POST_ATOL)
    ga = mods["ffi"].assignments(pg["posterior"], pg["doublet_posterior"], g.entries_per_cell(), g.excluded())
    oa = o.assignments(po["posterior"], po["doublet_posterior"])
    # bit-exact labels; qual may differ by one unit only where -10 log10(1-p) sits on an integer boundary
    assert np.array_equal(ga[0], oa[0]) and np.array_equal(ga[1], oa[1])
    assert np.max(np.abs(ga[2].astype(np.int64) - oa[2].astype(np.int64))) <= 1
    return pg, po


def test_synthetic_generator_matches_host_twin(mods):
    L, N, d = 300, 40000, 0.02  # spans three 16384-cell generator tiles
    lo, ce, al, re = mods["synth"].generate_coo(L, N, d, seed=11, minority_fraction=0.07, doublet_fraction=0.02)
    with mods["Cellector"](0) as g:
        g.load_synthetic(L, N, d, seed=11, minority_fraction=0.07, doublet_fraction=0.02, min_alt=0, min_ref=0)
        o = mods["ob"].Oracle.from_coo(L, N, lo, ce, al, re, 0, 0)
        _check_matrix(g, o)
        info = g.engine_info()
        assert info.engine == mods["engine"]
        if mods["engine"] == 2:
            assert info.nnz_regular + info.nnz_overflow == o.nnz and 0 < info.nnz_overflow < 0.06 * o.nnz
            assert info.cell_blocks == 40 and info.locus_chunks == 1


def test_deep_coverage_generator_matches_host_twin(mods):
    """option synth_continue_pct (bench.py's *-deep workloads): totals 1 + Geometric(0.4) instead of 1 + Geometric(0.7);
    the device generator and its numpy twin must still agree entry for entry, and the run must match the oracle (13 % of
    the entries take the overflow paths here instead of 0.8 %)."""
    L, N, d = 500, 3000, 0.05
    lo, ce, al, re = mods["synth"].generate_coo(L, N, d, seed=13, minority_fraction=0.08, continue_pct=60)
    assert (al + re).max() > 12 and ((al + re) > 4).mean() > 0.08
    g = mods["Cellector"](0)
    g.set_option("synth_continue_pct", 60)
    g.load_synthetic(L, N, d, seed=13, minority_fraction=0.08)
    o = mods["ob"].Oracle.from_coo(L, N, lo, ce, al, re)
    _check_matrix(g, o)
    _run_both(g, o)
    _check_posteriors(mods, g, o)
    g.close(); o.close()


def test_cfg1_full_loop_and_posteriors(mods):
    g, o, _ = _case(mods, 2000, 1000, 0.10)
    _check_matrix(g, o)
    assert 0 < o.loci_used < 2000  # the min_ref/min_alt filter really drops loci here
    iters = _run_both(g, o)
    assert iters >= 2 and o.excluded().sum() > 0
    _check_posteriors(mods, g, o)
    g.close()


def test_doublets_and_small_minority(mods):
    g, o, _ = _case(mods, 3000, 1500, 0.08, seed=7, minority=0.1, doublet=0.03)
    _run_both(g, o)
    pg, po = _check_posteriors(mods, g, o)
    g.close()


def test_wide_compact_entries(mods):
    """The locus pass stores 24-bit entries for shards of up to 2^20 cells; force the 32-bit form used beyond that."""
    lo, ce, al, re = mods["synth"].generate_coo(1200, 900, 0.1, seed=21, minority_fraction=0.09)
    g = mods["Cellector"](0)
    g.set_option("compact_bits", 32)
    g.load_coo(1200, 900, lo, ce, al, re)
    o = mods["ob"].Oracle.from_coo(1200, 900, lo, ce, al, re)
    _run_both(g, o)
    _check_posteriors(mods, g, o)
    g.close()


def test_locus_pass_forms_are_bit_identical(mods):
    """Engine 2 tallies the per-locus statistics either by streaming the compact CSC past the exclusion bitmask
    (locus_mode 1) or by walking only the excluded cells' rows into an integer histogram (locus_mode 2; mode 0 picks
    one on the device per iteration).  Same counts, same fixed-shape f64 sums: every output must agree to the bit,
    and each form must agree with the oracle."""
    if mods["engine"] != 2:
        pytest.skip("engine 2 option")
    lo, ce, al, re = mods["synth"].generate_coo(1500, 1100, 0.1, seed=33, minority_fraction=0.2)
    runs = []
    for mode in (1, 2, 0):
        g = mods["Cellector"](0)
        g.set_option("locus_mode", mode)
        g.load_coo(1500, 1100, lo, ce, al, re)
        o = mods["ob"].Oracle.from_coo(1500, 1100, lo, ce, al, re)
        outs = []
        for _ in range(30):
            sg, so = g.em_iteration(5.0), o.em_iteration(5.0)
            _check_iteration(g, o, sg, so)
            outs.append(g.locus_outputs())
            if not sg.any_change:
                break
        runs.append(outs)
        g.close()
        o.close()
    assert len(runs[0]) == len(runs[1]) == len(runs[2])
    for a, b, c in zip(*runs):
        for k in a:
            assert np.array_equal(a[k], b[k]), k
            assert np.array_equal(a[k], c[k]), k


def test_overflow_side_stream_equals_in_stream(mods):
    """Engine 2 runs the kernels of the entries with alt+ref = 0 or > 4 on a side stream beside the table-lookup kernel
    (option overlap, default 1) — same kernels, same arithmetic as in one stream: every output must agree to the bit.
    Counts are drawn wide here so that a good part of the entries takes that path, some of them the rare-total kernel."""
    if mods["engine"] != 2:
        pytest.skip("engine 2 option")
    rng = np.random.default_rng(11)
    L, N = 700, 1300
    lo, ce, al, re = mods["synth"].generate_coo(L, N, 0.1, seed=8, minority_fraction=0.1)
    big = rng.random(len(al)) < 0.2
    al = np.where(big, al * rng.integers(1, 12, len(al)), al).astype(np.uint32)
    re = np.where(big, re * rng.integers(1, 12, len(re)), re).astype(np.uint32)
    outs = []
    # beside the lookup kernel (default) / in one stream / locus-side values after the lookup kernel / the side kernels under
    # the residency throttle a big shard gets (which also selects the packed-record form of the cell-side kernel)
    # ... and, last, the form a matrix with this many such entries gets by itself (ovf_deep: one unthrottled kernel for
    # totals up to 17): same values within rounding — the summation order inside a row's overflow entries differs
    # (the deep form has two kernels: 16 lanes per row — the default — and a thread per row)
    # (t2: the entries with totals 5..8 through per-(locus, pair) tables — the default unless the matrix is deep — or, 0, one by
    # one like the other totals: same values within rounding)
    # (t2_tiles: the deep form's cell side of the totals 5..8 — the default — or 5..6 through a second tile set with chunk tables
    # in LDS, or, 0, evaluated one by one by the 16-lanes-per-row kernel like the other totals: same values within rounding)
    variants = ((1, -1, 0, 1, -1, -1), (0, -1, 0, 1, -1, -1), (2, -1, 0, 1, -1, -1), (1, 5000, 0, 1, -1, -1), (1, -1, 0, 1, 0, -1),
                (1, -1, -1, 1, -1, -1), (0, -1, 1, 1, -1, -1), (1, -1, 1, 0, -1, -1), (1, -1, 1, 1, 1, -1), (0, -1, 1, 1, 1, -1),
                (1, -1, 1, 1, -1, 0), (1, -1, 1, 1, -1, 6), (0, -1, 1, 1, 1, 6), (1, -1, 1, 1, 0, 8))
    for ov, lds, deep, wide, t2, tt in variants:
        g = mods["Cellector"](0)
        g.set_option("overlap", ov)
        g.set_option("side_lds", lds)
        g.set_option("ovf_deep", deep)
        g.set_option("ovf_deep_wide", wide)
        g.set_option("t2", t2)
        g.set_option("t2_tiles", tt)
        g.load_coo(L, N, lo, ce, al, re)
        o = mods["ob"].Oracle.from_coo(L, N, lo, ce, al, re)
        run = []
        for _ in range(30):
            sg, so = g.em_iteration(5.0), o.em_iteration(5.0)
            _check_iteration(g, o, sg, so)
            run.append((g.cell_outputs(), g.locus_outputs()))
            if not sg.any_change:
                break
        pg, _ = _check_posteriors(mods, g, o)
        outs.append((run, pg))
        g.close()
        o.close()
    run_a, post_a = outs[0]
    for vi, (run_b, post_b) in enumerate(outs[1:], 1):
        assert len(run_a) == len(run_b)
        same = (lambda x, y: np.array_equal(x, y)) if vi < 4 else (lambda x, y: np.allclose(x, y, rtol=0, atol=1e-9))
        for (ca, la), (cb, lb) in zip(run_a, run_b):
            for k in ca:
                assert same(ca[k], cb[k]), (vi, k)
            for k in la:
                assert same(la[k], lb[k]), (vi, k)
        for k in post_a:
            assert same(post_a[k], post_b[k]), (vi, k)
    # the deep-form runs beside the lookup kernel / in one stream agree to the bit, without and with tier-2 tables
    for i, j in ((5, 6), (8, 9), (11, 12)):
        for (ca, la), (cb, lb) in zip(outs[i][0], outs[j][0]):
            for k in ca:
                assert np.array_equal(ca[k], cb[k]), (i, j, k)
            for k in la:
                assert np.array_equal(la[k], lb[k]), (i, j, k)


def test_device_text_writer_round_trip(mods, tmp_path):
    """cellector_write_staged_mtx (benchmark utility): the text pair it formats on the device loads back — through the
    device tokeniser and through the oracle's loader — as the very matrix it was written from."""
    if mods["engine"] != 2:
        pytest.skip("one engine is enough")
    L, N = 900, 40000  # several generator tiles; line lengths from 6 to 13 bytes
    g = mods["Cellector"](0)
    g.set_option("keep_coo", 1)
    g.ingest_synthetic(L, N, 0.02, seed=9)
    alt, ref = str(tmp_path / "alt.mtx"), str(tmp_path / "ref.mtx")
    g.write_staged_mtx(alt, ref)
    g.ingest_finish(4, 4)
    head = open(alt).read(200).splitlines()[:3]
    assert head[0] == "%%MatrixMarket matrix coordinate real general" and head[2].split()[:2] == [str(L), str(N)]
    h = mods["Cellector"](0)
    h.load_mtx(alt, ref, 4, 4)
    o = mods["ob"].Oracle.from_mtx(alt, ref, 4, 4)
    _check_matrix(h, o)
    dg, dh = g.dims(), h.dims()
    assert (dg.loci_used, dg.nnz_used) == (dh.loci_used, dh.nnz_used)
    rg, eg = g.csr_rows(0, N)
    rh, eh = h.csr_rows(0, N)
    assert np.array_equal(rg, rh) and np.array_equal(eg, eh)
    g.close(); h.close(); o.close()


def test_ll_pass_under_caller_alpha_beta_and_mask(mods):
    g, o, _ = _case(mods, 1500, 800, 0.1, seed=3)
    rng = np.random.default_rng(0)
    L = o.loci_used
    alpha = rng.uniform(0.5, 5000.0, L)   # non-integer alpha/beta like the posterior phase
    beta = rng.uniform(0.5, 5000.0, L)
    mask = (rng.random(L) > 0.3).astype(np.uint8)
    for m in (None, mask):
        lg = g.cell_log_likelihoods(alpha, beta, m)
        lo_ = o.cell_log_likelihoods(alpha, beta, m)
        np.testing.assert_allclose(lg[0], lo_[0], rtol=0, atol=LL_ATOL)
        np.testing.assert_allclose(lg[1], lo_[1], rtol=0, atol=LL_ATOL)
        assert np.array_equal(lg[2], lo_[2])
    g.close()


def test_edge_counts_and_empty_cells(mods):
    """Hand-built matrix: zero/zero entries (quirk Q14), counts far above the 8-factor fast path, totals above
    the 170-entry factorial cache, cells without entries (quirk Q4), unsorted input order."""
    rng = np.random.default_rng(5)
    L, N = 40, 300
    lo, ce, al, re = [], [], [], []
    for l in range(L):
        for c in range(N):
            if c % 7 == 3:          # cells with no entries at all
                continue
            if rng.random() < 0.5:
                big = rng.random() < 0.15
                a = int(rng.integers(0, 120 if big else 4))
                r = int(rng.integers(0, 120 if big else 4))
                lo.append(l); ce.append(c); al.append(a); re.append(r)
    lo, ce, al, re = map(lambda x: np.array(x, np.uint32), (lo, ce, al, re))
    al[5], re[5] = 0, 0
    al[17], re[17] = 150, 90       # n = 240 > 170: ln_gamma branch of ln_factorial
    perm = rng.permutation(len(lo))  # file order not sorted by locus
    lo, ce, al, re = lo[perm], ce[perm], al[perm], re[perm]
    g = mods["Cellector"](0)
    g.load_coo(L, N, lo, ce, al, re, 1, 1)
    o = mods["ob"].Oracle.from_coo(L, N, lo, ce, al, re, 1, 1)
    d = g.dims()
    assert (d.loci_used, d.nnz_used) == (o.loci_used, o.nnz)
    assert np.array_equal(g.locus_ids(), o.locus_ids())
    assert np.array_equal(g.locus_counts(), o.locus_counts())
    assert np.array_equal(g.entries_per_cell(), o.entries_per_cell())
    assert (g.entries_per_cell() == 0).sum() >= N // 7
    _run_both(g, o, max_iter=6)
    _check_posteriors(mods, g, o)
    g.close()


def test_near_ties_are_reported(mods):
    """Constructed tie: every cell carries the same entries, so all normalised LLs are equal, Q1 = Q3, and the threshold
    sits exactly on every cell (strict `<`, main.rs:330-332: nobody is excluded).  The reference and the device differ
    by ~1e-11 there, so which side such a cell falls on is not covered by the parity claim: the summary must say so."""
    L, N = 40, 50
    lo = np.repeat(np.arange(L, dtype=np.uint32), N)
    ce = np.tile(np.arange(N, dtype=np.uint32), L)
    al = np.ones(L * N, np.uint32)
    re = np.ones(L * N, np.uint32)
    g = mods["Cellector"](0)
    g.load_coo(L, N, lo, ce, al, re)
    o = mods["ob"].Oracle.from_coo(L, N, lo, ce, al, re)
    sg, so = g.em_iteration(5.0), o.em_iteration(5.0)
    assert sg.iqr == 0.0 and sg.n_excluded == 0 and not sg.any_change
    assert sg.n_near_threshold == N
    assert abs(sg.threshold - so.threshold) < 1e-9
    # one cell pushed well below: it is excluded and no longer near; the others still tie with the threshold
    al2 = al.copy(); re2 = re.copy()
    al2[ce == 7] = 4; re2[ce == 7] = 0
    g.load_coo(L, N, lo, ce, al2, re2)
    sg = g.em_iteration(5.0)
    assert sg.n_excluded == 1 and g.excluded()[7] == 1 and sg.n_near_threshold == N - 1
    g.close(); o.close()


def test_context_reload_starts_clean(mods):
    """A ctx that ran iterations on one matrix and is then loaded with another must not carry tables built ahead for the
    old one (em_finish queues the next iteration's k_build_tables): the second run must equal a fresh ctx and the oracle."""
    la, ca, aa, ra = mods["synth"].generate_coo(900, 700, 0.12, seed=5, minority_fraction=0.1)
    lb, cb, ab_, rb = mods["synth"].generate_coo(1300, 500, 0.1, seed=6, minority_fraction=0.07)
    g = mods["Cellector"](0)
    g.load_coo(900, 700, la, ca, aa, ra)
    for _ in range(3):
        g.em_iteration(5.0)                      # leaves tables prebuilt for iteration 4 of matrix A
    g.load_coo(1300, 500, lb, cb, ab_, rb)      # same ctx, new matrix
    f = mods["Cellector"](0)
    f.load_coo(1300, 500, lb, cb, ab_, rb)
    o = mods["ob"].Oracle.from_coo(1300, 500, lb, cb, ab_, rb)
    _check_matrix(g, o)
    for _ in range(30):
        sg, sf, so = g.em_iteration(5.0), f.em_iteration(5.0), o.em_iteration(5.0)
        _check_iteration(g, o, sg, so)
        cg, cf = g.cell_outputs(), f.cell_outputs()
        for k in cg:
            assert np.array_equal(cg[k], cf[k]), k
        assert (sg.threshold, sg.n_excluded) == (sf.threshold, sf.n_excluded)
        if not so.any_change:
            break
    _check_posteriors(mods, g, o)
    g.close(); f.close(); o.close()


def test_locus_filter_triggers(mods):
    """A locus where the minority population is fixed for the other allele with deep coverage contributes far
    below -80 per minority cell and must be masked from the next iteration on (main.rs:444-447)."""
    L, N = 400, 600
    lo, ce, al, re = mods["synth"].generate_coo(L, N, 0.25, seed=9, minority_fraction=0.1)
    cls = mods["synth"].cell_classes(N, seed=9, minority_fraction=0.1)
    extra_l = np.full(N, L, np.uint32)            # one extra, pathological locus
    extra_c = np.arange(N, dtype=np.uint32)
    extra_a = np.where(cls == 1, 60, 0).astype(np.uint32)
    extra_r = np.where(cls == 1, 0, 60).astype(np.uint32)
    lo = np.concatenate([lo, extra_l]); ce = np.concatenate([ce, extra_c])
    al = np.concatenate([al, extra_a]); re = np.concatenate([re, extra_r])
    g = mods["Cellector"](0)
    g.load_coo(L + 1, N, lo, ce, al, re)
    o = mods["ob"].Oracle.from_coo(L + 1, N, lo, ce, al, re)
    _run_both(g, o)
    assert (o.loci_mask() == 0).sum() >= 1
    _check_posteriors(mods, g, o)
    g.close()


def test_mtx_files_plain_and_gz(mods, tmp_path):
    L, N = 500, 400
    lo, ce, al, re = mods["synth"].generate_coo(L, N, 0.15, seed=2)
    for gz in (False, True):
        a_path, r_path = mods["synth"].write_mtx_pair(str(tmp_path / ("gz" if gz else "plain")), L, N, lo, ce, al, re,
                                                      gz=gz, header_nnz=0)
        g = mods["Cellector"](0)
        g.load_mtx(a_path, r_path)
        o = mods["ob"].Oracle.from_mtx(a_path, r_path)
        _check_matrix(g, o)
        sg, so = g.em_iteration(5.0), o.em_iteration(5.0)
        _check_iteration(g, o, sg, so)
        g.close()
    # block-compressed input (bgzip: every gzip member carries its own size — inflated block-parallel), the same file with a
    # plain gzip member appended (no index: the serial reader), and a damaged block (an error, never silent garbage)
    import gzip as _gzip
    a_plain, r_plain = str(tmp_path / "plain" / "alt.mtx"), str(tmp_path / "plain" / "ref.mtx")
    os.makedirs(tmp_path / "bgzf", exist_ok=True)
    o = mods["ob"].Oracle.from_mtx(a_plain, r_plain)
    for variant in ("bgzf", "mixed"):
        paths = []
        for src in (a_plain, r_plain):
            text = open(src, "rb").read()
            cut = len(text) * 2 // 3 if variant == "mixed" else len(text)
            blob = mods["synth"].bgzf_compress(text[:cut], block=4000 if variant == "bgzf" else 0xff00, eof_block=variant == "bgzf")
            if variant == "mixed":
                blob += _gzip.compress(text[cut:])
            dst = str(tmp_path / "bgzf" / (variant + "_" + os.path.basename(src) + ".gz"))
            open(dst, "wb").write(blob)
            assert _gzip.decompress(blob) == text  # (a valid multi-member gzip either way)
            paths.append(dst)
        g = mods["Cellector"](0)
        g.load_mtx(paths[0], paths[1])
        _check_matrix(g, o)
        g.close()
    blob = bytearray(open(str(tmp_path / "bgzf" / "bgzf_alt.mtx.gz"), "rb").read())
    blob[len(blob) // 2] ^= 0x55
    open(str(tmp_path / "bgzf" / "broken_alt.mtx.gz"), "wb").write(bytes(blob))
    g = mods["Cellector"](0)
    with pytest.raises(mods["ffi"].CellectorError):
        g.load_mtx(str(tmp_path / "bgzf" / "broken_alt.mtx.gz"), str(tmp_path / "bgzf" / "bgzf_ref.mtx.gz"))
    g.close()
    g = mods["Cellector"](0)
    with pytest.raises(mods["ffi"].CellectorError) as ei:
        g.load_mtx(str(tmp_path / "missing.mtx"), str(tmp_path / "missing.mtx"))
    assert "couldn't open file" in str(ei.value)
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real general\n%\n3 3 1\n1 1 1.0\n")
    with pytest.raises(mods["ffi"].CellectorError):
        g.load_mtx(str(bad), str(bad))
    g.close()


# ---- two shards on one GPU, exchange done by hand through the host ------------------------------------------
def _hip():
    lib = C.CDLL("libamdhip64.so")
    lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.hipMemcpy.restype = C.c_int
    return lib


def _allreduce(hip, shards, which):
    bufs = [s.exchange_buffer(which) for s in shards]
    n = bufs[0][1]
    tot = np.zeros(n, np.float64)
    tmp = np.empty(n, np.float64)
    for ptr, _ in bufs:
        assert hip.hipMemcpy(tmp.ctypes.data, ptr, n * 8, 2) == 0  # D2H
        tot += tmp
    for ptr, _ in bufs:
        assert hip.hipMemcpy(ptr, tot.ctypes.data, n * 8, 1) == 0  # H2D


def test_two_shards_equal_single_shard(mods):
    """(Bit for bit: with option bank_order = 0 the entries of a row keep their file order, so a cell's sum does not depend on
    which cells share its shard.  The default order — chosen against LDS bank conflicts per 32 rows — makes sums of differently
    sharded runs differ in the last bits: tests/test_gpu_sharded.py::test_nnz_balanced_partition_on_a_skewed_matrix.)"""
    L, N, d = 1500, 1201, 0.1
    lo, ce, al, re = mods["synth"].generate_coo(L, N, d, seed=4, minority_fraction=0.06, doublet_fraction=0.01)
    hip = _hip()
    single = mods["Cellector"](0)
    single.set_option("bank_order", 0)
    single.load_coo(L, N, lo, ce, al, re)
    cut = 523
    shards = [mods["Cellector"](0), mods["Cellector"](0)]
    for s in shards:
        s.set_option("bank_order", 0)
    shards[0].set_shard(0, cut)
    shards[1].set_shard(cut, N)
    for s in shards:
        s.ingest_coo(L, N, lo, ce, al, re)
    _allreduce(hip, shards, mods["ffi"].XCHG_PASS1)
    for s in shards:
        s.ingest_finish()
    assert shards[0].dims().loci_used == single.dims().loci_used
    assert shards[0].dims().nnz_used + shards[1].dims().nnz_used == single.dims().nnz_used
    for it in range(30):
        s1 = single.em_iteration(5.0)
        for s in shards:
            s.em_begin()
        _allreduce(hip, shards, mods["ffi"].XCHG_NORM)
        for s in shards:
            s.em_threshold(5.0)
        _allreduce(hip, shards, mods["ffi"].XCHG_LOCUS)
        ss = [s.em_finish() for s in shards]
        for s2 in ss:
            assert (s2.any_change, s2.n_new_excluded, s2.n_rescued, s2.n_excluded, s2.n_loci_filtered) == \
                   (s1.any_change, s1.n_new_excluded, s1.n_rescued, s1.n_excluded, s1.n_loci_filtered)
            assert (s2.median, s2.iqr, s2.threshold) == (s1.median, s1.iqr, s1.threshold)
        ex = np.concatenate([s.excluded() for s in shards])
        assert np.array_equal(ex, single.excluded())
        c1 = single.cell_outputs()
        for k in c1:
            assert np.array_equal(np.concatenate([s.cell_outputs()[k] for s in shards]), c1[k]), k
        l1, l2 = single.locus_outputs(), shards[1].locus_outputs()
        for k in l1:
            if k.startswith("contrib"):
                np.testing.assert_allclose(l2[k], l1[k], rtol=0, atol=1e-8)
            else:
                assert np.array_equal(l2[k], l1[k]), k
        if not s1.any_change:
            break
    p1 = single.posteriors()
    p2 = [s.posteriors() for s in shards]
    for k in p1:
        assert np.array_equal(np.concatenate([p[k] for p in p2]), p1[k]), k
    for s in shards + [single]:
        s.close()


def test_order_statistics_on_adversarial_keys(mods):
    """The device order statistics (csrc/kernels_select.hip) against the oracle's restatement of statrs Data::median /
    quantile on keys no EM run produces: the NORM exchange buffer is overwritten between em_begin and em_threshold, the
    summary's median / iqr / threshold must be bit-identical (ties, one shared 22-bit prefix, huge range, signs, tiny n)."""
    if mods["engine"] != 2:
        pytest.skip("one select implementation; run once")
    hip, ob, ffi = _hip(), mods["ob"], mods["ffi"]
    rng = np.random.default_rng(7)
    cases = [
        (1, lambda n: np.array([-0.25])),
        (2, lambda n: np.array([3.0, -1.0])),
        (3, lambda n: np.array([0.0, -0.0, -1e-310])),
        (4, lambda n: np.array([-1.0, -1.0, -1.0, -2.0])),
        (5000, lambda n: np.full(n, -0.5)),                                      # every key in every step's one bin
        (5000, lambda n: np.where(rng.random(n) < 0.5, -0.5, -0.75)),            # two heavy ties
        (40000, lambda n: -0.5 - np.arange(n) * 1e-13),                          # one 22-bit prefix, all distinct
        (40000, lambda n: -rng.uniform(0.25, 1.0, n)),
        (40000, lambda n: rng.standard_normal(n) * 10.0 ** rng.integers(-300, 300, n)),   # every exponent, both signs
        (40000, lambda n: np.round(rng.standard_normal(n), 1)),                  # ~80 distinct values
        (300000, lambda n: -np.exp(rng.standard_normal(n) * 0.2) * 0.5),
        (300001, lambda n: np.concatenate([np.zeros(n // 3), -rng.uniform(0.4, 0.6, n - n // 3)])),
    ]
    for n, make in cases:
        L = 40
        lo, ce, al, re = mods["synth"].generate_coo(L, n, 0.3 if n < 100000 else 0.05, seed=11)
        if len(lo) == 0:
            lo, ce = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
            al, re = np.ones(1, np.uint32), np.ones(1, np.uint32)
        g = mods["Cellector"](0)
        g.load_coo(L, n, lo, ce, al, re, 0, 0)
        g.em_begin()
        ptr, m = g.exchange_buffer(ffi.XCHG_NORM)
        assert m >= n
        keys = np.ascontiguousarray(make(n), dtype=np.float64)
        assert len(keys) == n and not np.isnan(keys).any()
        assert hip.hipMemcpy(ptr, keys.ctypes.data, n * 8, 1) == 0
        g.em_threshold(5.0)
        s = g.em_finish()
        q1, q3 = ob.quantile(keys, 0.25), ob.quantile(keys, 0.75)
        assert s.median == ob.median(keys), n
        assert s.iqr == q3 - q1, n
        assert s.threshold == q1 - 5.0 * (q3 - q1), n
        assert s.n_excluded == int((keys < s.threshold).sum()), n
        g.close()


def test_medium_matrix_properties_and_parity(mods):
    """20k cells x 20k loci at 1% (device-generated, ~4e6 entries): oracle parity for the first iteration and
    the size-independent properties the path offers."""
    L, N, d = 20000, 20000, 0.01
    g = mods["Cellector"](0)
    g.load_synthetic(L, N, d, seed=4, minority_fraction=0.05)
    dm = g.dims()
    rp, ent = g.csr_rows(0, N)
    lc = g.locus_counts()
    o = mods["ob"].Oracle.from_csr(dm.loci_used, rp, ent, lc)
    # conservation: the per-locus totals equal the column sums of the CSR
    li = (ent & np.uint64(0xFFFFFFFF)).astype(np.int64)
    alt = ((ent >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.float64)
    ref = (ent >> np.uint64(48)).astype(np.float64)
    assert np.array_equal(np.bincount(li, alt, dm.loci_used), lc[:, 1])
    assert np.array_equal(np.bincount(li, ref, dm.loci_used), lc[:, 0])
    sg, so = g.em_iteration(5.0), o.em_iteration(5.0)
    cg, co = g.cell_outputs(), o.cell_outputs()
    np.testing.assert_allclose(cg["ll"], co["ll"], rtol=0, atol=LL_ATOL)
    np.testing.assert_allclose(cg["expected_ll"], co["expected_ll"], rtol=0, atol=LL_ATOL)
    assert np.array_equal(g.excluded(), o.excluded())
    # the cell pass and the locus pass see the same log-pmfs: Σ_cells LL == Σ_loci (contrib_min + contrib_maj)
    lg = g.locus_outputs()
    assert abs(cg["ll"].sum() - (lg["contrib_min"].sum() + lg["contrib_maj"].sum())) < 1e-5
    assert int(lg["cells_min"].sum() + lg["cells_maj"].sum()) == dm.nnz_used
    # idempotence at the fixed point: once nothing changes, another iteration reproduces the same state
    summaries = g.run(5.0, max_iter=30)
    assert not summaries[-1].any_change
    ex = g.excluded().copy()
    s2 = g.em_iteration(5.0)
    assert not s2.any_change and np.array_equal(g.excluded(), ex)
    cls = mods["synth"].cell_classes(N, seed=4, minority_fraction=0.05)
    assert (ex[cls == 1] == 1).mean() > 0.95 and (ex[cls == 0] == 1).mean() < 0.01
    g.close()


def test_mtx_text_contract_on_device(mods, tmp_path):
    """The device tokeniser honours the reference's text contract (quirk Q11 and friends): dims from the REF header,
    indices from the ALT file only, nnz ignored, zip stops at the shorter file, CRLF / '+' / missing final newline
    accepted like split_whitespace + parse::<usize>(), anything else is a parse error with the entry number."""
    alt, ref = tmp_path / "alt.mtx", tmp_path / "ref.mtx"
    alt.write_text("%%MatrixMarket\n%\n99 99 99\n1 1 2\r\n1 2 +1\n2 1 0\n  2   2 3  \n2\t3\t1\n2 3 7")  # last line unterminated
    ref.write_text("%%MatrixMarket\n%\n2 3 0\nx y 1\n8 8 0\n9 9 4\n1 1 1\n7 7 2\n")                    # indices nonsense, one line shorter
    g = mods["Cellector"](0)
    g.load_mtx(str(alt), str(ref), 1, 1)
    o = mods["ob"].Oracle.from_mtx(str(alt), str(ref), 1, 1)
    d = g.dims()
    assert (d.total_loci, d.total_cells, d.loci_used, d.nnz_used) == (2, 3, o.loci_used, o.nnz) == (2, 3, 2, 5)
    assert np.array_equal(g.locus_counts(), o.locus_counts())
    rp, ent = g.csr_rows(0, 3)
    assert rp.tolist() == o.row_ptr().tolist() == [0, 2, 4, 5]
    g.close()
    cases = {
        "float": "1 1 1.0\n", "blank": "1 1 1\n\n1 2 1\n", "two_tokens": "1 1\n", "negative": "1 -1 1\n",
        "index0": "0 1 1\n", "locus_range": "3 1 1\n", "cell_range": "1 4 1\n", "count_range": "1 1 70000\n",
    }
    for name, body in cases.items():
        bad = tmp_path / f"{name}.mtx"
        bad.write_text("%%MatrixMarket\n%\n2 3 0\n" + body)
        g = mods["Cellector"](0)
        with pytest.raises(mods["ffi"].CellectorError) as ei:
            g.load_mtx(str(bad), str(bad), 1, 1)
        assert ei.value.status in (1, 3) and "entry" in str(ei.value), name
        g.close()
    hdr = tmp_path / "hdr.mtx"
    hdr.write_text("%%MatrixMarket\n%\n")
    g = mods["Cellector"](0)
    with pytest.raises(mods["ffi"].CellectorError, match="size line"):
        g.load_mtx(str(hdr), str(hdr), 1, 1)
    g.close()


def test_text_ingest_on_a_non_blocking_stream(mods, tmp_path):
    """cellector_set_stream documents torch's current stream; inside torch.cuda.stream(s) that is a NON-blocking stream,
    which null-stream copies do not order against.  The parser's validation (index 0, out-of-range indices, counts above
    65535) must be read behind its kernel on that stream: a malformed file must still be refused, a valid one must load
    to the same matrix as on the null stream — through the whole-file and the windowed path."""
    if mods["engine"] != 2:
        pytest.skip("ingest path: one engine is enough")
    import torch
    L, N = 600, 500
    lo, ce, al, re = mods["synth"].generate_coo(L, N, 0.08, seed=31)
    a_path, r_path = mods["synth"].write_mtx_pair(str(tmp_path), L, N, lo, ce, al, re)
    ref = mods["Cellector"](0)
    ref.load_mtx(a_path, r_path, 2, 2)
    rr, er = ref.csr_rows(0, N)
    s2 = ref.em_iteration(5.0)
    side = torch.cuda.Stream(device=0)
    many = "".join(f"1 {1 + i % 3} 1\n" for i in range(3000))
    bad_cases = {"index0": many + "0 1 1\n", "locus_range": many + "3 1 1\n", "cell_range": many + "1 4 1\n",
                 "count_range": many + "1 1 70000\n"}
    with torch.cuda.stream(side):
        for window in (0, 4096):
            g = mods["Cellector"](0)
            g.set_stream(torch.cuda.current_stream().cuda_stream)
            assert torch.cuda.current_stream().cuda_stream == side.cuda_stream != 0
            if window:
                g.set_option("parse_window", window)
            g.load_mtx(a_path, r_path, 2, 2)
            rg, eg = g.csr_rows(0, N)
            assert np.array_equal(rg, rr) and np.array_equal(eg, er)
            assert np.array_equal(g.locus_counts(), ref.locus_counts())
            s1 = g.em_iteration(5.0)
            assert (s1.threshold, s1.n_excluded) == (s2.threshold, s2.n_excluded)
            for name, body in bad_cases.items():
                bad = tmp_path / f"nb_{name}.mtx"
                bad.write_text("%%MatrixMarket\n%\n2 3 0\n" + body)
                with pytest.raises(mods["ffi"].CellectorError) as ei:
                    g.load_mtx(str(bad), str(bad), 1, 1)
                assert ei.value.status == 1 and "entry 3000" in str(ei.value), (name, str(ei.value))
            g.close()
    ref.close()


def test_windowed_text_parse_equals_whole_file_parse(mods, tmp_path):
    """Files of 4 GB and more are uploaded and tokenised in windows (csrc/kernels_parse.hip, parse_windowed); option
    parse_window forces that path with tiny windows here, so that lines straddle window ends everywhere: the staged matrix,
    the error reporting and the text contract must be those of the whole-file path."""
    if mods["engine"] != 2:
        pytest.skip("ingest path: one engine is enough")
    L, N = 700, 900
    lo, ce, al, re = mods["synth"].generate_coo(L, N, 0.05, seed=21)
    a_path, r_path = mods["synth"].write_mtx_pair(str(tmp_path), L, N, lo, ce, al, re)
    # an unterminated last line and a ref file with a header of another length: the two files' windows never line up
    with open(a_path, "rb") as f:
        body = f.read()
    with open(a_path, "wb") as f:
        f.write(body.rstrip(b"\n"))
    ref_lines = open(r_path).read().split("\n", 3)
    open(r_path, "w").write(ref_lines[0] + "\n% a longer comment line than the alt file has, to shift every window\n" + ref_lines[2] + "\n" + ref_lines[3])
    whole = mods["Cellector"](0)
    whole.load_mtx(a_path, r_path, 2, 2)
    for window in (512, 640, 4096, 1 << 16, -512):
        g = mods["Cellector"](0)
        g.set_option("parse_window", abs(window))
        if window < 0:  # ... and the way files of 4 GB and more are read: never mapped, pread() into the upload buffers
            os.environ["CELLECTOR_UNMAPPED_MIN"] = "1"
        try:
            g.load_mtx(a_path, r_path, 2, 2)
        finally:
            os.environ.pop("CELLECTOR_UNMAPPED_MIN", None)
        dg, dw = g.dims(), whole.dims()
        assert (dg.total_loci, dg.total_cells, dg.loci_used, dg.nnz_used) == (dw.total_loci, dw.total_cells, dw.loci_used, dw.nnz_used)
        assert np.array_equal(g.locus_ids(), whole.locus_ids())
        assert np.array_equal(g.locus_counts(), whole.locus_counts())
        ra, ea = g.csr_rows(0, N)
        rb, eb = whole.csr_rows(0, N)
        assert np.array_equal(ra, rb) and np.array_equal(ea, eb), window
        g.close()
    whole.close()
    # the text contract's corner cases through the windowed path (the size line promises fewer entries than there are:
    # the token arrays grow)
    alt, ref = tmp_path / "alt2.mtx", tmp_path / "ref2.mtx"
    alt.write_text("%%MatrixMarket\n%\n99 99 1\n1 1 2\r\n1 2 +1\n2 1 0\n  2   2 3  \n2\t3\t1\n2 3 7")
    ref.write_text("%%MatrixMarket\n%\n2 3 0\nx y 1\n8 8 0\n9 9 4\n1 1 1\n7 7 2\n")
    g = mods["Cellector"](0)
    g.set_option("parse_window", 512)
    g.load_mtx(str(alt), str(ref), 1, 1)
    d = g.dims()
    assert (d.total_loci, d.total_cells, d.loci_used, d.nnz_used) == (2, 3, 2, 5)
    g.close()
    many = "".join(f"1 {1 + i % 3} 1\n" for i in range(400))  # several windows of 512 bytes
    for name, body, where in (("float", many + "1 1 1.0\n" + many, 400), ("two_tokens", many + many + "1 1\n", 800)):
        bad = tmp_path / f"w_{name}.mtx"
        bad.write_text("%%MatrixMarket\n%\n2 3 0\n" + body)
        g = mods["Cellector"](0)
        g.set_option("parse_window", 512)
        with pytest.raises(mods["ffi"].CellectorError) as ei:
            g.load_mtx(str(bad), str(bad), 1, 1)
        assert ei.value.status == 3 and f"entry {where} " in str(ei.value), (name, str(ei.value))
        g.close()


def test_sharded_mtx_ingest_equals_coo_ingest(mods, tmp_path):
    L, N = 400, 300
    lo, ce, al, re = mods["synth"].generate_coo(L, N, 0.2, seed=12)
    a_path, r_path = mods["synth"].write_mtx_pair(str(tmp_path), L, N, lo, ce, al, re)
    g1, g2 = mods["Cellector"](0), mods["Cellector"](0)
    for g in (g1, g2):
        g.set_shard(100, 250)
    g1.ingest_mtx(a_path, r_path)
    g2.ingest_coo(L, N, lo, ce, al, re)
    p1, p2 = g1.exchange_buffer(mods["ffi"].XCHG_PASS1), g2.exchange_buffer(mods["ffi"].XCHG_PASS1)
    hip = _hip()
    b1, b2 = np.empty(p1[1]), np.empty(p2[1])
    assert hip.hipMemcpy(b1.ctypes.data, p1[0], p1[1] * 8, 2) == 0 and hip.hipMemcpy(b2.ctypes.data, p2[0], p2[1] * 8, 2) == 0
    assert np.array_equal(b1, b2) and b1.sum() > 0
    for g in (g1, g2):
        g.ingest_finish(0, 0)
    assert g1.dims().nnz_used == g2.dims().nnz_used
    assert np.array_equal(g1.csr_rows(0, 150)[1], g2.csr_rows(0, 150)[1])
    g1.close(); g2.close()
