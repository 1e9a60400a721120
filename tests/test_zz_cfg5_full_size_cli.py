"""GPU: BASELINE configs[4] at FULL size through the drop-in binary — 1M cells x 200k loci at 1 % as a vartrix-style
text pair (61 GB), barcodes, ground truth and a VCF, run as `host/cellector ... --vcf ... --min_alleles_posterior 5`
(the argument list cellector_pipeline.py:223-226 passes, plus the flag configs[4] names), every output file checked:

  * a strided sample of ~25k cells spread over the whole cell range (first block, 23 interior offsets that straddle
    1024-cell blocks, the ragged last block): iteration_*.tsv (log_likelihood, expected_log_likelihood, num_loci_used)
    and cellector_assignments.tsv (label, qual, majority/minority log-likelihood) against the CPU oracle evaluating the
    reference formulas under the per-locus alpha/beta that the run's own files imply;
  * ALL cells: threshold = exact order statistics (numpy sort) of ll / num_loci_used, anomaly flags = strict `<`, the
    "detected ... rescued" stdout lines, num_loci_used against the matrix;
  * ALL loci: every integer column of iteration_*_locus_contribution.tsv and the AO/RO fields of cellector.vcf against
    numpy bincounts over the whole matrix; GT/GP of every VCF record against the oracle's genotype rule; conservation
    between the per-cell and the per-locus log-likelihood sums; row order, derived columns;
  * the confusion table on stdout byte for byte (main.rs:177-226);
  * quirk Q1: `--min_alleles_posterior` changes nothing (second run with another value: identical assignments file).

The matrix the checks use comes from the device generator (cellector_ingest_synthetic, the same seed); the binary sees
only the TEXT written from it, so the text ingest at full size is covered too.  The oracle cannot run 2e9 entries in
seconds: it runs the sample.  /dev/shm must hold the 61 GB pair: the test FAILS (not skips) if it cannot.
"""
import filecmp
import os
import shutil
import subprocess
import time

import numpy as np
import pytest

from test_host_cli import host_bin, pretty_print, rust_display  # noqa: F401  (fixture + the reference's table / `{}` models)

pytestmark = pytest.mark.gpu

N, L_TOTAL, DENSITY, SEED, MINORITY = 1_000_000, 200_000, 0.01, 4, 0.05
LL_ATOL, POST_ATOL = 1e-7, 1e-6
NEED_BYTES = 70e9  # 61.3 GB of text + barcodes, VCF, two sets of output files


def _lse(a, b):
    m = np.maximum(a, b)
    return m + np.log(np.exp(a - m) + np.exp(b - m))


def _read_tsv(path, usecols=None):
    import pandas as pd
    return pd.read_csv(path, sep="\t", usecols=usecols, float_precision="round_trip", keep_default_na=False, na_values=[])


def _r8_threshold(norm, mult=5.0):
    """statrs Data::median / lower_quartile / upper_quartile on exact order statistics, main.rs:324-329."""
    srt = np.sort(norm)
    n = len(srt)
    k = n // 2
    med = srt[k] if n % 2 else (srt[k - 1] + srt[k]) / 2.0
    h1, h3 = (n + 1.0 / 3.0) * 0.25 + 1.0 / 3.0, (n + 1.0 / 3.0) * 0.75 + 1.0 / 3.0
    q1 = srt[int(h1) - 1] + (h1 - int(h1)) * (srt[int(h1)] - srt[int(h1) - 1])
    q3 = srt[int(h3) - 1] + (h3 - int(h3)) * (srt[int(h3)] - srt[int(h3) - 1])
    return med, q3 - q1, q1 - mult * (q3 - q1)


def test_cfg5_full_size_cli_run(host_bin, oracle_lib, hip_lib_path):
    from cellector_amd import Cellector, synth
    ob = oracle_lib
    free = shutil.disk_usage("/dev/shm").free
    if free < NEED_BYTES:
        pytest.fail(f"/dev/shm has {free / 1e9:.1f} GB free; the 1M x 200k text pair needs {NEED_BYTES / 1e9:.0f} GB "
                    "(BASELINE configs[4] cannot be exercised on this box)")
    d = f"/dev/shm/cellector_cfg5_{os.getpid()}"
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    t0 = time.time()
    try:
        _run(d, host_bin, ob, Cellector, synth)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    print(f"cfg5 test: {time.time() - t0:.0f} s")


def _run(d, host_bin, ob, Cellector, synth):
    alt, ref = os.path.join(d, "alt.mtx"), os.path.join(d, "ref.mtx")
    # ---- inputs: the text pair formatted on the device from the generator's matrix
    with Cellector(0) as g:
        g.set_option("keep_coo", 1)
        g.ingest_synthetic(L_TOTAL, N, DENSITY, seed=SEED, minority_fraction=MINORITY)
        g.write_staged_mtx(alt, ref)
    assert os.path.getsize(alt) > 25e9 and os.path.getsize(ref) > 25e9
    bc, gt, vcf = os.path.join(d, "barcodes.tsv"), os.path.join(d, "gt.tsv"), os.path.join(d, "variants.vcf")
    cls = synth.cell_classes(N, seed=SEED, minority_fraction=MINORITY)
    with open(bc, "w") as f:
        f.write("".join(f"CELL{i:07d}-1\n" for i in range(N)))
    names = np.array(["majority", "minority", "doublet"])
    with open(gt, "w") as f:
        f.write("".join(f"CELL{i:07d}-1\t{n}\n" for i, n in enumerate(names[cls])))
    with open(vcf, "w") as f:
        f.write("##fileformat=VCFv4.2\n##source=synthetic\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        f.write("".join(f"chr{1 + l % 22}\t{1000 + 37 * l}\t.\tA\tG\t50\tPASS\t.\n" for l in range(L_TOTAL)))

    # ---- the run (cellector_pipeline.py:223-226's argument list + the flag configs[4] names + ground truth)
    out = os.path.join(d, "out")
    cmd = [host_bin, "-a", alt, "-r", ref, "--output_directory", out, "--min_alt", "4", "--min_ref", "4",
           "--barcodes", bc, "--vcf", vcf, "--min_alleles_posterior", "5", "-g", gt]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    print(f"host/cellector on 1M x 200k text: {wall:.1f} s wall")
    # quirk Q1: the flag is parsed and changes nothing (main.rs:301-303)
    out2 = os.path.join(d, "out2")
    cmd2 = [c if c != "5" else "500" for c in cmd]
    cmd2[cmd2.index(out)] = out2
    r2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    for name in ("cellector_assignments.tsv", "cellector.vcf", "iteration_0.tsv"):
        assert filecmp.cmp(os.path.join(out, name), os.path.join(out2, name), shallow=False), name
    assert r.stdout == r2.stdout
    shutil.rmtree(out2)
    # ---- the same run sharded over four logical shards of the GPU, the text ingest in its split form (every shard tokenises
    # a byte range of both files and the entries are routed to their owners: what a multi-GPU node does from three devices
    # on): the sharded ingest, scoring loop, posteriors and tallies at BASELINE size.  Integer results must be identical;
    # per-cell sums may differ in the last bits (a shard picks its own number of chunk groups).
    out3 = os.path.join(d, "out3")
    cmd3 = list(cmd) + ["--devices", "0,0,0,0"]
    cmd3[cmd3.index(out)] = out3
    t0 = time.time()
    r3 = subprocess.run(cmd3, capture_output=True, text=True, timeout=600, env=dict(os.environ, CELLECTOR_MULTI_SPLIT="1"))
    assert r3.returncode == 0, r3.stderr[-2000:]
    print(f"host/cellector --devices 0,0,0,0 (split ingest) on 1M x 200k text: {time.time() - t0:.1f} s wall")
    assert sorted(os.listdir(out3)) == sorted(os.listdir(out))
    assert filecmp.cmp(os.path.join(out, "cellector.vcf"), os.path.join(out3, "cellector.vcf"), shallow=False)
    det1 = [ln for ln in r.stdout.splitlines() if ln.startswith("detected ")]
    det3 = [ln for ln in r3.stdout.splitlines() if ln.startswith("detected ")]
    assert det1 == det3 and len(det1) >= 2
    for name in sorted(os.listdir(out)):
        if name.endswith("_threshold.tsv"):
            a, b = float(open(os.path.join(out, name)).read()), float(open(os.path.join(out3, name)).read())
            assert abs(a - b) <= 1e-12 * abs(a), name
        elif name.startswith("iteration_") and "locus" not in name:
            x = _read_tsv(os.path.join(out, name))
            y = _read_tsv(os.path.join(out3, name))
            assert (x["barcode"] == y["barcode"]).all() and (x["assignment"] == y["assignment"]).all(), name
            assert np.array_equal(x["num_loci_used"].to_numpy(), y["num_loci_used"].to_numpy()), name
            for col in ("log_likelihood", "expected_log_likelihood"):
                u, v = x[col].to_numpy(np.float64), y[col].to_numpy(np.float64)
                assert np.max(np.abs(u - v) / np.maximum(1.0, np.abs(u))) < 1e-12, (name, col)
    x = _read_tsv(os.path.join(out, "cellector_assignments.tsv"))
    y = _read_tsv(os.path.join(out3, "cellector_assignments.tsv"))
    for col in ("barcode", "posterior_assignment", "anomally_assignment", "ground_truth_assignment"):
        assert (x[col] == y[col]).all(), col
    assert np.array_equal(x["loci_used"].to_numpy(), y["loci_used"].to_numpy())
    assert np.max(np.abs(x["posterior_assign_qual"].to_numpy(np.int64) - y["posterior_assign_qual"].to_numpy(np.int64))) <= 1
    for col in ("log_likelihood_loci_normalized", "majority_log_likelihood", "minority_log_likelihood"):
        u, v = x[col].to_numpy(np.float64), y[col].to_numpy(np.float64)
        assert np.max(np.abs(u - v) / np.maximum(1.0, np.abs(u))) < 1e-12, col
    del x, y
    shutil.rmtree(out3)
    os.remove(alt); os.remove(ref)  # (61 GB of host memory back before the checks allocate theirs)

    # ---- the matrix for the checks: generator -> CSR on the device (same seed; never saw the text)
    g = Cellector(0)
    g.set_option("keep_coo", 0)
    g.load_synthetic(L_TOTAL, N, DENSITY, seed=SEED, minority_fraction=MINORITY)
    dm = g.dims()
    L = dm.loci_used
    locus_ids = g.locus_ids()

    # ---- read the run's files
    stdout = r.stdout
    det = [ln for ln in stdout.splitlines() if ln.startswith("detected ")]
    med_lines = [ln for ln in stdout.splitlines() if ln.startswith("median normalized")]
    n_iter = len(det)
    assert 2 <= n_iter <= 30 and len(med_lines) == n_iter
    assert not os.path.exists(os.path.join(out, f"iteration_{n_iter}.tsv"))
    it_cells, it_loci, thr_file = [], [], []
    for i in range(n_iter):
        df = _read_tsv(os.path.join(out, f"iteration_{i}.tsv"))
        assert list(df.columns) == ["cell_id", "barcode", "assignment", "log_likelihood", "expected_log_likelihood", "num_loci_used"]
        assert len(df) == N and np.array_equal(df["cell_id"].to_numpy(), np.arange(N))
        assert df["barcode"].iloc[123456] == "CELL0123456-1" and df["assignment"].iloc[7] == names[cls[7]]
        it_cells.append({k: df[k].to_numpy(np.float64) for k in ("log_likelihood", "expected_log_likelihood", "num_loci_used")})
        lf = _read_tsv(os.path.join(out, f"iteration_{i}_locus_contribution.tsv"))
        assert len(lf.columns) == 17 and lf.columns[0] == "locus_id" and lf.columns[-1] == "majority_af" and len(lf) == L
        it_loci.append(lf)
        t = open(os.path.join(out, f"iteration_{i}_threshold.tsv")).read()
        assert not t.endswith("\n") and rust_display(float(t)) == t
        thr_file.append(float(t))
    # (text form of the floats: every 97th row of the first iteration's big file; test_host_cli.py checks every row)
    with open(os.path.join(out, "iteration_0.tsv")) as f:
        for k, ln in enumerate(f):
            if k and k % 97 == 0:
                for x in ln.rstrip("\n").split("\t")[3:6]:
                    assert rust_display(float(x)) == x, x

    # ---- per-iteration state implied by the files: exclusion sets (all cells), locus masks
    pos_of = np.full(L_TOTAL, -1, np.int64)
    pos_of[locus_ids.astype(np.int64)] = np.arange(L)
    flags, masks = [], [np.ones(L, np.uint8)]
    prev = np.zeros(N, bool)
    for i in range(n_iter):
        c = it_cells[i]
        nl = c["num_loci_used"]
        with np.errstate(invalid="ignore", divide="ignore"):
            norm = np.where(nl > 0, c["log_likelihood"] / nl, 0.0)  # main.rs:315-322
        c["norm"] = norm
        med, iqr, thr = _r8_threshold(norm)
        assert thr_file[i] == thr, (i, thr_file[i], thr)  # exact order statistics of ALL cells
        tok = med_lines[i].split()
        assert (float(tok[4]), float(tok[8].rstrip(",")), float(tok[-1])) == (med, iqr, thr)
        ex = norm < thr
        assert det[i] == (f"detected {int((ex & ~prev).sum())} new anomylous cells and rescued {int((~ex & prev).sum())} "
                          f"cells to the majority in iteration {i + 1}")
        assert (i + 1 == n_iter) == (not (ex ^ prev).any())  # the loop stops exactly when nothing changes (main.rs:42-46)
        near = np.abs(norm - thr) <= 1e-9 * max(1.0, abs(thr))
        assert ("warning: iteration %d:" % (i + 1) in r.stderr) == bool(near.any())
        flags.append(ex)
        prev = ex
        lf = it_loci[i]
        pc = lf["log_likelihood_minority_per_cell"].to_numpy(np.float64)
        li = pos_of[lf["locus_id"].to_numpy(np.int64)]
        assert (li >= 0).all() and len(np.unique(li)) == L
        assert np.all(np.diff(pc) >= 0) and np.all(np.diff(li)[np.diff(pc) == 0] > 0)  # stable argsort (main.rs:440,453)
        m = masks[-1].copy()
        m[li[pc < -80.0]] = 0  # main.rs:444-447
        masks.append(m)
        n_filtered = int((masks[-2] != m).sum())
        assert stdout.count("filtering locus") >= n_filtered
    excl_final = flags[-1]
    assert 0.03 * N < excl_final.sum() < 0.07 * N
    assert (excl_final[cls == 1]).mean() > 0.95 and (excl_final[cls == 0]).mean() < 0.01  # the planted minority is found

    # ---- one chunked pass over the whole matrix: per-locus tallies by bincount, per-cell used-locus counts
    tot = np.zeros((3, L))                          # entries, alt, ref per locus (all cells)
    mino = np.zeros((n_iter, 3, L))                 # the same over each iteration's NEW exclusion set
    nl_true = [np.zeros(N) for _ in range(n_iter)]
    CH = 50_000
    for b in range(0, N, CH):
        e = min(N, b + CH)
        rp, ent = g.csr_rows(b, e)
        lens = np.diff(rp.astype(np.int64))
        loc = (ent & np.uint64(0xFFFFFFFF)).astype(np.int64)
        a = ((ent >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.float64)
        rf = (ent >> np.uint64(48)).astype(np.float64)
        tot[0] += np.bincount(loc, minlength=L)
        tot[1] += np.bincount(loc, weights=a, minlength=L)
        tot[2] += np.bincount(loc, weights=rf, minlength=L)
        for i in range(n_iter):
            sel = np.repeat(flags[i][b:e], lens)
            ls = loc[sel]
            mino[i, 0] += np.bincount(ls, minlength=L)
            mino[i, 1] += np.bincount(ls, weights=a[sel], minlength=L)
            mino[i, 2] += np.bincount(ls, weights=rf[sel], minlength=L)
            if masks[i].all():
                nl_true[i][b:e] = lens
            else:
                cell_of = np.repeat(np.arange(e - b), lens)
                nl_true[i][b:e] = np.bincount(cell_of, weights=masks[i][loc].astype(np.float64), minlength=e - b)
    assert int(tot[0].sum()) == dm.nnz_used
    for i in range(n_iter):
        assert np.array_equal(it_cells[i]["num_loci_used"], nl_true[i]), i
        lf = it_loci[i]
        li = pos_of[lf["locus_id"].to_numpy(np.int64)]
        live = masks[i][li].astype(np.float64)  # a masked locus has no entries in the pass (main.rs:556): all zeros
        want = {"minority_cellcount": mino[i, 0], "majority_cellcount": tot[0] - mino[i, 0],
                "minority_alt": mino[i, 1], "minority_ref": mino[i, 2],
                "majority_alt": tot[1] - mino[i, 1], "majority_ref": tot[2] - mino[i, 2]}
        for k, v in want.items():
            assert np.array_equal(lf[k].to_numpy(np.float64), v[li] * live), (i, k)
        cmin, cmaj = lf["log_likelihood_minority"].to_numpy(), lf["log_likelihood_majority"].to_numpy()
        assert np.array_equal(lf["expected_loglike_minority"].to_numpy(), cmin)  # quirk Q6
        assert np.array_equal(lf["expected_loglike_majority"].to_numpy(), cmaj)
        nmin, nmaj = lf["minority_cellcount"].to_numpy(np.float64), lf["majority_cellcount"].to_numpy(np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            assert np.array_equal(lf["log_likelihood_minority_per_cell"].to_numpy(), np.where(nmin > 0, cmin / nmin, 0.0))
            assert np.array_equal(lf["log_likelihood_majority_per_cell"].to_numpy(), np.where(nmaj > 0, cmaj / nmaj, 0.0))
            am, rm = lf["minority_alt"].to_numpy(np.float64), lf["minority_ref"].to_numpy(np.float64)
            assert np.array_equal(lf["minority_af"].to_numpy(), np.where(am + rm > 0, am / (am + rm), 0.0))
        assert (lf["chrom"] == [f"chr{1 + int(x) % 22}" for x in lf["locus_id"]]).all()
        assert (lf["pos"].to_numpy(np.int64) == 1000 + 37 * lf["locus_id"].to_numpy(np.int64)).all()
        # conservation: the cell pass and the locus pass saw the same log-pmfs
        s_cells = it_cells[i]["log_likelihood"].sum()
        assert abs(s_cells - (cmin.sum() + cmaj.sum())) < 1e-9 * abs(s_cells), i

    # ---- the oracle on a strided sample of cells, under the alpha/beta the files imply
    starts = [0] + [k * (N // 24) + 517 * k for k in range(1, 24)] + [N - 1024]
    ranges = [(s, min(N, s + 1024)) for s in starts]
    sel = np.concatenate([np.arange(a, b) for a, b in ranges])
    assert len(np.unique(sel)) == len(sel) >= 20_000 and sel[-1] == N - 1
    rps, ents = zip(*(g.csr_rows(a, b) for a, b in ranges))
    rp = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(x.astype(np.int64)) for x in rps]))]).astype(np.uint64)
    lc = np.stack([tot[2], tot[1]], axis=1)  # locus_counts[l] = [sum ref, sum alt] (load_data.rs:157-158)
    assert np.array_equal(lc, g.locus_counts())
    epc_sample = np.diff(rp.astype(np.int64))
    g.close()
    o = ob.Oracle.from_csr(L, rp, np.concatenate(ents), lc)
    ob.set_threads(ob.host_threads())
    try:
        for i in range(n_iter):
            # init_alpha_betas(previous exclusion set), main.rs:598-611 — over ALL entries, masked loci included
            sub = mino[i - 1] if i else np.zeros((3, L))
            alpha, beta = (tot[1] + 1.0) - sub[1], (tot[2] + 1.0) - sub[2]
            ll_o, ell_o, nl_o = o.cell_log_likelihoods(alpha, beta, masks[i])
            np.testing.assert_allclose(it_cells[i]["log_likelihood"][sel], ll_o, rtol=0, atol=LL_ATOL)
            np.testing.assert_allclose(it_cells[i]["expected_log_likelihood"][sel], ell_o, rtol=0, atol=LL_ATOL)
            assert np.array_equal(it_cells[i]["num_loci_used"][sel], nl_o)
        # calculate_posteriors (main.rs:228-280): all loci, final exclusion set
        alt_min, ref_min = mino[-1, 1], mino[-1, 2]
        n_exc = int(excl_final.sum())
        a_maj, b_maj = (tot[1] + 1.0) - alt_min, (tot[2] + 1.0) - ref_min
        a_min, b_min = (tot[1] + 1.0) - (tot[1] - alt_min), (tot[2] + 1.0) - (tot[2] - ref_min)
        mf0 = (n_exc + 1.0) / (N + 1.0)
        a_dbl, b_dbl = (a_maj - 1.0) * mf0 + (a_min - 1.0) + 1.0, (b_maj - 1.0) * mf0 + (b_min - 1.0) + 1.0
        mf = max(mf0, 0.01)
        a_maj, b_maj = (a_maj - 1.0) * mf + 1.0, (b_maj - 1.0) * mf + 1.0
        lp_min, lp_maj, lp_dbl = np.log(mf), np.log(1.0 - mf), np.log(N / 1000.0 / 100.0 * max(mf, 0.1))
        l_min = o.cell_log_likelihoods(a_min, b_min)[0]
        l_maj = o.cell_log_likelihoods(a_maj, b_maj)[0]
        l_dbl = o.cell_log_likelihoods(a_dbl, b_dbl)[0]
    finally:
        ob.set_threads(1)
        o.close()
    num = lp_min + l_min
    den = _lse(_lse(num, lp_maj + l_maj), lp_dbl + l_dbl)
    post_o, dbl_o = np.exp(num - den), np.exp(lp_dbl + l_dbl - den)

    # ---- cellector_assignments.tsv: all cells for what the files determine, the sample against the oracle
    af = _read_tsv(os.path.join(out, "cellector_assignments.tsv"))
    assert list(af.columns) == ["barcode", "posterior_assignment", "anomally_assignment", "log_likelihood_loci_normalized",
                                "loci_used", "posterior_assign_qual", "majority_log_likelihood", "minority_log_likelihood",
                                "ground_truth_assignment"]
    assert len(af) == N and af["barcode"].iloc[999_999] == "CELL0999999-1"
    assert np.array_equal(af["anomally_assignment"].to_numpy(np.int64), np.where(excl_final, 0, 1))
    assert np.array_equal(af["log_likelihood_loci_normalized"].to_numpy(np.float64), it_cells[-1]["norm"])
    assert np.array_equal(af["loci_used"].to_numpy(np.float64), it_cells[-1]["num_loci_used"])
    assert (af["ground_truth_assignment"].to_numpy() == names[cls]).all()
    pa = af["posterior_assignment"].astype(str).to_numpy()
    np.testing.assert_allclose(af["majority_log_likelihood"].to_numpy(np.float64)[sel], l_maj, rtol=0, atol=LL_ATOL)
    np.testing.assert_allclose(af["minority_log_likelihood"].to_numpy(np.float64)[sel], l_min, rtol=0, atol=LL_ATOL)
    from cellector_amd import ffi
    pa_o, _, q_o = ffi.assignments(post_o, dbl_o, epc_sample, excl_final[sel])
    # a posterior within 1e-6 of a decision boundary could legitimately land on either side: none may sit there
    edge = (np.abs(post_o - 0.999) < POST_ATOL) | (np.abs((1.0 - post_o) - 0.999) < POST_ATOL) | (np.abs(dbl_o - 0.5) < POST_ATOL)
    assert not edge.any()
    assert np.array_equal(pa[sel], np.array(["0", "1", "doublet", "unassigned"])[pa_o])  # labels bit-identical
    assert np.max(np.abs(af["posterior_assign_qual"].to_numpy(np.int64)[sel] - q_o.astype(np.int64))) <= 1
    assert {"0", "1"} <= set(pa[sel].tolist())
    # confusion table, byte for byte (main.rs:177-226)
    import pandas as pd
    ct = pd.crosstab(af["posterior_assignment"].astype(str), af["ground_truth_assignment"])
    agc = {str(k): {c: int(v) for c, v in row.items() if v} for k, row in ct.iterrows()}
    gtc = {c: int(v) for c, v in af["ground_truth_assignment"].value_counts().items()}
    assert len(set(gtc.values())) == len(gtc)
    assert stdout.endswith(pretty_print(agc, gtc) + "\n")

    # ---- cellector.vcf: AO/RO of every record = bincount tallies by the final exclusion set (all loci: the mask does
    # not apply, load_data.rs:109-132), GT/GP = the oracle's genotype rule
    assert L == L_TOTAL  # (every locus passes min_alt/min_ref = 4 at this size: the used-loci tallies are all there is)
    amin, rmin = mino[-1, 1].astype(np.int64), mino[-1, 2].astype(np.int64)
    amaj, rmaj = (tot[1] - mino[-1, 1]).astype(np.int64), (tot[2] - mino[-1, 2]).astype(np.int64)
    gts = {0: "./.", 1: "1/1", 2: "0/1", 3: "0/0"}
    seen = set()
    with open(os.path.join(out, "cellector.vcf")) as f:
        assert f.readline() == "##fileformat=VCFv4.2\n" and f.readline() == "##source=synthetic\n"
        assert f.readline().endswith("\tINFO\tmajority\tminority\n")
        n_rec = 0
        for l, ln in enumerate(f):
            t = ln.rstrip("\n").split("\t")
            assert t[0] == f"chr{1 + l % 22}" and t[1] == str(1000 + 37 * l) and t[8] == "GT:GP:AO:RO"
            fmaj, fmin = t[9].split(":"), t[10].split(":")
            assert (int(fmaj[2]), int(fmaj[3]), int(fmin[2]), int(fmin[3])) == (amaj[l], rmaj[l], amin[l], rmin[l]), l
            gmaj, pmaj, gmin, pmin = ob.vcf_genotype(amin[l], rmin[l], amaj[l], rmaj[l])
            assert fmaj[0] == gts[gmaj] and fmin[0] == gts[gmin], l
            assert abs(float(fmaj[1]) - pmaj) <= 1e-9 * pmaj and abs(float(fmin[1]) - pmin) <= 1e-9 * pmin, l
            seen.add(fmaj[0])
            n_rec += 1
    assert n_rec == L_TOTAL and {"0/0", "0/1", "1/1"} <= seen
