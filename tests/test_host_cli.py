"""The C++ host binary `host/cellector`: command line of params.yml, files of SURVEY Appendix C.

CPU part: argument handling (exits before any GPU call).  GPU part: a full run on synthetic vartrix-style files,
every output file checked against the oracle (values) and against Rust's `{}` float formatting (text)."""
import gzip
import math
import os
import subprocess
from decimal import Decimal

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "cellector")


@pytest.fixture(scope="module")
def host_bin(hip_lib_path):
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(os.path.join(ROOT, "host", "cellector.cpp")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    return BIN


def rust_display(x):
    """Rust `format!("{}", f64)`: shortest round-trip digits, positional, integral values without '.0'."""
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    s = format(Decimal(repr(float(x))), "f")
    return s[:-2] if s.endswith(".0") else s


def test_rust_display_model():
    assert rust_display(1.0) == "1" and rust_display(-0.5) == "-0.5" and rust_display(1e-7) == "0.0000001"
    assert rust_display(-0.0) == "-0" and rust_display(1e21) == "1000000000000000000000"
    assert rust_display(0.1 + 0.2) == "0.30000000000000004" and rust_display(float("inf")) == "inf"


def pretty_print(assignment_gt_counts, gt_counts):
    """The confusion table of main.rs:177-226, built string for string (what `println!("\\n\\n{}", string_build)` prints,
    without that final newline).  Ground truths are listed by descending count (ties: hash order in the reference — the
    tests use ground truths with distinct counts)."""
    count_vec = sorted(gt_counts.items(), key=lambda kv: -kv[1])
    first_header, header = "cellector assignment   ", "      0      1      unassigned\n"
    sb = first_header + header
    xoffset = max(3, len(first_header) + 2)
    sb += "cell_hashing" + " " * max(0, xoffset - 12) + "|" + "-" * max(0, len(header) - 1) + "|\n"
    for gt, _ in count_vec:
        xoffset = max(xoffset, len(gt) + 3)
        c0, c1, un = (str(assignment_gt_counts.get(k, {}).get(gt, 0)) for k in ("0", "1", "unassigned"))
        sb += gt + " " * max(0, xoffset - max(0, len(gt) - 1))
        sb += " |  " + c0 + " " * max(0, 4 - len(c0))
        sb += " |  " + c1 + " " * max(0, 4 - len(c1))
        sb += " |  " + un + " " * max(0, 12 - len(un)) + "|\n"
    sb += " " * xoffset + "|" + "-" * max(0, len(header) - 1) + "|\n"
    return "\n\n" + sb


def test_pretty_print_model():
    t = pretty_print({"0": {"minority": 7}, "1": {"majority": 12345, "minority": 1}, "unassigned": {"majority": 2}},
                     {"majority": 12347, "minority": 8})
    assert t == ("\n\ncellector assignment         0      1      unassigned\n"
                 "cell_hashing             |------------------------------|\n"
                 "majority                   |  0    |  12345 |  2           |\n"
                 "minority                   |  7    |  1    |  0           |\n"
                 "                         |------------------------------|\n")


def test_cli_usage_errors(host_bin):
    r = subprocess.run([host_bin, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--output_directory" in r.stdout and "-r, --ref <ref>" in r.stdout
    r = subprocess.run([host_bin, "-a", "x", "-r", "y"], capture_output=True, text=True)
    assert r.returncode != 0 and "required arguments were not provided" in r.stderr
    r = subprocess.run([host_bin, "--bogus", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "wasn't expected" in r.stderr
    r = subprocess.run([host_bin, "-a", "a", "-r", "r", "-b", "b", "--output_directory", "o", "--min_alt", "4.5"],
                       capture_output=True, text=True)
    assert r.returncode == 101 and "min_alt" in r.stderr  # parse::<usize>().unwrap() panics in the reference


def _write_inputs(tmp, L, N, d, seed, minority, gz=False):
    from cellector_amd import synth
    lo, ce, al, re = synth.generate_coo(L, N, d, seed=seed, minority_fraction=minority, doublet_fraction=0.01)
    alt, ref = synth.write_mtx_pair(str(tmp), L, N, lo, ce, al, re, gz=gz, header_nnz=0)
    bc = os.path.join(tmp, "barcodes.tsv.gz" if gz else "barcodes.tsv")
    synth.write_barcodes(bc, N)
    gt = os.path.join(tmp, "gt.tsv")
    synth.write_ground_truth(gt, synth.cell_classes(N, seed=seed, minority_fraction=minority, doublet_fraction=0.01))
    vcf = os.path.join(tmp, "variants.vcf")
    with open(vcf, "w") as f:
        f.write("##fileformat=VCFv4.2\n##source=synthetic\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for l in range(L):
            f.write(f"chr{1 + l % 22}\t{1000 + 37 * l}\t.\tA\tG\t50\tPASS\t.\n")
    return (lo, ce, al, re), alt, ref, bc, gt, vcf


@pytest.mark.gpu
@pytest.mark.parametrize("gz", [False, True])
def test_full_run_matches_oracle(host_bin, oracle_lib, tmp_path, gz):
    L, N = 1500, 700
    coo, alt, ref, bc, gt, vcf = _write_inputs(str(tmp_path), L, N, 0.12, seed=4, minority=0.08, gz=gz)
    out = str(tmp_path / "out")
    cmd = [host_bin, "-a", alt, "-r", ref, "--output_directory", out, "--min_alt", "4", "--min_ref", "4",
           "--barcodes", bc, "--vcf", vcf, "-g", gt]  # the argument shape cellector_pipeline.py:223-226 uses (+ -g)
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    o = oracle_lib.Oracle.from_mtx(alt, ref, 4, 4)
    summaries = []
    lines = [ln for ln in r.stdout.splitlines() if not ln.startswith("filtering locus")]
    it = 0
    while True:
        s = o.em_iteration(5.0)
        summaries.append(s)
        co, lo_ = o.cell_outputs(), o.locus_outputs()
        # stdout progress lines (main.rs:338-339)
        assert lines[2 * it] == (f"detected {s.n_new_excluded} new anomylous cells and rescued {s.n_rescued} cells to the "
                                 f"majority in iteration {it + 1}")
        tok = lines[2 * it + 1].split()
        assert tok[:4] == ["median", "normalized", "log", "likelihood"]
        assert float(tok[4]) == pytest.approx(s.median, abs=1e-9) and float(tok[-1]) == pytest.approx(s.threshold, abs=1e-9)
        # iteration_N.tsv (Appendix C.2)
        rows = [ln.split("\t") for ln in open(os.path.join(out, f"iteration_{it}.tsv")).read().splitlines()]
        assert rows[0] == ["cell_id", "barcode", "assignment", "log_likelihood", "expected_log_likelihood", "num_loci_used"]
        assert len(rows) == N + 1
        got = np.array([[float(x) for x in row[3:6]] for row in rows[1:]])
        np.testing.assert_allclose(got[:, 0], co["ll"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(got[:, 1], co["expected_ll"], rtol=0, atol=1e-7)
        assert np.array_equal(got[:, 2], co["loci_used"])
        assert rows[1][0] == "0" and rows[1][1] == "CELL0000000-1" and rows[1][2] in ("majority", "minority", "doublet")
        for row in rows[1:]:  # every float of every row is in Rust's `{}` form
            for x in row[3:6]:
                assert rust_display(float(x)) == x, x
        thr = open(os.path.join(out, f"iteration_{it}_threshold.tsv")).read()
        assert not thr.endswith("\n") and float(thr) == pytest.approx(s.threshold, abs=1e-9) and rust_display(float(thr)) == thr
        # iteration_N_locus_contribution.tsv (Appendix C.4)
        rows = [ln.split("\t") for ln in open(os.path.join(out, f"iteration_{it}_locus_contribution.tsv")).read().splitlines()]
        assert len(rows[0]) == 17 and rows[0][0] == "locus_id" and rows[0][-1] == "majority_af" and len(rows) == o.loci_used + 1
        ids = o.locus_ids()
        pos_of = {int(v): i for i, v in enumerate(ids)}
        per_cell = np.array([float(row[9]) for row in rows[1:]])
        assert np.all(np.diff(per_cell) >= 0)  # ascending log_likelihood_minority_per_cell (argsort, main.rs:440,453)
        for row in rows[1:]:
            l = pos_of[int(row[0])]
            for x in row[3:7] + row[9:11] + row[15:17]:
                assert rust_display(float(x)) == x, x
            assert row[1] == f"chr{1 + int(row[0]) % 22}" and row[2] == str(1000 + 37 * int(row[0]))
            assert float(row[3]) == pytest.approx(lo_["contrib_min"][l], abs=1e-7) and row[5] == row[3] and row[6] == row[4]
            assert float(row[4]) == pytest.approx(lo_["contrib_maj"][l], abs=1e-6)
            assert [int(row[7]), int(row[8])] == [lo_["cells_min"][l], lo_["cells_maj"][l]]
            assert [int(x) for x in row[11:15]] == [lo_["alt_min"][l], lo_["ref_min"][l], lo_["alt_maj"][l], lo_["ref_maj"][l]]
        it += 1
        if not s.any_change:
            break
    assert not os.path.exists(os.path.join(out, f"iteration_{it}.tsv"))
    # cellector_assignments.tsv (Appendix C.5)
    po = o.posteriors()
    pa, aa, q = o.assignments(po["posterior"], po["doublet_posterior"], 0.999, 30)
    names = {0: "0", 1: "1", 2: "doublet", 3: "unassigned"}
    rows = [ln.split("\t") for ln in open(os.path.join(out, "cellector_assignments.tsv")).read().splitlines()]
    assert rows[0] == ["barcode", "posterior_assignment", "anomally_assignment", "log_likelihood_loci_normalized",
                       "loci_used", "posterior_assign_qual", "majority_log_likelihood", "minority_log_likelihood",
                       "ground_truth_assignment"]
    co = o.cell_outputs()
    for c, row in enumerate(rows[1:]):
        for x in (row[3], row[6], row[7]):
            assert rust_display(float(x)) == x, x
        assert row[0] == f"CELL{c:07d}-1" and row[1] == names[pa[c]] and row[2] == str(aa[c])
        assert float(row[3]) == pytest.approx(co["normalized"][c], abs=1e-9) and int(row[4]) == int(co["loci_used"][c])
        assert abs(int(row[5]) - int(q[c])) <= 1
        assert float(row[6]) == pytest.approx(po["ll_majority"][c], abs=1e-7)
        assert float(row[7]) == pytest.approx(po["ll_minority"][c], abs=1e-7)
    assert {"0", "1"} <= {row[1] for row in rows[1:]}
    # confusion table on stdout (main.rs:177-226)
    agc, gtc = {}, {}
    for row in rows[1:]:
        agc.setdefault(row[1], {}).setdefault(row[8], 0)
        agc[row[1]][row[8]] += 1
        gtc[row[8]] = gtc.get(row[8], 0) + 1
    assert len(set(gtc.values())) == len(gtc) == 3  # distinct counts: the table's row order is defined
    table = pretty_print(agc, gtc) + "\n"  # println!
    assert r.stdout.endswith(table), (r.stdout[-600:], table)  # byte for byte
    assert r.stdout.count("\n\n\ncellector assignment") == 1
    # cellector.vcf (Appendix C.8)
    tallies = oracle_lib.final_tallies_coo(L, *coo, o.excluded())
    vl = open(os.path.join(out, "cellector.vcf")).read().splitlines()
    assert vl[0] == "##fileformat=VCFv4.2" and vl[2].endswith("\tINFO\tmajority\tminority")
    gts = {0: "./.", 1: "1/1", 2: "0/1", 3: "0/0"}
    seen = set()
    for l, ln in enumerate(vl[3:]):
        t = ln.split("\t")
        assert t[8] == "GT:GP:AO:RO"
        gmaj, pmaj, gmin, pmin = oracle_lib.vcf_genotype(tallies["alt_min"][l], tallies["ref_min"][l],
                                                         tallies["alt_maj"][l], tallies["ref_maj"][l])
        fmaj, fmin = t[9].split(":"), t[10].split(":")
        assert fmaj[0] == gts[gmaj] and fmin[0] == gts[gmin]
        assert float(fmaj[1]) == pytest.approx(pmaj, rel=1e-9) and float(fmin[1]) == pytest.approx(pmin, rel=1e-9)
        assert [int(fmaj[2]), int(fmaj[3])] == [tallies["alt_maj"][l], tallies["ref_maj"][l]]
        assert [int(fmin[2]), int(fmin[3])] == [tallies["alt_min"][l], tallies["ref_min"][l]]
        seen.add(fmaj[0])
    assert len(vl) == L + 3 and {"0/0", "0/1", "1/1"} <= seen


@pytest.mark.gpu
def test_devices_flag_shards_the_run(host_bin, tmp_path):
    """`--devices 0,0,0`: the cells sharded over three logical shards of GPU 0, exchanges inside the library — the files
    must be those of the single-device run: byte-identical where only per-cell results and integer tallies go in
    (iteration files, thresholds, assignments, VCF, stdout), equal to 1e-9 for the per-locus f64 contribution sums (their
    summation order follows the shard count).  (Byte identity of the per-cell files holds where the shards pick the same number
    of chunk groups as the single device — small matrices like this one; at 200k x 100k over four shards the log-likelihood
    columns differ by 8e-16 relative, labels and VCF stay identical: tools/devices_check.sh.)"""
    _, alt, ref, bc, gt, vcf = _write_inputs(str(tmp_path), 1500, 700, 0.12, seed=4, minority=0.08)
    outs, stdouts = [], []
    # (CELLECTOR_BANK_ORDER=0: the tile rows keep their file order, the layout whose per-cell sums do not depend on which cells share
    #  a shard; the default order is chosen per 32 rows against LDS bank conflicts and makes them differ in the last bits)
    env = dict(os.environ, CELLECTOR_BANK_ORDER="0")
    for name, extra in (("one", ["--device", "0"]), ("three", ["--devices", "0,0,0"])):
        out = str(tmp_path / name)
        r = subprocess.run([host_bin, "-a", alt, "-r", ref, "--output_directory", out, "--min_alt", "4", "--min_ref", "4",
                            "--barcodes", bc, "--vcf", vcf, "-g", gt] + extra, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        outs.append(out)
        stdouts.append(r.stdout)
    assert stdouts[0] == stdouts[1]
    files = sorted(os.listdir(outs[0]))
    assert files == sorted(os.listdir(outs[1])) and "cellector.vcf" in files and "iteration_1.tsv" in files
    for f in files:
        a, b = open(os.path.join(outs[0], f)).read(), open(os.path.join(outs[1], f)).read()
        if not f.endswith("_locus_contribution.tsv"):
            assert a == b, f
            continue
        ra = {ln.split("\t")[0]: ln.split("\t") for ln in a.splitlines()[1:]}
        rb = {ln.split("\t")[0]: ln.split("\t") for ln in b.splitlines()[1:]}
        assert ra.keys() == rb.keys()
        for k, x in ra.items():
            y = rb[k]
            assert x[1:3] == y[1:3] and x[7:9] == y[7:9] and x[11:15] == y[11:15], (f, k)
            for i in (3, 4, 5, 6, 9, 10, 15, 16):
                assert abs(float(x[i]) - float(y[i])) <= 1e-9 * max(1.0, abs(float(x[i]))), (f, k, i)
    r = subprocess.run([host_bin, "-a", alt, "-r", ref, "-b", bc, "--output_directory", outs[0], "--device", "0", "--devices", "0"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "cannot be used with" in r.stderr


@pytest.mark.gpu
def test_runtime_errors_exit_nonzero(host_bin, tmp_path):
    _, alt, ref, bc, gt, vcf = _write_inputs(str(tmp_path), 200, 100, 0.2, seed=1, minority=0.1)
    out = str(tmp_path / "o")
    r = subprocess.run([host_bin, "-a", str(tmp_path / "nope.mtx"), "-r", ref, "-b", bc, "--output_directory", out],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "couldn't open file" in r.stderr
    short = tmp_path / "short_barcodes.tsv"
    short.write_text("A\nB\n")
    r = subprocess.run([host_bin, "-a", alt, "-r", ref, "-b", str(short), "--output_directory", out],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "barcodes" in r.stderr
    r = subprocess.run([host_bin, "-a", alt, "-r", ref, "-b", bc, "--output_directory", str(tmp_path / "no" / "parent")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Unable to create file" in r.stderr  # quirk Q13
