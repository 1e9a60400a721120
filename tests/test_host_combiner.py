"""`host/combiner`: the reference's synthetic-mixture maker (combiner/src/main.rs:23-116,197-255) —
two vartrix datasets -> one majority / minority experiment with ground truth.

CPU: the file contract on two synthetic datasets with partly shared loci — locus renumbering by (chrom, pos), seeded cell
sample, barcodes.tsv / gt.tsv, header, tab-separated lines sorted by (locus, cell), every entry of the selected cells
recomputed independently with numpy; the mask form; per-read downsampling (deterministic per seed, right rate).
GPU: the mixture through the drop-in binary with `-g gt.tsv`: the planted minority population is what it labels "0".
The random stream is this tool's own (splitmix64), not rand 0.7's StdRng: same format and statistics, other picks."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "combiner")


@pytest.fixture(scope="module")
def combiner_bin():
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(os.path.join(ROOT, "host", "combiner.cpp")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s", "combiner"])
    return BIN


def _dataset(tmp, name, L, N, density, seed, positions, suffix="-1"):
    """one vartrix-style dataset: alt/ref mtx (sorted by locus, cell), barcodes, VCF with the given (chrom, pos) per locus"""
    from cellector_amd import synth
    d = os.path.join(tmp, name)
    lo, ce, al, re = synth.generate_coo(L, N, density, seed=seed, minority_fraction=0.0)
    alt, ref = synth.write_mtx_pair(d, L, N, lo, ce, al, re, header_nnz=0)
    bc = os.path.join(d, "barcodes.tsv")
    with open(bc, "w") as f:
        for i in range(N):
            f.write(f"BC{i:05d}{suffix}\n")
    vcf = os.path.join(d, "variants.vcf")
    with open(vcf, "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for chrom, pos in positions:
            f.write(f"{chrom}\t{pos}\t.\tA\tG\t50\tPASS\t.\n")
    return dict(alt=alt, ref=ref, bc=bc, vcf=vcf, coo=(lo, ce, al, re), L=L, N=N)


def _two_datasets(tmp):
    L1, L2 = 300, 260
    pos1 = [(f"chr{1 + l % 5}", 100 + 7 * l) for l in range(L1)]
    # dataset 2: its first 200 loci are dataset 1's loci 50..249 in another order, the other 60 are new positions
    shared = list(range(50, 250))
    rng = np.random.default_rng(3)
    rng.shuffle(shared)
    pos2 = [pos1[l] for l in shared] + [("chrX", 5000 + 3 * k) for k in range(60)]
    d1 = _dataset(tmp, "d1", L1, 120, 0.15, 11, pos1)
    d2 = _dataset(tmp, "d2", L2, 90, 0.15, 12, pos2)
    locus2to1 = {j: shared[j] for j in range(200)}
    locus2to1.update({200 + k: L1 + k for k in range(60)})  # appended after dataset 1's last locus, in vcf2 order
    return d1, d2, locus2to1, L1 + 60


def _run(combiner_bin, d1, d2, out, *extra):
    cmd = [combiner_bin, "--vcf1", d1["vcf"], "--vcf2", d2["vcf"], "--alt1", d1["alt"], "--ref1", d1["ref"], "--alt2", d2["alt"],
           "--ref2", d2["ref"], "--barcodes1", d1["bc"], "--barcodes2", d2["bc"], "-o", out] + list(extra)
    return subprocess.run(cmd, capture_output=True, text=True)


def _read_mtx(path):
    lines = open(path).read().split("\n")
    assert lines[0] == "%%MatrixMarket matrix coordinate real general" and lines[1] == "% written by sprs"
    body = [ln for ln in lines[3:] if ln]
    assert all(ln.count("\t") == 2 for ln in body)  # tab-separated (main.rs:113-114)
    return lines[2], np.array([[int(x) for x in ln.split("\t")] for ln in body], dtype=np.int64).reshape(-1, 3)


def _expected_entries(d, out_cell_of, locus_map):
    lo, ce, al, re = d["coo"]
    keep = np.isin(ce, list(out_cell_of))
    rows = []
    for l, c, a, r in zip(lo[keep], ce[keep], al[keep], re[keep]):
        rows.append(((locus_map[int(l)] if locus_map else int(l)) + 1, out_cell_of[int(c)], int(r), int(a)))
    return rows


def test_mixture_files(combiner_bin, tmp_path):
    d1, d2, locus2to1, L_out = _two_datasets(str(tmp_path))
    out = str(tmp_path / "mix")
    r = _run(combiner_bin, d1, d2, out, "--num_cells_1", "80", "--num_cells_2", "25", "--seed", "7")
    assert r.returncode == 0, r.stderr
    assert r.stdout == "80,25\n"
    bcs = open(os.path.join(out, "barcodes.tsv")).read().splitlines()
    gts = [ln.split("\t") for ln in open(os.path.join(out, "gt.tsv")).read().splitlines()]
    assert len(bcs) == 105 and [g[0] for g in gts] == bcs
    assert [g[1] for g in gts] == ["majority"] * 80 + ["minority"] * 25
    # dataset-1 barcodes as they are; dataset-2 barcodes with the last character replaced by '2'
    c1 = [int(b[2:7]) for b in bcs[:80]]
    c2 = [int(b[2:7]) for b in bcs[80:]]
    assert all(b.endswith("-1") for b in bcs[:80]) and all(b.endswith("-2") for b in bcs[80:])
    assert len(set(c1)) == 80 and max(c1) < 120 and len(set(c2)) == 25 and max(c2) < 90
    assert c1 != sorted(c1)  # a sample in selection order, not the first 80 cells
    size_a, alt = _read_mtx(os.path.join(out, "alt.mtx"))
    size_r, ref = _read_mtx(os.path.join(out, "ref.mtx"))
    assert size_a == size_r == f"{L_out}\t105\t0"  # the entry count is written as 0 (main.rs:66-70)
    assert np.array_equal(alt[:, :2], ref[:, :2])
    want = _expected_entries(d1, {c: i + 1 for i, c in enumerate(c1)}, None) + \
        _expected_entries(d2, {c: 81 + i for i, c in enumerate(c2)}, locus2to1)
    want.sort()  # (locus, cell, ref, alt)
    got = [(int(a[0]), int(a[1]), int(rr[2]), int(a[2])) for a, rr in zip(alt, ref)]
    assert got == want and len(got) > 3000
    assert alt[:, 0].max() > 300 and (alt[:, 0] >= 1).all()  # dataset 2's private loci landed behind dataset 1's
    # same seed -> same files; another seed -> another sample
    out2, out3 = str(tmp_path / "mix2"), str(tmp_path / "mix3")
    assert _run(combiner_bin, d1, d2, out2, "--num_cells_1", "80", "--num_cells_2", "25", "--seed", "7").returncode == 0
    assert _run(combiner_bin, d1, d2, out3, "--num_cells_1", "80", "--num_cells_2", "25", "--seed", "8").returncode == 0
    for f in ("alt.mtx", "ref.mtx", "barcodes.tsv", "gt.tsv"):
        assert open(os.path.join(out, f)).read() == open(os.path.join(out2, f)).read(), f
    assert open(os.path.join(out, "barcodes.tsv")).read() != open(os.path.join(out3, "barcodes.tsv")).read()


def test_mask_and_downsampling(combiner_bin, tmp_path):
    d1, d2, locus2to1, L_out = _two_datasets(str(tmp_path))
    mask = tmp_path / "mask.txt"
    picked = [5, 17, 18, 60, 89]
    mask.write_text("".join(f"BC{i:05d}-1\n" for i in picked) + "NOT_A_BARCODE-1\n")
    out = str(tmp_path / "masked")
    r = _run(combiner_bin, d1, d2, out, "--num_cells_1", "120", "--dataset2_mask", str(mask))
    assert r.returncode == 0 and r.stdout == "120,5\n", r.stderr
    bcs = open(os.path.join(out, "barcodes.tsv")).read().splitlines()
    assert bcs[120:] == [f"BC{i:05d}-2" for i in picked]  # file order of dataset 2
    _, full_alt = _read_mtx(os.path.join(out, "alt.mtx"))
    _, full_ref = _read_mtx(os.path.join(out, "ref.mtx"))
    # per-read downsampling: same entries (kept even at 0/0), counts never grow, about half of the reads gone
    out_ds = str(tmp_path / "ds")
    r = _run(combiner_bin, d1, d2, out_ds, "--num_cells_1", "120", "--dataset2_mask", str(mask), "--downsample_rate", "0.5")
    assert r.returncode == 0, r.stderr
    _, ds_alt = _read_mtx(os.path.join(out_ds, "alt.mtx"))
    _, ds_ref = _read_mtx(os.path.join(out_ds, "ref.mtx"))
    assert len(ds_alt) == len(full_alt)
    # (the sort key includes the counts: compare per (locus, cell) key — unique in these inputs)
    key = lambda m: {(int(x[0]), int(x[1])): int(x[2]) for x in m}
    fa, da, fr, dr = key(full_alt), key(ds_alt), key(full_ref), key(ds_ref)
    assert fa.keys() == da.keys()
    assert all(da[k] <= fa[k] for k in fa) and all(dr[k] <= fr[k] for k in fr)
    kept = (sum(da.values()) + sum(dr.values())) / (sum(fa.values()) + sum(fr.values()))
    assert 0.45 < kept < 0.55
    # errors
    r = _run(combiner_bin, d1, d2, out, "--num_cells_1", "121", "--num_cells_2", "5")
    assert r.returncode == 101 and "more cells than exist" in r.stderr
    r = _run(combiner_bin, d1, d2, out, "--num_cells_1", "10")
    assert r.returncode == 101 and "num_cells_2 or dataset2_mask" in r.stderr
    r = subprocess.run([combiner_bin, "--vcf1", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "required arguments were not provided" in r.stderr


@pytest.mark.gpu
def test_mixture_through_the_drop_in_binary(combiner_bin, hip_lib_path, tmp_path):
    """two populations with independent genotypes mixed 10:1: cellector must call the minority dataset's cells "0" """
    from cellector_amd import synth
    L, N1, N2 = 1500, 900, 700
    pos = [(f"chr{1 + l % 9}", 1000 + 11 * l) for l in range(L)]
    d1 = _dataset(str(tmp_path), "maj", L, N1, 0.12, 21, pos)
    d2 = _dataset(str(tmp_path), "min", L, N2, 0.12, 22, pos)  # same loci, another seed: other genotypes at half of them
    out = str(tmp_path / "mix")
    r = _run(combiner_bin, d1, d2, out, "--num_cells_1", "800", "--num_cells_2", "80")
    assert r.returncode == 0 and r.stdout == "800,80\n", r.stderr
    host = os.path.join(ROOT, "host", "cellector")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s", "cellector"])
    res = str(tmp_path / "res")
    r = subprocess.run([host, "-a", os.path.join(out, "alt.mtx"), "-r", os.path.join(out, "ref.mtx"), "-b", os.path.join(out, "barcodes.tsv"),
                        "-g", os.path.join(out, "gt.tsv"), "--output_directory", res, "--min_alt", "4", "--min_ref", "4"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rows = [ln.split("\t") for ln in open(os.path.join(res, "cellector_assignments.tsv")).read().splitlines()[1:]]
    assert len(rows) == 880
    minority = [row for row in rows if row[8] == "minority"]
    majority = [row for row in rows if row[8] == "majority"]
    assert len(minority) == 80 and len(majority) == 800
    assert sum(row[2] == "0" for row in minority) >= 76 and sum(row[2] == "0" for row in majority) <= 8
    assert sum(row[1] == "0" for row in minority) >= 72 and sum(row[1] == "1" for row in majority) >= 760
    assert "minority" in r.stdout and "cell_hashing" in r.stdout  # the confusion table the reference prints for this flow
