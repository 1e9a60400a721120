"""Host-side logic that needs no GPU: synthetic-matrix twin, mtx writers, assignment rule, oracle flow on the
BASELINE cfg1 case (1k cells x 2k loci) and on hand-built quirk fixtures."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

from cellector_amd import ffi, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_synth_is_deterministic_and_shardable():
    a = synth.generate_coo(200, 300, 0.1, seed=4)
    b = synth.generate_coo(200, 300, 0.1, seed=4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    lo, ce, al, re = a
    assert np.all(np.diff(lo.astype(np.int64) * 300 + ce) > 0)  # sorted by (locus, cell), no duplicates
    assert abs(len(lo) / (200 * 300) - 0.1) < 0.01
    n = al + re
    assert n.min() >= 1 and n.max() <= 19 and abs(n.mean() - 1 / 0.7) < 0.05
    # shards generated independently concatenate (per locus) to the unsharded matrix
    s0 = synth.generate_coo(200, 300, 0.1, seed=4, cell_begin=0, cell_end=123)
    s1 = synth.generate_coo(200, 300, 0.1, seed=4, cell_begin=123, cell_end=300)
    key = np.concatenate([s0[0].astype(np.int64) * 300 + s0[1], s1[0].astype(np.int64) * 300 + s1[1]])
    order = np.argsort(key, kind="stable")
    for k in range(4):
        assert np.array_equal(np.concatenate([s0[k], s1[k]])[order], a[k])
    assert not np.array_equal(synth.generate_coo(200, 300, 0.1, seed=5)[1], ce)


def test_synth_genotypes_separate_populations():
    lo, ce, al, re = synth.generate_coo(400, 500, 0.3, seed=4, minority_fraction=0.2)
    cls = synth.cell_classes(500, seed=4, minority_fraction=0.2)
    assert 60 < (cls == 1).sum() < 140
    af = al.sum() / (al.sum() + re.sum())
    assert 0.25 < af < 0.55


def test_mtx_writer_roundtrip(tmp_path, oracle_lib):
    lo, ce, al, re = synth.generate_coo(120, 90, 0.2, seed=1)
    for gz in (False, True):
        a, r = synth.write_mtx_pair(str(tmp_path / str(gz)), 120, 90, lo, ce, al, re, gz=gz, header_nnz=0)
        opener = gzip.open if gz else open
        lines = opener(a, "rt").read().splitlines()
        assert lines[0].startswith("%%MatrixMarket") and lines[2] == "120 90 0"
        assert lines[3] == f"{lo[0] + 1} {ce[0] + 1} {al[0]}"
        o1 = oracle_lib.Oracle.from_mtx(a, r, 2, 2)
        o2 = oracle_lib.Oracle.from_coo(120, 90, lo, ce, al, re, 2, 2)
        assert o1.nnz == o2.nnz and np.array_equal(o1.locus_ids(), o2.locus_ids())
        assert np.array_equal(o1.row_ptr(), o2.row_ptr())
        for x, y in zip(o1.entries(), o2.entries()):
            assert np.array_equal(x, y)
    # the same pair block-compressed (bgzip layout, several members): a multi-member gzip like any other to the oracle's zlib reader
    # (load_data.rs:246: MultiGzDecoder); the product inflates such files block-parallel (tests/test_gpu_parity.py)
    a, r = synth.write_mtx_pair(str(tmp_path / "plain2"), 120, 90, lo, ce, al, re, header_nnz=0)
    paths = []
    for src in (a, r):
        dst = src + ".gz"
        open(dst, "wb").write(synth.bgzf_compress(open(src, "rb").read(), block=700))
        paths.append(dst)
    o3 = oracle_lib.Oracle.from_mtx(paths[0], paths[1], 2, 2)
    assert o3.nnz == o2.nnz and np.array_equal(o3.row_ptr(), o2.row_ptr())
    for x, y in zip(o3.entries(), o2.entries()):
        assert np.array_equal(x, y)


def test_oracle_mtx_text_contract(tmp_path, oracle_lib):
    """quirk Q11: dims from the ref header, indices from the alt file, ref indices ignored, nnz ignored, zip stops at
    the shorter file; a float token is a parse error like parse::<usize>()."""
    alt = tmp_path / "alt.mtx"
    ref = tmp_path / "ref.mtx"
    alt.write_text("%%MatrixMarket\n%\n99 99 99\n1 1 2\n1 2 1\n2 1 0\n2 2 3\n2 3 1\n")
    ref.write_text("%%MatrixMarket\n%\n2 3 0\n7 7 1\n8 8 0\n9 9 4\n1 1 1\n")  # shorter; indices nonsense
    o = oracle_lib.Oracle.from_mtx(str(alt), str(ref), 1, 1)
    assert (o.total_loci, o.total_cells) == (2, 3)
    assert o.loci_used == 2 and o.nnz == 4
    assert np.array_equal(o.locus_counts(), np.array([[1.0, 3.0], [5.0, 3.0]]))
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket\n%\n2 3 0\n1 1 1.0\n")
    with pytest.raises(RuntimeError):
        oracle_lib.Oracle.from_mtx(str(bad), str(bad), 1, 1)
    with pytest.raises(RuntimeError, match="couldn't open file"):
        oracle_lib.Oracle.from_mtx(str(tmp_path / "nope.mtx"), str(ref), 1, 1)


def test_locus_filter_counts_cells_not_reads(oracle_lib):
    """quirk Q3: min_alt / min_ref count CELLS with >= 1 read."""
    lo = np.array([0, 0, 0, 1, 1, 1], np.uint32)
    ce = np.array([0, 1, 2, 0, 1, 2], np.uint32)
    al = np.array([9, 0, 0, 1, 1, 0], np.uint32)   # locus 0: 9 alt reads but ONE alt cell
    re = np.array([1, 1, 1, 1, 0, 1], np.uint32)
    o = oracle_lib.Oracle.from_coo(2, 3, lo, ce, al, re, min_alt=2, min_ref=2)
    assert o.loci_used == 1 and o.locus_ids().tolist() == [1]
    assert o.entries_per_cell().tolist() == [1, 1, 1]


def test_zero_locus_cells_enter_the_quartiles(oracle_lib):
    """quirk Q4: a cell without used loci has normalised LL 0.0 and still counts in median/quartiles."""
    lo, ce, al, re = synth.generate_coo(300, 200, 0.2, seed=8)
    keep = ce < 150                        # cells 150..199 have no entries at all
    o = oracle_lib.Oracle.from_coo(300, 200, lo[keep], ce[keep], al[keep], re[keep])
    s = o.em_iteration(5.0)
    out = o.cell_outputs()
    assert np.all(out["normalized"][150:] == 0.0) and np.all(out["loci_used"][150:] == 0.0)
    assert s.median == oracle_lib.median(out["normalized"])
    p = o.posteriors()
    pa, aa, q = o.assignments(p["posterior"], p["doublet_posterior"])
    assert np.all(pa[150:] == 3)           # unassigned: fewer than min_loci_for_assignment entries (Q5)


def test_assignment_rule_matches_oracle(oracle_lib):
    lo, ce, al, re = synth.generate_coo(800, 400, 0.15, seed=6, minority_fraction=0.1, doublet_fraction=0.05)
    o = oracle_lib.Oracle.from_coo(800, 400, lo, ce, al, re)
    o.run()
    p = o.posteriors()
    want = o.assignments(p["posterior"], p["doublet_posterior"], 0.999, 30)
    got = ffi.assignments(p["posterior"], p["doublet_posterior"], o.entries_per_cell(), o.excluded(), 0.999, 30)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    special = np.array([0.0, 1.0, 0.5, np.nan, 0.9995, 1e-12])
    got = ffi.assignments(special, np.zeros(6), np.full(6, 100), np.zeros(6), 0.999, 30)
    assert got[0].tolist() == [1, 0, 3, 3, 0, 1]
    assert got[2].tolist() == [255, 255, 3, 255, 33, 120]   # inf -> 255, NaN.min(255) = 255 (f64::min), 33.01 -> 33


def test_cfg1_oracle_regression():
    """BASELINE cfg1 (1k cells x 2k loci, 10% density): the oracle's loop outcome is frozen in a committed fixture
    (made by the oracle itself at round 1 — a regression guard for the checker, not a reference-made vector)."""
    from oracle import binding as ob
    lo, ce, al, re = synth.generate_coo(2000, 1000, 0.10, seed=4)
    o = ob.Oracle.from_coo(2000, 1000, lo, ce, al, re)
    summaries = o.run()
    p = o.posteriors()
    pa, aa, q = o.assignments(p["posterior"], p["doublet_posterior"])
    got = {
        "coo_sha1": hashlib.sha1(b"".join(x.tobytes() for x in (lo, ce, al, re))).hexdigest(),
        "loci_used": int(o.loci_used), "nnz": int(o.nnz), "iterations": len(summaries),
        "thresholds": [s.threshold for s in summaries], "medians": [s.median for s in summaries],
        "n_new": [int(s.n_new_excluded) for s in summaries],
        "excluded_cells": np.nonzero(o.excluded())[0].tolist(),
        "assignment_counts": np.bincount(pa, minlength=4).tolist(),
        "posterior_sum": float(p["posterior"].sum()),
    }
    path = os.path.join(GOLD, "cfg1_oracle.json")
    if os.environ.get("CELLECTOR_REGEN_GOLDEN"):
        json.dump(got, open(path, "w"), indent=1)
    want = json.load(open(path))
    assert got["coo_sha1"] == want["coo_sha1"]
    for k in ("loci_used", "nnz", "iterations", "n_new", "excluded_cells", "assignment_counts"):
        assert got[k] == want[k], k
    assert got["thresholds"] == pytest.approx(want["thresholds"], abs=1e-12)
    assert got["medians"] == pytest.approx(want["medians"], abs=1e-12)
    assert got["posterior_sum"] == pytest.approx(want["posterior_sum"], abs=1e-9)
    cls = synth.cell_classes(1000, seed=4)
    assert sorted(np.nonzero(cls == 1)[0].tolist()) == want["excluded_cells"]  # finds exactly the planted minority


def test_bgzf_writer_makes_valid_multi_member_gzip():
    """synth.bgzf_compress (the test inputs of the block-parallel .gz reader): every member carries the 'BC' size field and the
    whole file inflates to the text with any multi-member gzip reader (the reference uses flate2's MultiGzDecoder)."""
    import gzip
    import struct

    from cellector_amd import synth
    text = b"".join(b"%d %d %d\n" % (i, i % 97 + 1, i % 5) for i in range(1, 60000))
    blob = synth.bgzf_compress(text, block=5000)
    assert gzip.decompress(blob) == text
    pos, blocks, total = 0, 0, 0
    while pos < len(blob):
        assert blob[pos:pos + 4] == b"\x1f\x8b\x08\x04"
        xlen, = struct.unpack_from("<H", blob, pos + 10)
        assert blob[pos + 12:pos + 14] == b"BC" and xlen == 6
        bsize, = struct.unpack_from("<H", blob, pos + 16)
        isize, = struct.unpack_from("<I", blob, pos + bsize + 1 - 4)
        total += isize
        pos += bsize + 1
        blocks += 1
    assert pos == len(blob) and total == len(text) and blocks == (len(text) + 4999) // 5000 + 1  # (+ the empty end-of-file member)
