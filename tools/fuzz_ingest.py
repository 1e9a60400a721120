#!/usr/bin/env python3
"""Randomised check of the multi-device text ingest (needs an MI355X): random small matrices written as vartrix-style text
pairs with random quirks (CRLF, '+' signs, blanks around tokens, unterminated last line, a ref file that is shorter or has
another header length, entries out of locus order), loaded by a single-device ctx and by multi-device ctxs of 2..7 logical
shards in both forms of the ingest (routed: one parser; split: every shard tokenises a window range of both files) with
random window sizes.  The staged matrices must be identical (dims, locus ids / counts, every CSR row in file order).

  python tools/fuzz_ingest.py [--cases 60] [--seed 1]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_pair(d, rng, L, N, lo, ce, al, re):
    quirks = {"crlf": rng.random() < 0.2, "plus": rng.random() < 0.2, "blanks": rng.random() < 0.2,
              "unterminated": rng.random() < 0.5, "ref_short": rng.random() < 0.3, "long_ref_header": rng.random() < 0.5}
    eol = "\r\n" if quirks["crlf"] else "\n"

    def tok(v):
        s = str(int(v))
        if quirks["plus"] and rng.random() < 0.3:
            s = "+" + s
        return s

    def lines(vals, n_keep):
        out = []
        for i in range(n_keep):
            sep = "  \t " if quirks["blanks"] and rng.random() < 0.3 else " "
            out.append(tok(lo[i] + 1) + sep + tok(ce[i] + 1) + sep + tok(vals[i]))
        return out

    n = len(lo)
    n_ref = n - int(rng.integers(1, max(2, n // 3))) if quirks["ref_short"] and n > 3 else n
    a_path, r_path = os.path.join(d, "alt.mtx"), os.path.join(d, "ref.mtx")
    body = eol.join(lines(al, n))
    with open(a_path, "w", newline="") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% written by sprs\n" + f"{L} {N} {n}\n" + body +
                ("" if quirks["unterminated"] or not body else eol))
    hdr2 = "% " + "x" * int(rng.integers(1, 300)) + "\n" if quirks["long_ref_header"] else "% written by sprs\n"
    rbody = eol.join(lines(re, n_ref))
    with open(r_path, "w", newline="") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n" + hdr2 + f"{L} {N} {n}\n" + rbody + (eol if rbody else ""))
    return a_path, r_path, quirks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from cellector_amd import Cellector, synth
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    for case in range(args.cases):
        N = int(rng.choice([3, 40, 700, 1500]))  # (the text is written token by token in Python: keep it small)
        L = int(rng.choice([20, 300, 1500]))
        dens = float(rng.choice([0.01, 0.05, 0.3]))
        lo, ce, al, re = synth.generate_coo(L, N, dens, seed=int(rng.integers(1, 1 << 30)), minority_fraction=0.1)
        if len(lo) < 2:
            continue
        if rng.random() < 0.3:  # not locus-major
            p = rng.permutation(len(lo))
            lo, ce, al, re = lo[p], ce[p], al[p], re[p]
        al = al.copy()
        al[rng.random(len(al)) < 0.05] += int(rng.integers(10, 60000))  # long tokens: lines of different lengths in the two files
        with tempfile.TemporaryDirectory(prefix="cellector_fuzz_ingest_") as d:
            a_path, r_path, quirks = write_pair(d, rng, L, N, lo, ce, al, re)
            mn = int(rng.choice([1, 2, 4]))
            single = Cellector(0)
            single.load_mtx(a_path, r_path, mn, mn)
            ds = single.dims()
            rs, es = single.csr_rows(0, N)
            ids, cnt = single.locus_ids(), single.locus_counts()
            for form in ("routed", "split"):
                shards = int(rng.integers(2, 8))
                window = int(rng.choice([512, 640, 1024, 4096, 1 << 15]))
                os.environ["CELLECTOR_MULTI_SPLIT"] = "1" if form == "split" else "0"
                try:
                    m = Cellector(devices=[0] * shards)
                    m.set_option("parse_window", window)
                    m.load_mtx(a_path, r_path, mn, mn)
                finally:
                    os.environ.pop("CELLECTOR_MULTI_SPLIT", None)
                dm = m.dims()
                what = (case, form, shards, window, quirks)
                assert (dm.total_loci, dm.total_cells, dm.loci_used, dm.nnz_used) == (ds.total_loci, ds.total_cells, ds.loci_used, ds.nnz_used), what
                assert np.array_equal(m.locus_ids(), ids) and np.array_equal(m.locus_counts(), cnt), what
                rm, em = m.csr_rows(0, N)
                assert np.array_equal(rm, rs) and np.array_equal(em, es), what
                if ds.loci_used and ds.nnz_used:
                    s1, s2 = single.em_iteration(5.0) if form == "routed" else None, None
                m.close()
            single.close()
        print(f"case {case}: N={N} L={L} d={dens} nnz={len(lo)} {[k for k, v in quirks.items() if v]}: ok", flush=True)
    print(f"{args.cases} cases ok in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
