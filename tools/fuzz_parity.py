#!/usr/bin/env python3
"""Randomised parity sweep (needs an MI355X): whole runs of the HIP path against the CPU oracle on matrices of random shape,
density, minority share and count distribution — the checks of tests/test_gpu_parity.py on inputs nobody picked by hand.

  python tools/fuzz_parity.py [--cases 40] [--seed 1] [--max-cells 60000]

Every case: ingest (device generator, then counts widened at random so that all overflow tiers occur), both engines or
engine 2 with a random option set (locus-pass form, overlap, two shards), EM loop until the oracle stops, posteriors,
assignments.  Prints one line per case and a summary; exits non-zero on the first mismatch."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cells", type=int, default=60000)
    args = ap.parse_args()
    from cellector_amd import Cellector, ffi, synth
    from oracle import binding as ob
    import test_gpu_parity as T  # the checkers

    ob.build()
    ob.set_threads(ob.host_threads())
    mods = dict(ffi=ffi, synth=synth, ob=ob, engine=2)
    rng = np.random.default_rng(args.seed)
    t_all = time.time()
    for case in range(args.cases):
        N = int(rng.choice([1, 3, 70, 700, 1100, 5000, 20000, args.max_cells]))
        L = int(rng.choice([50, 400, 1500, 4200, 9000]))  # (one or two loci: every cell ties with thousands of others)
        dens = float(rng.choice([0.005, 0.02, 0.1, 0.5]))
        if N * L * dens > 6e6:
            dens = max(0.002, 6e6 / (N * L))
        # keep a locus' totals moderate: the ORACLE's ln_gamma differences (the reference's formula) lose ~1e-8 per entry
        # at alpha + beta ~ 1e6, which would drown the comparison, not the device's product form
        dens = min(dens, 3000.0 / N)
        minority = float(rng.choice([0.0, 0.02, 0.08, 0.2, 0.45]))
        doublet = float(rng.choice([0.0, 0.03]))
        seed = int(rng.integers(1, 1 << 30))
        lo, ce, al, re = synth.generate_coo(L, N, dens, seed=seed, minority_fraction=minority, doublet_fraction=doublet)
        if len(lo) and rng.random() < 0.6:  # widen counts: overflow tiers (totals 5..8, 9..17, above), zero/zero entries
            big = rng.random(len(al)) < rng.choice([0.02, 0.2])
            f = rng.integers(1, int(rng.choice([4, 12, 60])), len(al))
            al = np.where(big, al * f, al).astype(np.uint32)
            re = np.where(big, re * f, re).astype(np.uint32)
            zero = rng.random(len(al)) < 0.002
            al = np.where(zero, 0, al).astype(np.uint32)
            re = np.where(zero, 0, re).astype(np.uint32)
        min_alt, min_ref = (int(x) for x in rng.choice([[0, 0], [1, 1], [4, 4]]))
        engine = int(rng.choice([2, 2, 2, 1]))
        opts = {}
        if engine == 2:
            opts = {"locus_mode": int(rng.choice([0, 0, 1, 2])), "overlap": int(rng.choice([1, 1, 0, 2])),
                    "compact_bits": int(rng.choice([0, 0, 32])), "side_lds": int(rng.choice([-1, -1, 5000])),
                    "ovf_deep": int(rng.choice([-1, -1, 0, 1])), "ovf_deep_wide": int(rng.choice([1, 1, 0])), "t2": int(rng.choice([-1, -1, 0, 1])),
                    "bank_order": int(rng.choice([1, 1, 1, 0])), "t2_tiles": int(rng.choice([-1, -1, 0, 6, 8]))}
        two_shards = engine == 2 and N >= 2 and rng.random() < 0.25
        desc = f"case {case}: N={N} L={L} d={dens:.3g} min={minority} dbl={doublet} nnz={len(lo)} engine={engine} {opts}" \
               f"{' 2 shards' if two_shards else ''} filter=({min_alt},{min_ref})"
        t0 = time.time()
        o = ob.Oracle.from_coo(L, N, lo, ce, al, re, min_alt, min_ref)
        if two_shards:
            ok = run_two_shards(Cellector, ffi, o, L, N, lo, ce, al, re, min_alt, min_ref, opts)
        else:
            g = Cellector(0)
            g.set_option("engine", engine)
            for k, v in opts.items():
                g.set_option(k, v)
            g.load_coo(L, N, lo, ce, al, re, min_alt, min_ref)
            T._check_matrix(g, o)
            ok = True
            if o.loci_used and N:
                mods["engine"] = engine
                try:
                    T._run_both(g, o)
                    T._check_posteriors(mods, g, o)
                except AssertionError as e:
                    if "near-tie" not in str(e):
                        raise
                    ok = None  # cells sit on the threshold to 1e-9: the comparison cannot decide, not a mismatch
            g.close()
        o.close()
        print(f"{desc}: {'ok' if ok else ('undecidable (near-ties)' if ok is None else 'MISMATCH')} ({time.time() - t0:.1f} s)", flush=True)
        if ok is False:
            sys.exit(1)
    print(f"{args.cases} cases ok in {time.time() - t_all:.0f} s")


def run_two_shards(Cellector, ffi, o, L, N, lo, ce, al, re, min_alt, min_ref, opts):
    """two shard contexts driven through the exchange buffers on one GPU (host-summed), against the oracle"""
    cut = int(np.random.default_rng(N * 7919 + L).choice([0, N // 3, N // 2, N]))  # an empty shard now and then
    gs = []
    for cb, cend in ((0, cut), (cut, N)):
        g = Cellector(0)
        for k, v in opts.items():
            g.set_option(k, v)
        g.set_shard(cb, cend)
        g.ingest_coo(L, N, lo, ce, al, re)
        gs.append(g)
    import test_gpu_parity as T
    hip = T._hip()
    T._allreduce(hip, gs, ffi.XCHG_PASS1)
    for g in gs:
        g.ingest_finish(min_alt, min_ref)
    if not (o.loci_used and N):
        for g in gs:
            g.close()
        return True
    for _ in range(30):
        so = o.em_iteration(5.0)
        for g in gs:
            g.em_begin()
        T._allreduce(hip, gs, ffi.XCHG_NORM)
        for g in gs:
            g.em_threshold(5.0)
        T._allreduce(hip, gs, ffi.XCHG_LOCUS)
        outs = [g.em_finish() for g in gs]
        s = outs[0]
        exc = np.concatenate([g.excluded() for g in gs])
        if not (np.array_equal(exc, o.excluded()) and abs(s.threshold - so.threshold) < 1e-9
                and (s.any_change, s.n_new_excluded, s.n_rescued) == (so.any_change, so.n_new_excluded, so.n_rescued)
                and np.array_equal(gs[0].loci_mask(), o.loci_mask())):
            return False
        if not so.any_change:
            break
    po = o.posteriors()
    post = np.concatenate([g.posteriors()["posterior"] for g in gs])
    ok = bool(np.allclose(post, po["posterior"], rtol=0, atol=1e-6))
    for g in gs:
        g.close()
    return ok


if __name__ == "__main__":
    main()
