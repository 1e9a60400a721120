#!/usr/bin/env python3
"""Times the text ingest (alt.mtx/ref.mtx -> device matrix) against the oracle's loader on the same files.

  python tools/ingest_bench.py [--cells 50000 --loci 50000 --density 0.01] [--gz]

Generates BASELINE cfg2 on the device, writes it as vartrix-style text (locus-major, 3-line header), then times
cellector_load_mtx (host: bytes only; device: tokenise, pass 1, CSC/CSR, tiled layouts) and orc_load_mtx (the CPU
restatement of the reference's 2-pass loader).  PCIe transfers are inside the GPU number."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=50000)
    ap.add_argument("--loci", type=int, default=50000)
    ap.add_argument("--density", type=float, default=0.01)
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--bgzf", action="store_true", help="block-compressed .gz (what bgzip writes): inflated block-parallel")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--engine2-only", action="store_true")
    args = ap.parse_args()
    import ctypes
    import subprocess
    from cellector_amd import Cellector
    here = os.path.dirname(os.path.abspath(__file__))
    lib_path = os.path.join(tempfile.gettempdir(), "libfastmtx.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", lib_path, os.path.join(here, "fastmtx.c")])
    fast = ctypes.CDLL(lib_path)
    fast.fastmtx_write.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64] + [ctypes.c_void_p] * 3
    t_gen = time.perf_counter()
    N, L = args.cells, args.loci
    with Cellector(0) as g:
        g.load_synthetic(L, N, args.density, seed=4, min_alt=0, min_ref=0)
        rp, ent = g.csr_rows(0, N)
    cell = np.repeat(np.arange(N, dtype=np.int64), np.diff(rp.astype(np.int64)))
    locus = (ent & np.uint64(0xFFFFFFFF)).astype(np.int64)
    alt = ((ent >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
    ref = (ent >> np.uint64(48)).astype(np.int64)
    order = np.argsort(locus, kind="stable")  # rows are in cell order already: locus-major, cells ascending
    locus, cell, alt, ref = locus[order] + 1, cell[order] + 1, alt[order], ref[order]
    del order
    d = tempfile.mkdtemp(prefix="cellector_ingest_")
    ext = ".mtx.gz" if (args.gz or args.bgzf) else ".mtx"
    paths = {}
    for name, vals in (("alt", alt), ("ref", ref)):
        p = os.path.join(d, name + ext)
        hdr = f"%%MatrixMarket matrix coordinate real general\n% written by sprs\n{L} {N} {len(locus)}\n"
        plain = p[:-3] if (args.gz or args.bgzf) else p
        vals = np.ascontiguousarray(vals)
        rc = fast.fastmtx_write(plain.encode(), hdr.encode(), len(locus), locus.ctypes.data, cell.ctypes.data, vals.ctypes.data)
        assert rc == 0, "mtx write failed"
        if args.bgzf:
            from cellector_amd import synth
            open(p, "wb").write(synth.bgzf_compress(open(plain, "rb").read()))
            os.remove(plain)
        elif args.gz:
            subprocess.check_call(["gzip", "-1", "-f", plain])
        paths[name] = p
    t_gen = time.perf_counter() - t_gen
    size = sum(os.path.getsize(p) for p in paths.values())
    out = {"cells": N, "loci": L, "entries": int(len(locus)), "text_bytes": int(size), "gz": args.gz, "bgzf": args.bgzf,
           "generate_and_write_s": t_gen}
    n_entries = int(len(locus))
    del locus, cell, alt, ref
    for engine in ((2,) if args.engine2_only else (2, 1)):
        with Cellector(0) as g:
            g.set_option("engine", engine)
            g.set_option("keep_coo", 0)
            t0 = time.perf_counter()
            g.load_mtx(paths["alt"], paths["ref"], 4, 4)
            dt = time.perf_counter() - t0
            out[f"gpu_engine{engine}_s"] = dt
            out[f"gpu_engine{engine}_entries_per_s"] = n_entries / dt
            out[f"gpu_engine{engine}_text_GBps"] = size / dt / 1e9
            nnz = g.dims().nnz_used
    if not args.no_oracle:
        from oracle import binding as ob
        t0 = time.perf_counter()
        o = ob.Oracle.from_mtx(paths["alt"], paths["ref"], 4, 4)
        out["oracle_s"] = time.perf_counter() - t0
        out["oracle_entries_per_s"] = n_entries / out["oracle_s"]
        assert o.nnz == nnz
        out["speedup_engine2"] = out["oracle_s"] / out["gpu_engine2_s"]
    print(json.dumps(out))
    for p in paths.values():
        os.remove(p)
    os.rmdir(d)


if __name__ == "__main__":
    main()
