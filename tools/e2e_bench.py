#!/usr/bin/env python3
"""End-to-end time of the drop-in binary on vartrix-style text files, BASELINE configs[4] shape (text ingest, the whole
EM loop, posteriors, `--vcf` genotype output, every TSV) at a size one box writes in seconds:

  python tools/e2e_bench.py [--cells 50000 --loci 50000 --density 0.01] [--cpu-threads 1]

Left: `host/cellector -a alt.mtx -r ref.mtx -b barcodes.tsv --vcf variants.vcf --output_directory out`, wall clock of the
process (device initialisation, PCIe, file output included).  Right: the CPU restatement of the reference (oracle/) doing
the same computation in-process — its two-pass loader, the loop, the posteriors, the final allele tallies — WITHOUT
writing any file, single-threaded like the reference.  Prints one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=50000)
    ap.add_argument("--loci", type=int, default=50000)
    ap.add_argument("--density", type=float, default=0.01)
    ap.add_argument("--cpu-threads", type=int, default=1)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--device-writer", action="store_true",
                    help="format the text files on the device (cellector_write_staged_mtx): for BASELINE-sized inputs")
    ap.add_argument("--tmp", default=None, help="directory for the input / output files (default: a temp dir)")
    ap.add_argument("--devices", default=None, help="passed to host/cellector as --devices (e.g. 0,0,0,0: logical shards of GPU 0)")
    args = ap.parse_args()
    import ctypes
    from cellector_amd import Cellector, synth
    here = os.path.dirname(os.path.abspath(__file__))
    lib_path = os.path.join(tempfile.gettempdir(), "libfastmtx.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", lib_path, os.path.join(here, "fastmtx.c")])
    fast = ctypes.CDLL(lib_path)
    fast.fastmtx_write.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64] + [ctypes.c_void_p] * 3
    N, L = args.cells, args.loci
    d = tempfile.mkdtemp(prefix="cellector_e2e_", dir=args.tmp)
    paths = {name: os.path.join(d, name + ".mtx") for name in ("alt", "ref")}
    t_write = time.perf_counter()
    if args.device_writer:
        with Cellector(0) as g:
            g.set_option("keep_coo", 1)
            g.ingest_synthetic(L, N, args.density, seed=4)
            g.write_staged_mtx(paths["alt"], paths["ref"])
        with open(paths["ref"], "rb") as f:
            f.readline(); f.readline()
            n_entries = int(f.readline().split()[2])
    else:
        with Cellector(0) as g:
            g.load_synthetic(L, N, args.density, seed=4, min_alt=0, min_ref=0)
            rp, ent = g.csr_rows(0, N)
        cell = np.repeat(np.arange(N, dtype=np.int64), np.diff(rp.astype(np.int64)))
        locus = (ent & np.uint64(0xFFFFFFFF)).astype(np.int64)
        alt = ((ent >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
        ref = (ent >> np.uint64(48)).astype(np.int64)
        order = np.argsort(locus, kind="stable")
        locus, cell, alt, ref = locus[order] + 1, cell[order] + 1, alt[order], ref[order]
        for name, vals in (("alt", alt), ("ref", ref)):
            hdr = f"%%MatrixMarket matrix coordinate real general\n% written by sprs\n{L} {N} {len(locus)}\n"
            vals = np.ascontiguousarray(vals)
            assert fast.fastmtx_write(paths[name].encode(), hdr.encode(), len(locus), locus.ctypes.data, cell.ctypes.data,
                                      vals.ctypes.data) == 0
        n_entries = int(len(locus))
        del locus, cell, alt, ref, order
    t_write = time.perf_counter() - t_write
    bc = os.path.join(d, "barcodes.tsv")
    synth.write_barcodes(bc, N)
    vcf = os.path.join(d, "variants.vcf")
    with open(vcf, "w") as f:
        f.write("##fileformat=VCFv4.2\n##source=synthetic\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        f.write("".join(f"chr{1 + l % 22}\t{1000 + 37 * l}\t.\tA\tG\t50\tPASS\t.\n" for l in range(L)))
    out_dir = os.path.join(d, "out")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    cmd = [os.path.join(ROOT, "host", "cellector"), "-a", paths["alt"], "-r", paths["ref"], "--output_directory", out_dir,
           "--min_alt", "4", "--min_ref", "4", "--barcodes", bc, "--vcf", vcf, "--min_alleles_posterior", "5"]
    if args.devices:
        cmd += ["--devices", args.devices]
    times = []
    for _ in range(2):  # second run: page cache and driver warm, like the oracle's files
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, CELLECTOR_TIMING="1"))
        times.append(time.perf_counter() - t0)
        assert r.returncode == 0, r.stderr
    phases = [ln.split("]", 1)[1].strip() for ln in r.stderr.splitlines() if ln.startswith("[timing]")]
    n_iter = sum(1 for ln in r.stdout.splitlines() if ln.startswith("detected "))
    out_bytes = sum(os.path.getsize(os.path.join(out_dir, f)) for f in os.listdir(out_dir))
    res = {"workload": f"{N} cells x {L} loci, density {args.density}: text .mtx pair + barcodes + VCF -> all TSVs + cellector.vcf",
           "entries": n_entries, "text_bytes": int(sum(os.path.getsize(p) for p in paths.values())),
           "inputs_written_s": t_write, "device_writer": bool(args.device_writer),
           "em_iterations": n_iter, "output_bytes": int(out_bytes), "devices": args.devices,
           "gpu_cli_wall_s": times[1], "gpu_cli_first_run_s": times[0], "gpu_cli_phases": phases}
    if not args.no_oracle:
        from oracle import binding as ob
        ob.set_threads(args.cpu_threads)
        t0 = time.perf_counter()
        o = ob.Oracle.from_mtx(paths["alt"], paths["ref"], 4, 4)
        t_load = time.perf_counter() - t0
        k = 0
        while True:
            s = o.em_iteration(5.0)
            k += 1
            if not s.any_change or k >= 30:
                break
        t_loop = time.perf_counter() - t0 - t_load
        post = o.posteriors()
        o.assignments(post["posterior"], post["doublet_posterior"])
        t_all = time.perf_counter() - t0
        ob.set_threads(1)
        assert k == n_iter, (k, n_iter)
        res.update({"cpu_port_s": t_all, "cpu_port_load_s": t_load, "cpu_port_loop_s": t_loop, "cpu_threads": args.cpu_threads,
                    "cpu_note": "oracle in-process: loader + EM loop + posteriors + assignments, no file output",
                    "speedup": t_all / times[1]})
        o.close()
    print(json.dumps(res))
    subprocess.call(["rm", "-rf", d])


if __name__ == "__main__":
    main()
