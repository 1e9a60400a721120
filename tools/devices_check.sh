#!/bin/bash
# usage (GPU box): bash tools/devices_check.sh <cells> <loci> "<devices>"  — the CLI on a device-written text pair, single device
# vs --devices: per-cell files must be byte-identical
export TMPDIR=/tmp
N=$1; L=$2; DEV=$3
d=/dev/shm/cellector_devchk_$$; mkdir -p $d
python3 - <<PY
import sys
sys.path.insert(0, ".")
from cellector_amd import Cellector, synth
with Cellector(0) as g:
    g.set_option("keep_coo", 1)
    g.ingest_synthetic($L, $N, 0.01, seed=4, minority_fraction=0.05)
    g.write_staged_mtx("$d/alt.mtx", "$d/ref.mtx")
synth.write_barcodes("$d/barcodes.tsv", $N)
open("$d/v.vcf","w").write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" + "".join(f"chr{1+l%22}\t{1000+37*l}\t.\tA\tG\t50\tPASS\t.\n" for l in range($L)))
PY
make -C host -s
host/cellector -a $d/alt.mtx -r $d/ref.mtx -b $d/barcodes.tsv --vcf $d/v.vcf --output_directory $d/one --device 0 > $d/one.out
host/cellector -a $d/alt.mtx -r $d/ref.mtx -b $d/barcodes.tsv --vcf $d/v.vcf --output_directory $d/multi --devices $DEV > $d/multi.out
python3 - <<PY
import os, numpy as np, pandas as pd
a, b = "$d/one", "$d/multi"
for f in sorted(os.listdir(a)):
    same = open(f"{a}/{f}", "rb").read() == open(f"{b}/{f}", "rb").read()
    msg = "byte-identical" if same else "differs"
    if not same and f.endswith(".tsv") and "threshold" not in f:
        x, y = pd.read_csv(f"{a}/{f}", sep="\t", float_precision="round_trip"), pd.read_csv(f"{b}/{f}", sep="\t", float_precision="round_trip")
        if "locus_id" in x.columns:
            x, y = x.sort_values("locus_id").reset_index(drop=True), y.sort_values("locus_id").reset_index(drop=True)
        num = [c for c in x.columns if x[c].dtype.kind in "fi"]
        oth = [c for c in x.columns if c not in num]
        worst = max(float(np.max(np.abs(x[c].to_numpy(float) - y[c].to_numpy(float)) / np.maximum(1.0, np.abs(x[c].to_numpy(float))))) for c in num)
        msg += f": max relative difference of the numeric columns {worst:.3g}; text columns equal: {all((x[c] == y[c]).all() for c in oth)}"
    elif not same and "threshold" in f:
        msg += f": {open(f'{a}/{f}').read()} vs {open(f'{b}/{f}').read()}"
    print(f, msg)
PY
rm -rf $d
