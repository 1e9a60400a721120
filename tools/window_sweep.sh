#!/bin/bash
# usage (GPU box): bash tools/window_sweep.sh <cells> <loci> — text ingest time of the single-device CLI against the parser's window size
export TMPDIR=/tmp
N=$1; L=$2
d=/dev/shm/cellector_winsweep_$$; mkdir -p $d
python3 - <<PY
import sys
sys.path.insert(0, ".")
from cellector_amd import Cellector, synth
with Cellector(0) as g:
    g.set_option("keep_coo", 1)
    g.ingest_synthetic($L, $N, 0.01, seed=4, minority_fraction=0.05)
    g.write_staged_mtx("$d/alt.mtx", "$d/ref.mtx")
synth.write_barcodes("$d/barcodes.tsv", $N)
PY
ls -l $d/*.mtx
make -C host -s
for rep in 1 2; do
for mb in ${MBS:-8 16 32 64 128 256}; do
  CELLECTOR_WINDOW_MB=$mb CELLECTOR_TIMING=1 host/cellector -a $d/alt.mtx -r $d/ref.mtx -b $d/barcodes.tsv --output_directory $d/o --device 0 2>&1 >/dev/null | grep -E "load_mtx|alt file|ref file" | tr '\n' ';' | sed "s/^/window $mb MB: /"; echo
done; done
rm -rf $d
