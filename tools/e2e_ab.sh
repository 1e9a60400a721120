#!/bin/bash
# usage (GPU box): bash tools/e2e_ab.sh <cells> <loci> "<ENV=a>" "<ENV=b>" ... — the CLI on one device-written text pair, run in turn
# under each environment setting (three rounds), wall time and the [timing] phases that matter
export TMPDIR=/tmp
N=$1; L=$2; shift; shift
d=/dev/shm/cellector_e2eab_$$; mkdir -p $d
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
make -C host -s
for round in 1 2 3; do
  for v in "$@"; do
    t0=$(date +%s.%N)
    env $v CELLECTOR_TIMING=1 host/cellector -a $d/alt.mtx -r $d/ref.mtx -b $d/barcodes.tsv --output_directory $d/out > /dev/null 2> $d/err.txt || { echo "FAILED $v"; tail -3 $d/err.txt; }
    t1=$(date +%s.%N)
    echo "[$v] wall $(python3 -c "print('%.3f' % ($t1 - $t0))") s; $(grep -E 'load_mtx|file \(|allocations|sort by|pre-mapped|CSC / CSR' $d/err.txt | sed 's/\[timing\] *//' | tr -s ' ' | tr '\n' ';')"
  done
done
rm -rf $d
