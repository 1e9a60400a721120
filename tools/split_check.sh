#!/bin/bash
# usage (GPU box): bash tools/split_check.sh <cells> <loci> "<devices>" — the CLI's text ingest on a multi-device ctx, routed
# (one parser, CELLECTOR_MULTI_SPLIT=0) against split (every shard tokenises a byte range, =1): the outputs must be byte-identical
export TMPDIR=/tmp
N=$1; L=$2; DEV=$3
d=/dev/shm/cellector_splitchk_$$; mkdir -p $d
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
for mode in 0 1 0 1; do
  [ $mode = 1 ] && for mb in 16 32 64 128 256; do CELLECTOR_SPLIT_WINDOW_MB=$mb CELLECTOR_MULTI_SPLIT=1 CELLECTOR_TIMING=1 host/cellector -a $d/alt.mtx -r $d/ref.mtx -b $d/barcodes.tsv --output_directory $d/mx --devices $DEV 2>&1 >/dev/null | grep load_mtx | sed "s/^/window $mb MB: /"; done
  CELLECTOR_MULTI_SPLIT=$mode CELLECTOR_TIMING=1 host/cellector -a $d/alt.mtx -r $d/ref.mtx -b $d/barcodes.tsv --output_directory $d/m$mode --devices $DEV > $d/m$mode.out 2> $d/m$mode.err || { echo "mode $mode FAILED"; tail -5 $d/m$mode.err; }
  echo "split=$mode: $(grep -E "split:|load_mtx" $d/m$mode.err | tr '\n' ';')"
done
for f in $(ls $d/m0); do cmp -s $d/m0/$f $d/m1/$f && echo "$f byte-identical" || echo "$f DIFFERS"; done
cmp -s $d/m0.out $d/m1.out && echo "stdout byte-identical" || echo "stdout DIFFERS"
rm -rf $d
