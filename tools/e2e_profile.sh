#!/bin/bash
# usage (GPU box): bash tools/e2e_profile.sh <cells> <loci> <tag>  — writes a text pair with the device writer, runs host/cellector
# on it twice, the second time under rocprofv3 --kernel-trace --stats, prints the longest kernels
export TMPDIR=/tmp
N=$1; L=$2; tag=$3
d=/dev/shm/cellector_prof_$$; mkdir -p $d gpurun_out/$tag
python3 - <<PY
import sys
sys.path.insert(0, ".")
from cellector_amd import Cellector, synth
with Cellector(0) as g:
    g.set_option("keep_coo", 1)
    g.ingest_synthetic($L, $N, 0.01, seed=4)
    g.write_staged_mtx("$d/alt.mtx", "$d/ref.mtx")
synth.write_barcodes("$d/barcodes.tsv", $N)
PY
make -C host -s
CELLECTOR_TIMING=1 host/cellector -a $d/alt.mtx -r $d/ref.mtx -b $d/barcodes.tsv --output_directory $d/out --device 0 > /dev/null 2> gpurun_out/$tag/timing1.txt
CELLECTOR_TEARDOWN=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- host/cellector -a $d/alt.mtx -r $d/ref.mtx -b $d/barcodes.tsv --output_directory $d/out --device 0 > gpurun_out/$tag/run.log 2>&1
python3 tools/kstats.py gpurun_out/$tag/trace 14
cat gpurun_out/$tag/timing1.txt | head -12
rm -rf $d gpurun_out/$tag/trace
