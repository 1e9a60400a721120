#!/bin/bash
# usage (on the GPU box, through gpurun): bash tools/gpu_quick.sh <tag> [pytest args]
# The parity file of the GPU suite (or the given pytest selection), then what tools/gpu_timeline.sh collects: kernel timelines of
# rank 0's shard at 1 and 8 ranks, the default bench line and the shard rehearsal.
export TMPDIR=/tmp
out=gpurun_out/$1; mkdir -p $out; shift
sel=${@:-tests/test_gpu_parity.py}
timeout -k 10 600 python -m pytest $sel -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?
echo pytest=$rc; tail -3 $out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
for R in 1 8; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace$R -- python3 tools/shard_rehearsal.py --ranks $R --steps 4 --warmup 2 > $out/run$R.log 2>&1 || { tail -5 $out/run$R.log; exit 1; }
f=$(ls $out/trace$R/*/*_kernel_trace.csv | head -1)
python3 tools/timeline.py $f 40 > $out/timeline$R.txt
rm -rf $out/trace$R
done
timeout -k 10 600 python3 bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/bench.json')); print('ms', d['ms_per_step'], d['kernels_ms'], 'frac', d['roofline']['frac'], 'launch', d['roofline']['launch_ms'])"
timeout -k 10 600 python3 tools/shard_rehearsal.py > $out/rehearsal.json 2> $out/rehearsal.err || { tail -5 $out/rehearsal.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/rehearsal.json')); print([round(r['ms_per_step'],3) for r in d['results']])"
