export TMPDIR=/tmp
out=gpurun_out/tl8; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/shard_rehearsal.py --ranks 8 --steps 4 --warmup 2 > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
f=$(ls $out/trace/*/*_kernel_trace.csv | head -1)
python3 tools/timeline.py $f 90 > $out/timeline.txt
rm -rf $out/trace
tail -1 $out/run.log
