#!/bin/bash
# usage (on the GPU box, through gpurun): bash tools/gpu_profile.sh <tag>
# One profile set of the default bench (cfg4, 1 GPU): rocprofv3 kernel-trace stats, two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE: they do not fit one pass), the plain bench line with the CPU baseline, and the cfg3 line.
# tools/make_profiles.py turns gpurun_out/<tag>/ into the files under profiles/.
export TMPDIR=/tmp
tag=$1
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline > $out/trace.log 2>&1 || { echo trace failed; tail -5 $out/trace.log; exit 1; }
echo trace ok
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/pmc_fetch.log 2>&1 || { echo pmc fetch failed; tail -5 $out/pmc_fetch.log; exit 1; }
echo fetch ok
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/pmc_write.log 2>&1 || { echo pmc write failed; tail -5 $out/pmc_write.log; exit 1; }
echo write ok
# keep only the small summaries (the raw traces are large)
for d in pmc_fetch pmc_write; do
python3 - "$out/$d" <<'PY'
import csv, glob, collections, sys, json
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    a = acc[(r["Kernel_Name"], r["Counter_Name"])]
    a[0] += float(r["Counter_Value"]); a[1] += 1
json.dump([{"kernel": k, "counter": c, "mean": v[0] / v[1], "launches": v[1]} for (k, c), v in acc.items()],
          open(sys.argv[1] + "_summary.json", "w"), indent=1)
PY
rm -rf $out/$d
done
cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv && rm -rf $out/trace
timeout -k 10 900 python3 bench.py > $out/bench_cfg4.json 2> $out/bench_cfg4.err || { echo bench failed; tail -5 $out/bench_cfg4.err; exit 1; }
timeout -k 10 600 python3 bench.py --workload cfg3 > $out/bench_cfg3.json 2> $out/bench_cfg3.err || { echo bench cfg3 failed; exit 1; }
echo done
