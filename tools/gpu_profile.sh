#!/bin/bash
# usage (on the GPU box, through gpurun): bash tools/gpu_profile.sh <tag>
# One profile set of the default bench (cfg4, 1 GPU) and of BASELINE configs[2] (cfg3): rocprofv3 kernel-trace stats, two
# separate PMC passes each (FETCH_SIZE, WRITE_SIZE: they do not fit one pass), and the plain bench lines (cfg4 with the CPU
# baseline).  tools/make_profiles.py turns gpurun_out/<tag>/ into the files under profiles/.
export TMPDIR=/tmp
tag=$1
out=gpurun_out/$tag; mkdir -p $out
summarise() {  # $1 = directory of a --pmc run: per-kernel means -> <dir>_summary.json
python3 - "$1" <<'PY'
import csv, glob, collections, sys, json
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    a = acc[(r["Kernel_Name"], r["Counter_Name"])]
    a[0] += float(r["Counter_Value"]); a[1] += 1
json.dump([{"kernel": k, "counter": c, "mean": v[0] / v[1], "launches": v[1]} for (k, c), v in acc.items()],
          open(sys.argv[1] + "_summary.json", "w"), indent=1)
PY
rm -rf "$1"
}
for w in cfg4 cfg3; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$w -- python3 bench.py --no-cpu-baseline --workload $w > $out/trace_$w.log 2>&1 || { echo trace $w failed; tail -5 $out/trace_$w.log; exit 1; }
  cp $out/trace_$w/*/*_kernel_stats.csv $out/kernel_stats_$w.csv && rm -rf $out/trace_$w
  echo trace $w ok
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${c}_$w -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $w > $out/pmc_${c}_$w.log 2>&1 || { echo pmc $c $w failed; tail -5 $out/pmc_${c}_$w.log; exit 1; }
    summarise $out/pmc_${c}_$w
    echo $c $w ok
  done
done
timeout -k 10 900 python3 bench.py > $out/bench_cfg4.json 2> $out/bench_cfg4.err || { echo bench failed; tail -5 $out/bench_cfg4.err; exit 1; }
timeout -k 10 600 python3 bench.py --workload cfg3 > $out/bench_cfg3.json 2> $out/bench_cfg3.err || { echo bench cfg3 failed; exit 1; }
echo done
