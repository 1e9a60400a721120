#!/bin/bash
# usage: bash tools/gpu_pmc.sh <tag> "<counters>" [bench args]   — one rocprofv3 --pmc pass of the bench, per-kernel averages
export TMPDIR=/tmp
tag=$1; ctr=$2; shift; shift
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $out/pmc -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $out/pmc.log 2>&1 || { echo pmc failed; tail -5 $out/pmc.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/pmc/*/*_counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if any(k in n for k in ("k_tile_ll","k_locus_stats2")):
        acc[n[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,d in acc.items():
    print(n)
    for c,v in d.items(): print("   %-28s %.4g (n=%d)"%(c,sum(v)/len(v),len(v)))
PY
