#!/bin/bash
# usage: bash tools/gpu_pmc_all.sh <tag> "<counters>" [bench args]  — one rocprofv3 --pmc pass of the bench; per-kernel means of every counter
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
    if any(k in n for k in ("synth","build","k_count","k_fill","k_pass1","rocprim","scan","k_range_offsets","k_cell_compact","k_row_ptr","k_used","k_compact","lists","nmask","locus_ids","k_t2_hist")): continue
    acc[n.split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$out/pmc_summary.txt","w") as o:
    for n,d in sorted(acc.items()):
        o.write(n+"\n")
        for c,v in d.items(): o.write("   %-28s %.4g (n=%d)\n"%(c,sum(v)/len(v),len(v)))
print(open("$out/pmc_summary.txt").read())
PY
rm -rf $out/pmc
