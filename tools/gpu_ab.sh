#!/bin/bash
# usage: bash tools/gpu_ab.sh <tag> "<bench args A>" "<bench args B>" ...   — one kernel-trace bench run per variant
export TMPDIR=/tmp
tag=$1; shift
i=0
for v in "$@"; do
  out=gpurun_out/$tag/v$i; mkdir -p $out
  # "lib.so|bench args": run this variant against another build of the library
  unset CELLECTOR_HIP_LIB
  case "$v" in *"|"*) export CELLECTOR_HIP_LIB="$PWD/${v%%|*}"; v="${v#*|}";; esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline $v > $out/bench.json 2> $out/bench.err || { echo "variant $i failed"; tail -5 $out/bench.err; exit 1; }
  python3 - <<PY
import json,glob,csv
line=[l for l in open("$out/bench.json") if l.startswith("{")][-1]
d=json.loads(line)
print("== variant $i [$v] ms_per_step %.3f"%d["ms_per_step"], {k:round(x,3) for k,x in d["kernels_ms"].items()})
f=glob.glob("$out/trace/*/*_kernel_stats.csv")
rows=list(csv.DictReader(open(f[0])))
tot=0
for r in rows:
    n=r["Name"]; c=int(r["Calls"])
    if c>=d["steps"] and not any(k in n for k in ("synth","build<","k_count","k_fill","k_pass1","rocprim","k_scan")):
        per=float(r["TotalDurationNs"])/1e3/(d["steps"]+d["warmup"]); tot+=per
        if per>3: print(f"   {n[:58]:58s} per_iter_us={per:9.1f}")
print("   sum of kernel time per iteration: %.1f us"%tot)
PY
  i=$((i+1))
done
