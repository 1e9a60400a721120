#!/bin/bash
# usage: bash tools/gpu_ab_env.sh <tag> "ENV=V bench args" ...  — like gpu_ab.sh (no tracer), variants may start with VAR=value words
export TMPDIR=/tmp
tag=$1; shift
i=0
for v in "$@"; do
  out=gpurun_out/$tag/v$i; mkdir -p $out
  envs=""; args=""
  for w in $v; do case "$w" in [A-Z_]*=*) envs="$envs $w";; *) args="$args $w";; esac; done
  env $envs timeout -k 10 300 python3 bench.py --no-cpu-baseline $args > $out/bench.json 2> $out/bench.err || { echo "variant $i failed"; tail -5 $out/bench.err; exit 1; }
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/bench.json") if l.startswith("{")][-1])
r=d["ms_per_step_repeats"]
print("== variant $i [$v] ms_per_step %.3f (min %.3f median %.3f)"%(d["ms_per_step"], r["min"], r["median"]), {k:round(x,3) for k,x in d["kernels_ms"].items()}, "launch_ms %.3f"%d["roofline"]["launch_ms"])
PY
  i=$((i+1))
done
