#!/bin/bash
# usage (on the GPU box, through gpurun): tools/gpu_check.sh <tag> [bench args...]
# runs the GPU test-suite, then the bench under rocprofv3 --kernel-trace --stats, then a plain bench line
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?
echo pytest=$rc; tail -3 $out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline "$@" > $out/trace.log 2>&1; echo trace=$?
timeout -k 10 600 python3 bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err; echo bench=$?
python3 - <<PY
import json,glob,csv
d=json.load(open("$out/bench.json"))
print("ms_per_step", d["ms_per_step"], "kernels", d["kernels_ms"], "frac", d["roofline"]["frac"])
f=glob.glob("$out/trace/*/*_kernel_stats.csv")
if f:
    rows=list(csv.DictReader(open(f[0])))
    for r in rows:
        n=r["Name"]
        if any(k in n for k in ("k_tile_ll","k_locus","k_minority","k_ovf_v","k_ovf_t","k_ovf_c","k_cell_f","k_sel","k_flag","k_build_t","k_alpha","k_pack","k_thr","k_post")):
            print(f"  {n[:60]:60s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
