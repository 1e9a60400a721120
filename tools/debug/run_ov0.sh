export TMPDIR=/tmp
out=gpurun_out/ov0; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --opt overlap=0 --steps 5 --warmup 2 > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv; rm -rf $out/trace
python3 - <<PY
import csv
for r in csv.DictReader(open("$out/kernel_stats.csv")):
    n=r["Name"]
    if any(k in n for k in ("k_tile_ll","k_ovf","k_cell_f","k_locus_f","k_minority")):
        print(f"  {n[:50]:50s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
