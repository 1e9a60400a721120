export TMPDIR=/tmp
out=gpurun_out/$1; mkdir -p $out
for o in "tile_groups=0" "tile_groups=8" "tile_groups=16" "tile_groups=32"; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --opt $o > $out/b_$o.json 2> $out/b_$o.err || { tail -3 $out/b_$o.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/b_$o.json')); print('$o', 'ms', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernels_ms'].items()}, 'launch', round(d['roofline']['launch_ms'],4))"
done
for o in "tile_groups=0" "tile_groups=8" "tile_groups=24" "tile_groups=16"; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload cfg3 --opt $o > $out/c_$o.json 2> $out/c_$o.err || { tail -3 $out/c_$o.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/c_$o.json')); print('cfg3 $o', 'ms', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernels_ms'].items()}, 'launch', round(d['roofline']['launch_ms'],4))"
done
