export TMPDIR=/tmp
out=gpurun_out/$1; mkdir -p $out
for w in cfg2 cfg3; do
for o in "tile_groups=0" "tile_groups=8" "tile_groups=16" "tile_groups=24" "tile_groups=32"; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 5 --opt $o > $out/c_$o.json 2> $out/c_$o.err || { tail -3 $out/c_$o.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/c_$o.json')); print('$w $o', 'ms', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernels_ms'].items()}, 'launch', round(d['roofline']['launch_ms'],4))"
done; done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
