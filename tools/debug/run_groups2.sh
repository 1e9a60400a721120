export TMPDIR=/tmp
out=gpurun_out/$1; mkdir -p $out
for o in "tile_groups=0" "tile_groups=8" "tile_groups=16"; do
timeout -k 10 600 python3 tools/shard_rehearsal.py --ranks 2,4,8 --opt $o > $out/r_$o.json 2> $out/r_$o.err || { tail -3 $out/r_$o.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/r_$o.json')); print('$o', [(r['ranks'], round(r['ms_per_step'],3), round(r['kernels_ms']['tile_ll'],3)) for r in d['results']])"
done
