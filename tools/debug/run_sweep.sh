export TMPDIR=/tmp
out=gpurun_out/$1; mkdir -p $out
for o in "side_lds=-1" "side_lds=0" "side_lds=20000" "side_lds=40000" "side_lds=70000" "overlap=2"; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --opt $o > $out/b_$o.json 2> $out/b_$o.err || { tail -3 $out/b_$o.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/b_$o.json')); print('$o', 'ms', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernels_ms'].items()}, 'launch', round(d['roofline']['launch_ms'],4))"
done
