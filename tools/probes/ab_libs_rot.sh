# as ab_libs_reps.sh, but the order of the libraries rotates from round to round (a run's speed depends on what ran before it on some boxes)
tag=$1; alts=$2; shift 2
mkdir -p gpurun_out/$tag
libs=(default $alts); n=${#libs[@]}
for rep in $(seq 1 ${REPS:-4}); do
  for k in $(seq 0 $((n-1))); do
    lib=${libs[$(((k+rep)%n))]}
    if [ $lib = default ]; then unset BP5_LIB; else export BP5_LIB=$PWD/$lib; fi
    nm=$(basename $lib .so)
    python3 bench.py "$@" --steps 30 --no-cpu-baseline --no-traffic-pass --sustained-iters 0 --no-mesh-116 > gpurun_out/$tag/${nm}_$rep.json 2> gpurun_out/$tag/${nm}_$rep.err
    python3 -c "
import json;d=json.loads(open('gpurun_out/$tag/${nm}_$rep.json').read().strip().splitlines()[-1]);print('$nm',$rep,round(d['value']/1e9,3),'GDoF/s',round(d['ms_per_step'],4),'ms',d['roofline']['kernel'],round(d['roofline']['avg_launch_ms'],4))"
  done
done
