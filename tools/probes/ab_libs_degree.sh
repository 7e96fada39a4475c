# A/B of two library builds on one workload of bench.py (alternating processes, same box): bash tools/probes/ab_libs_degree.sh <tag> <alt lib> <bench args...>
tag=$1; alt=$2; shift 2
mkdir -p gpurun_out/$tag
for rep in 1 2; do
  for lib in default alt; do
    if [ $lib = alt ]; then export BP5_LIB=$PWD/$alt; else unset BP5_LIB; fi
    python3 bench.py "$@" --steps 30 --no-cpu-baseline --no-traffic-pass --sustained-iters 0 > gpurun_out/$tag/${lib}_$rep.json 2> gpurun_out/$tag/${lib}_$rep.err
    python3 -c "
import json;d=json.loads(open('gpurun_out/$tag/${lib}_$rep.json').read().strip().splitlines()[-1]);print('$lib',$rep,round(d['value']/1e9,3),'GDoF/s',round(d['ms_per_step'],4),'ms',d['roofline']['kernel'],round(d['roofline']['avg_launch_ms'],4))"
  done
done
