# A/B of the default library against two alternative builds on one bench.py workload (alternating processes, same box):
#   bash tools/probes/ab_libs3.sh <tag> <alt lib 1> <alt lib 2> <bench args...>
tag=$1; a1=$2; a2=$3; shift 3
mkdir -p gpurun_out/$tag
for rep in 1 2; do
  for lib in default $a1 $a2; do
    if [ $lib = default ]; then unset BP5_LIB; else export BP5_LIB=$PWD/$lib; fi
    n=$(basename $lib .so)
    python3 bench.py "$@" --steps 30 --no-cpu-baseline --no-traffic-pass --sustained-iters 0 --no-mesh-116 > gpurun_out/$tag/${n}_$rep.json 2> gpurun_out/$tag/${n}_$rep.err
    python3 -c "
import json;d=json.loads(open('gpurun_out/$tag/${n}_$rep.json').read().strip().splitlines()[-1]);print('$n',$rep,round(d['value']/1e9,3),'GDoF/s',round(d['ms_per_step'],4),'ms',d['roofline']['kernel'],round(d['roofline']['avg_launch_ms'],4))"
  done
done
