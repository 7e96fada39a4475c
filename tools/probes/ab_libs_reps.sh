# A/B of the default library against alternative builds, REPS alternating rounds of one bench.py workload (one process per run, same box):
#   REPS=4 bash tools/probes/ab_libs_reps.sh <tag> "<alt lib> [<alt lib> ...]" <bench args...>
tag=$1; alts=$2; shift 2
mkdir -p gpurun_out/$tag
for rep in $(seq 1 ${REPS:-4}); do
  for lib in default $alts; do
    if [ $lib = default ]; then unset BP5_LIB; else export BP5_LIB=$PWD/$lib; fi
    n=$(basename $lib .so)
    python3 bench.py "$@" --steps 30 --no-cpu-baseline --no-traffic-pass --sustained-iters 0 --no-mesh-116 > gpurun_out/$tag/${n}_$rep.json 2> gpurun_out/$tag/${n}_$rep.err
    python3 -c "
import json;d=json.loads(open('gpurun_out/$tag/${n}_$rep.json').read().strip().splitlines()[-1]);print('$n',$rep,round(d['value']/1e9,3),'GDoF/s',round(d['ms_per_step'],4),'ms',d['roofline']['kernel'],round(d['roofline']['avg_launch_ms'],4))"
  done
done
