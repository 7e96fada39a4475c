# config 4 (degree sweep) with the default library and an alternative build, alternating, Gauss only: bash tools/probes/sweep_two_libs.sh <tag> <alt lib>
tag=$1; alt=$2
mkdir -p gpurun_out/$tag
for rep in 1 2; do
  for lib in default $alt; do
    if [ $lib = default ]; then unset BP5_LIB; else export BP5_LIB=$PWD/$lib; fi
    nm=$(basename $lib .so)
    python3 bench.py --gpus 1 --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --sustained-iters 0 --no-sweep-other-quadrature > gpurun_out/$tag/${nm}_$rep.json 2> gpurun_out/$tag/${nm}_$rep.err
    python3 -c "
import json;d=json.loads(open('gpurun_out/$tag/${nm}_$rep.json').read().strip().splitlines()[-1])
print('$nm',$rep,' '.join('p%d %.2f' % (e['degree'], e['value']/1e9) for e in d['sweep']))"
  done
done
