# kernel stats of a short driver-command bench (no PMC): bash tools/probes/stats_short.sh <tag> [bench args]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag/stats -o s --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --no-mesh-116 --sustained-iters 0 "$@" > $R/gpurun_out/$tag/bench_stats.log 2>&1
cd $R
cp $(find gpurun_out/$tag/stats -name 's_kernel_stats.csv' | head -1) gpurun_out/$tag/kernel_stats.csv
rm -rf gpurun_out/$tag/stats
grep "^{" gpurun_out/$tag/bench_stats.log > gpurun_out/$tag/bench_under_profiler.json
head -8 gpurun_out/$tag/kernel_stats.csv | cut -c1-200
