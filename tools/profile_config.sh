#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py configuration: bash tools/profile_config.sh <tag> <bench args...>
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag/stats -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-traffic-pass --no-mesh-116 --sustained-iters 0 "$@" > $R/gpurun_out/$tag/bench.log 2>&1
cd $R
f=$(find gpurun_out/$tag/stats -name s_kernel_stats.csv | head -1)
head -12 $f | cut -c1-200 > gpurun_out/$tag/kernel_stats_top.csv
grep "^{" gpurun_out/$tag/bench.log > gpurun_out/$tag/bench.json || true
rm -rf gpurun_out/$tag/stats
