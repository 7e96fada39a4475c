#!/bin/bash
# rocprofv3 passes over the default bench.py run (kernel trace + stats, then separate PMC passes); outputs under gpurun_out/$1
set -e
tag=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag/stats -o s --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $R/gpurun_out/$tag/bench_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/$tag/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$tag/bench_pmc_$c.log 2>&1
done
