#!/bin/bash
# Kernel traces of ONE exchange mode each of tools/halo_overhead_self.py (the slab of rank 3 of 8, self neighbour): where does the
# time of an iteration go in the overlapped / sequential schedule against the mesh without exchange.  Outputs under gpurun_out/$1.
set -e
tag=${1:-halo}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
for mode in ${MODES:-auto overlapped launches sequential none}; do
  timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/$tag/trace_$mode -o t --output-format csv -- python3 $R/tools/halo_overhead_self.py --modes $mode --iters 30 --reps 1 "$@" > $R/gpurun_out/$tag/run_$mode.log 2>&1
  f=$(find $R/gpurun_out/$tag/trace_$mode -name 't_kernel_trace.csv' | head -1)
  python3 $R/tools/trace_gaps.py $f cgm_update_kernel 40 > $R/gpurun_out/$tag/timeline_$mode.txt
  rm -rf $R/gpurun_out/$tag/trace_$mode
done
cd $R
tail -n 3 gpurun_out/$tag/run_*.log
