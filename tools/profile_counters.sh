#!/bin/bash
# Extra hardware counters of the dominant kernel of the driver's bench command, one rocprofv3 --pmc pass per counter group (--kernel-trace only):
#   bash tools/profile_counters.sh <tag> "TCC_HIT_sum TCC_MISS_sum" "LDSBankConflict" ...
# BENCH_ARGS="--config 4 --degree 8 ..." profiles another workload of bench.py (default: the driver's)
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/$tag/g$i -o p --output-format csv -- python3 $R/bench.py --gpus 1 $BENCH_ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-traffic-pass --no-mesh-116 --sustained-iters 0 > $R/gpurun_out/$tag/g$i.log 2>&1 || echo "group $i failed: $grp"
done
cd $R
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
out = open(f"gpurun_out/{tag}/counters_per_kernel.csv", "w")
out.write("kernel,counter,launches,avg_value\n")
for f in sorted(glob.glob(f"gpurun_out/{tag}/g*/**/p_counter_collection.csv", recursive=True)):
    per = {}
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"][:60], row["Counter_Name"])
        per.setdefault(k, []).append(float(row["Counter_Value"]))
    for (k, c), v in sorted(per.items()):
        if "apply_" in k or "combine_runs" in k or "cgm_update" in k:
            out.write(f"\"{k}\",{c},{len(v)},{sum(v) / len(v):.4g}\n")
out.close()
print(open(f"gpurun_out/{tag}/counters_per_kernel.csv").read())
PY
rm -rf $R/gpurun_out/$tag/g*/
