#!/bin/bash
# rocprofv3 passes over the DRIVER's bench command (python3 bench.py --gpus 1 --steps 20 --warmup 5): kernel trace + stats,
# then separate PMC passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only; outputs under gpurun_out/$1
# (the CPU-baseline and sustained legs are switched off under the profiler: they do not touch the timed region)
set -e
tag=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag/stats -o s --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --no-mesh-116 --sustained-iters 0 > $R/gpurun_out/$tag/bench_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/$tag/pmc_$c -o p --output-format csv -- python3 $R/bench.py --gpus 1 --steps 6 --warmup 1 --no-cpu-baseline --no-traffic-pass --no-mesh-116 --sustained-iters 0 > $R/gpurun_out/$tag/bench_pmc_$c.log 2>&1
done
cd $R
python3 - "$tag" <<'PY'
import csv, glob, json, os, sys
tag = sys.argv[1]
base = f"gpurun_out/{tag}"
out = {}
st = glob.glob(f"{base}/stats/**/s_kernel_stats.csv", recursive=True)
if st:
    os.system(f"cp {st[0]} {base}/kernel_stats.csv")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{base}/pmc_{c}/**/p_counter_collection.csv", recursive=True)
    if not f:
        continue
    per = {}
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"][:70]
        per.setdefault(k, []).append(float(row["Counter_Value"]))
    with open(f"{base}/pmc_{c}_per_kernel.csv", "w") as o:
        o.write("kernel,launches,avg_counter_value\n")
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            o.write(f"\"{k}\",{len(v)},{sum(v) / len(v):.1f}\n")
    out[c] = {k: sum(v) / len(v) for k, v in per.items()}
json.dump(out, open(f"{base}/pmc_summary.json", "w"), indent=1)
for line in open(f"{base}/bench_stats.log"):
    if line.startswith("{"):
        open(f"{base}/bench_under_profiler.json", "w").write(line)
PY
rm -rf $R/gpurun_out/$tag/stats $R/gpurun_out/$tag/pmc_FETCH_SIZE $R/gpurun_out/$tag/pmc_WRITE_SIZE
ls -la $R/gpurun_out/$tag
