#!/bin/bash
# Where the Helmholtz block kernel's traffic goes: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each, --kernel-trace only) of the
# p = 3 kernel on 122^3 cells (8x4x4 bricks) and of its timing-only ablations (libbp5_timing.so: 91 no write-out / 93 no plane loads / 95 no gather), unfused.
#   make -C deal-and-ceed-on-gpu_amd/csrc timing && bash tools/profile_traffic_ablation_helmholtz.sh <tag>
set -e
tag=${1:-traffic_ablation_helmholtz}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
export BP5_LIB=$R/deal-and-ceed-on-gpu_amd/libbp5_timing.so
ARGS="--operator helmholtz --degree 3 --cells 122 122 122 --cell-block 8 4 4 --numbering 1 --block-order 1 --variants 56 91 93 95 --overwrite"
python3 $R/tools/bench_apply.py $ARGS --rounds 3 --reps 5 > $R/gpurun_out/$tag/timing.txt 2>&1 || echo "timing run failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/$tag/$c -o p --output-format csv -- python3 $R/tools/bench_apply.py $ARGS --rounds 1 --reps 3 > $R/gpurun_out/$tag/$c.log 2>&1 || echo "pass $c failed"
done
cd $R
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
per = {}
for f in sorted(glob.glob(f"gpurun_out/{tag}/*/**/p_counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "apply_block_kernel" in row["Kernel_Name"] or "combine_runs" in row["Kernel_Name"]:
            per.setdefault((row["Kernel_Name"].split("(")[0][-48:], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
with open(f"gpurun_out/{tag}/traffic_per_kernel.txt", "w") as out:
    out.write("kernel  counter  launches  avg KiB (FETCH_SIZE: x2 for bytes on gfx950)\n")
    for (k, c), v in sorted(per.items()):
        avg = sum(v) / len(v)
        gb = avg * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9
        out.write(f"{k}  {c}  {len(v)}  {avg:.0f}  -> {gb:.3f} GB\n")
print(open(f"gpurun_out/{tag}/traffic_per_kernel.txt").read())
PY
tail -6 gpurun_out/$tag/timing.txt
rm -rf $R/gpurun_out/$tag/FETCH_SIZE $R/gpurun_out/$tag/WRITE_SIZE
