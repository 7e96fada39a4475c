set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=traffic_ablation2
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
export BP5_LATTICE_INDICES=0
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/$tag/$c -o p --output-format csv -- python3 $R/tools/bench_apply.py --degree 4 --cells 116 116 116 \
     --cell-block 4 4 4 --numbering 1 --block-order 1 --variants 56 61 60 --rounds 3 --reps 3 --overwrite > $R/gpurun_out/$tag/$c.log 2>&1 || echo "pass $c failed"
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
for (k, c), v in sorted(per.items()):
    avg = sum(v) / len(v)
    print(f"{k}  {c}  {len(v)}  {avg:.0f}  -> {avg * 1024 * (2 if c == 'FETCH_SIZE' else 1) / 1e9:.3f} GB")
PY
tail -5 gpurun_out/$tag/FETCH_SIZE.log | cut -c1-200
rm -rf $R/gpurun_out/$tag/FETCH_SIZE $R/gpurun_out/$tag/WRITE_SIZE
