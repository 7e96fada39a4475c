#!/bin/bash
# p = 5..8 at the config-4 sizes and config 5: bench.py default (atomic pencil kernel where it is ahead) against the deterministic block kernel
out=${1:-gpurun_out/high_degrees.jsonl}
: > $out
B="python bench.py --no-cpu-baseline --no-traffic-pass --sustained-iters 0 --steps 40"
for p in 5 6 7 8; do echo "# p=$p default" >> $out; $B --degree $p 2>/dev/null >> $out; done
echo "# p=5 block 4x4x2" >> $out; $B --degree 5 --cell-block 4 4 2 2>/dev/null >> $out
echo "# p=6 block 4x4x2" >> $out; $B --degree 6 --cell-block 4 4 2 2>/dev/null >> $out
echo "# p=7 block 4x2x2" >> $out; $B --degree 7 --cell-block 4 2 2 2>/dev/null >> $out
echo "# p=8 block 2x2x2" >> $out; $B --degree 8 --cell-block 2 2 2 2>/dev/null >> $out
echo "# config 5 default" >> $out; $B --degree 6 --deform 0.05 2>/dev/null >> $out
echo "# config 5 block 4x4x2" >> $out; $B --degree 6 --deform 0.05 --cell-block 4 4 2 2>/dev/null >> $out
python - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("#"):
        print(line.strip()); continue
    d = json.loads(line)
    print(f"   {d['value'] / 1e9:6.2f} GDoF/s  {d['ms_per_step']:.4f} ms/iter  CG frac {d['roofline_cg']['frac_of_hbm_peak']:.3f}  kernel {d['roofline']['kernel'][:44]} "
          f"{d['roofline']['avg_launch_ms']:.3f} ms frac {d['roofline']['frac']:.3f} (operator-only {d['roofline']['frac_operator_only']:.3f})  variant {d['config']['apply_variant']} fused {d['config']['cg_dot_products_fused']}")
PY
