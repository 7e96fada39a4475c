#!/bin/bash
# BASELINE configs 2, 4, 5 with bench.py's defaults (and the deterministic block kernel where it is not the default): one JSON line each
out=${1:-gpurun_out/r2_configs.jsonl}
: > $out
B="python bench.py --no-cpu-baseline --no-traffic-pass --sustained-iters 0 --steps 40"
echo "# config 2: p=4, 54^3" >> $out;            $B --cells 54 54 54 2>/dev/null >> $out
for p in 1 2 3 5 6 7 8; do echo "# config 4: p=$p (default brick order of bench.py)" >> $out; $B --degree $p 2>/dev/null >> $out; done
echo "# config 4, deterministic block kernel: p=2 8x8x4 bricks" >> $out; $B --degree 2 --cell-block 8 8 4 2>/dev/null >> $out
echo "# p=3 8x4x4 bricks" >> $out; $B --degree 3 --cell-block 8 4 4 2>/dev/null >> $out
echo "# p=5 4x4x2 bricks" >> $out; $B --degree 5 --cell-block 4 4 2 2>/dev/null >> $out
echo "# p=7 4x2x2 bricks" >> $out; $B --degree 7 --cell-block 4 2 2 2>/dev/null >> $out
echo "# config 5: p=6, 61^3, deformed (default = atomic pencil kernel on 8x8x8 bricks)" >> $out; $B --degree 6 --deform 0.05 2>/dev/null >> $out
echo "# config 5, deterministic block kernel on 4x4x2 bricks" >> $out; $B --degree 6 --deform 0.05 --cell-block 4 4 2 2>/dev/null >> $out
python - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("#"):
        print(line.strip()); continue
    d = json.loads(line)
    print(f"   {d['value'] / 1e9:6.2f} GDoF/s  {d['ms_per_step']:.4f} ms/iter  CG frac {d['roofline_cg']['frac_of_hbm_peak']:.3f}  kernel {d['roofline']['kernel'][:44]} "
          f"{d['roofline']['avg_launch_ms']:.3f} ms frac {d['roofline']['frac']:.3f} (operator-only {d['roofline']['frac_operator_only']:.3f})  variant {d['config']['apply_variant']} fused {d['config']['cg_dot_products_fused']}")
PY
