#!/bin/bash
# pencil kernel on lexicographic vs brick-ordered (parity-class, brick-major numbering) meshes, all degrees
for p in 1 2 3 5 6 7 8; do
  for blk in "0 0 0" "4 4 4" "8 8 8"; do
    echo -n "p=$p block=$blk : "
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-traffic-pass --degree $p --cell-block $blk --steps 30 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,2), 'GDoF/s', round(d['ms_per_step'],3), 'ms cg_frac', round(d['roofline_cg']['frac_of_hbm_peak'],3), 'kernel ms', round(d['roofline']['avg_launch_ms'],3), 'variant', d['config']['apply_variant'])"
  done
done
