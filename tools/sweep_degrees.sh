#!/bin/bash
# BASELINE config 4: p = 1..8 at ~5e7 DoFs (bench.py sizes), both quadratures, library defaults
for q in gauss gll; do for p in 1 2 3 4 5 6 7 8; do
  echo -n "p=$p $q : "
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-traffic-pass --degree $p --quadrature $q --steps 30 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,2), 'GDoF/s', round(d['ms_per_step'],3), 'ms cg_frac', round(d['roofline_cg']['frac_of_hbm_peak'],3), 'kernel ms', round(d['roofline']['avg_launch_ms'],3), 'op frac', round(d['roofline']['frac'],3), 'variant', d['config']['apply_variant'], 'dofs', d['config']['dofs_per_gpu'])"
done; done
