#!/bin/bash
# A/B of two builds of the library on ONE box: bench.py per (library, degree, brick); usage: tools/ab_libs.sh "<libA> <libB>" "<degree>:<bx> <by> <bz>" ...
# (each case is its own process: the library is chosen through BP5_LIB)
libs=$1; shift
for case in "$@"; do
  p=${case%%:*}; blk=${case#*:}
  for lib in $libs; do
    BP5_LIB=$PWD/deal-and-ceed-on-gpu_amd/$lib python3 bench.py --degree $p --steps 30 --no-cpu-baseline --no-traffic-pass --sustained-iters 0 --apply-variant 56 --cell-block $blk 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('p=$p $lib bricks $blk: %.2f GDoF/s, %.3f ms/iter, CG frac %.3f, %s %.3f ms' % (d['value']/1e9, d['ms_per_step'], d['roofline_cg']['frac_of_hbm_peak'], d['roofline']['kernel'], d['roofline']['avg_launch_ms']))
    elif 'rror' in l: print(l.strip()[:200])
"
  done
done
