mkdir -p gpurun_out/r4o
python3 bench.py --degree 2 --cells 184 184 184 --steps 30 --no-cpu-baseline --no-traffic-pass --sustained-iters 0 > gpurun_out/r4o/p2_wavepack.json 2> gpurun_out/r4o/p2_wavepack.err
python3 bench.py --degree 2 --cells 184 184 184 --steps 30 --quadrature gll --no-cpu-baseline --no-traffic-pass --sustained-iters 0 > gpurun_out/r4o/p2_wavepack_gll.json 2> gpurun_out/r4o/p2_wavepack_gll.err
BP5_LIB=$PWD/libbp5_alt.so python3 bench.py --degree 2 --cells 184 184 184 --steps 30 --no-cpu-baseline --no-traffic-pass --sustained-iters 0 > gpurun_out/r4o/p2_alt_4wg_no_pingpong.json 2> gpurun_out/r4o/p2_alt.err
for f in p2_wavepack p2_wavepack_gll p2_alt_4wg_no_pingpong; do python3 -c "
import json;d=json.loads(open('gpurun_out/r4o/$f.json').read().strip().splitlines()[-1]);print('$f',round(d['value']/1e9,3),'GDoF/s',round(d['ms_per_step'],4),'ms',d['roofline']['kernel'],round(d['roofline']['avg_launch_ms'],4), round(d['roofline_cg']['frac_of_hbm_peak'],3))"; done
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r4o/gputests.log 2>&1; tail -4 gpurun_out/r4o/gputests.log
