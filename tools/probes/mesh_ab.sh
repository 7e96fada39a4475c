mkdir -p gpurun_out/r4h
python3 bench.py --no-cpu-baseline --no-traffic-pass > gpurun_out/r4h/bench_116.json 2>gpurun_out/r4h/bench_116.err
python3 bench.py --cells 128 96 128 --no-cpu-baseline --no-traffic-pass > gpurun_out/r4h/bench_128x96x128.json 2>gpurun_out/r4h/bench_128.err
python3 bench.py --cells 128 128 128 --no-cpu-baseline --no-traffic-pass > gpurun_out/r4h/bench_128cubed.json 2>gpurun_out/r4h/bench_128c.err
python3 tools/halo_overhead_self.py --cell-block 4 4 2 --modes auto,none --iters 100 > gpurun_out/r4h/halo_116.txt 2>&1
python3 tools/halo_overhead_self.py --cell-block 4 4 2 --modes auto,none --iters 100 --cells 128 96 128 > gpurun_out/r4h/halo_128x96x128.txt 2>&1
python3 bench.py --degree 1 --cells 367 367 367 --steps 50 --no-cpu-baseline --no-traffic-pass --sustained-iters 0 > gpurun_out/r4h/p1_gauss.json 2>/dev/null
python3 bench.py --degree 1 --cells 367 367 367 --steps 50 --quadrature gll --no-cpu-baseline --no-traffic-pass --sustained-iters 0 > gpurun_out/r4h/p1_gll.json 2>/dev/null
python3 - <<'P'
import json
for f in ('bench_116','bench_128x96x128','bench_128cubed','p1_gauss','p1_gll'):
    try:
        d=json.loads(open(f'gpurun_out/r4h/{f}.json').read().strip().splitlines()[-1])
        print(f, round(d['value']/1e9,3), round(d['ms_per_step'],4), d['roofline']['avg_launch_ms'], d.get('sustained',{}) and round(d['sustained']['value']/1e9,3), d['roofline_cg']['frac_of_hbm_peak'])
    except Exception as e: print(f,'failed',e)
P
tail -4 gpurun_out/r4h/halo_116.txt gpurun_out/r4h/halo_128x96x128.txt
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r4h/gputests.log 2>&1; tail -5 gpurun_out/r4h/gputests.log
