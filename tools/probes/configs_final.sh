mkdir -p gpurun_out/r4z
python3 bench.py --config 2 --no-cpu-baseline --no-traffic-pass > gpurun_out/r4z/bench_config2.json 2> gpurun_out/r4z/c2.err
python3 bench.py --config 5 --no-cpu-baseline --no-traffic-pass > gpurun_out/r4z/bench_config5.json 2> gpurun_out/r4z/c5.err
python3 bench.py --config 4 --no-cpu-baseline --no-traffic-pass --sustained-iters 0 > gpurun_out/r4z/bench_config4.json 2> gpurun_out/r4z/c4.err
python3 bench.py --config 1 --no-traffic-pass > gpurun_out/r4z/bench_config1.json 2> gpurun_out/r4z/c1.err
python3 bench.py --operator helmholtz --degree 3 --cells 122 122 122 --no-cpu-baseline --no-traffic-pass > gpurun_out/r4z/bench_helmholtz_p3.json 2> gpurun_out/r4z/h.err
python3 - <<'P'
import json
for f in ('bench_config2','bench_config5','bench_config1','bench_helmholtz_p3'):
    try:
        d=json.loads(open(f'gpurun_out/r4z/{f}.json').read().strip().splitlines()[-1])
        print(f, round(d['value']/1e9,3),'GDoF/s', round(d['ms_per_step'],4),'ms frac', round(d['roofline_cg']['frac_of_hbm_peak'],3), 'sustained', d.get('sustained') and round(d['sustained']['value']/1e9,2), d['roofline']['kernel'])
    except Exception as e: print(f,'failed',e)
d=json.loads(open('gpurun_out/r4z/bench_config4.json').read().strip().splitlines()[-1])
for e in d['sweep']:
    print(e['degree'], e['quadrature'], round(e['value']/1e9,2), round(e['frac_of_hbm_peak'],3), e['kernel'], '| gll', round(e['gll']['value']/1e9,2), round(e['gll']['frac_of_hbm_peak'],3))
P
