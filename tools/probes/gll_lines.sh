set -e
mkdir -p gpurun_out/r4d
python3 bench.py --quadrature gll --no-cpu-baseline > gpurun_out/r4d/bench_gll_116.json 2> gpurun_out/r4d/bench_gll_116.err || tail -5 gpurun_out/r4d/bench_gll_116.err
python3 bench.py --config 4 --no-cpu-baseline --no-traffic-pass --sustained-iters 0 > gpurun_out/r4d/bench_config4_both_quadratures.json 2> gpurun_out/r4d/bench_config4.err || tail -5 gpurun_out/r4d/bench_config4.err
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r4d/bench_gll_116.json').read().strip().splitlines()[-1])
print('GLL 116^3', d['value']/1e9, d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
d=json.loads(open('gpurun_out/r4d/bench_config4_both_quadratures.json').read().strip().splitlines()[-1])
for e in d['sweep']:
    print(e['degree'], e['quadrature'], round(e['value']/1e9,2), round(e['frac_of_hbm_peak'],3), e['kernel'], '| gll', round(e['gll']['value']/1e9,2), round(e['gll']['frac_of_hbm_peak'],3), e['gll']['kernel'])
P
