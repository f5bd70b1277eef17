cd $GRAFT_REPO_ROOT
E=gpurun_out/evidence; mkdir -p $E
python bench.py --steps 20 --warmup 5 > $E/bench_cfg2.json 2> $E/bench_cfg2.err
python bench.py --config 3 --steps 20 --warmup 5 --no-pmc > $E/bench_cfg3.json 2> $E/bench_cfg3.err
python bench.py --config 5 --steps 6 --warmup 2 --no-pmc > $E/bench_cfg5.json 2> $E/bench_cfg5.err
for f in bench_cfg2 bench_cfg3 bench_cfg5; do python -c "
import json
d=json.loads([l for l in open('$E/$f.json') if l.startswith('{')][-1])
print('$f', round(d['value'],1), round(d['ms_per_step'],3), 'fresh', round(d['fresh_batches']['value']), 'dropin', round(d['dropin']['value']), 'dead', d['config']['dead_steps_per_step'], 'roof', round(d['roofline']['frac'],3), 'cpu', round(d['cpu_baseline']['value'],2), 'lat', [(r['tol'], r['nfe'], round(r['us_per_evaluation'],1)) for r in d['latency_bs1'].get('solves', [])], d['latency_bs1'].get('one_launch_per_solve'), d['latency_bs1'].get('error'))
"; done
