#!/bin/bash
# round-5 evidence run (what profiles/r05_bench_cfg*.json, r05_cfg*_steps.txt come from): whole suite with durations, bench lines cfg 2 / 3 / 5, step profiles, latency
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05final
mkdir -p $O gpurun_out/evidence
timeout 1500 python -m pytest tests/ -q -m gpu --durations=15 > $O/gpu_tests.log 2>&1
echo "pytest rc $?" >> $O/gpu_tests.log
tail -24 $O/gpu_tests.log | cut -c1-160
E=gpurun_out/evidence
python bench.py --steps 20 --warmup 5 > $E/bench_cfg2.json 2> $E/bench_cfg2.err
python bench.py --config 3 --steps 20 --warmup 5 --no-pmc > $E/bench_cfg3.json 2> $E/bench_cfg3.err
python bench.py --config 5 --steps 6 --warmup 2 --no-pmc > $E/bench_cfg5.json 2> $E/bench_cfg5.err
STEPS=12 WARM=3 TOP=70 bash tools/profile_bench.sh $E/cfg2 --no-pmc --no-fresh > $E/profile_cfg2.log 2>&1
STEPS=12 WARM=3 TOP=50 bash tools/profile_bench.sh $E/cfg3 --config 3 --no-pmc --no-fresh > $E/profile_cfg3.log 2>&1
for f in bench_cfg2 bench_cfg3 bench_cfg5; do python -c "
import json
try:
    d=json.loads([l for l in open('$E/$f.json') if l.startswith('{')][-1])
    fb=d.get('fresh_batches') or {}
    print('$f', round(d['value'],1), round(d['ms_per_step'],3), 'fresh', round(fb.get('value',0)), 'dropin', round((d.get('dropin') or {}).get('value',0)), 'dead', d['config']['dead_steps_per_step'], 'retries', d['config']['retries'], 'roof', round(d['roofline']['frac'],3), d['roofline']['avg_launch_us'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))
except Exception as e:
    print('$f failed', e)
"; done
grep -E "k_head_loss|TIMED" $E/cfg2_steps.txt | cut -c1-140
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
