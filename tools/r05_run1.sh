#!/bin/bash
# round 5, first GPU call: whole -m gpu suite with durations, then the side-stream A/B on the cfg-2 bench loop
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a
mkdir -p $O
timeout 1500 python -m pytest tests/ -q -m gpu --durations=30 > $O/gpu_tests.log 2>&1
echo "pytest rc $?" >> $O/gpu_tests.log
tail -45 $O/gpu_tests.log
for side in 1 0; do
  NODE_TUNE_SIDE=$side python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/bench_side$side.json 2> $O/bench_side$side.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open('$O/bench_side$side.json') if l.startswith('{')][-1])
    print('side $side', round(d['value'], 1), 'ms', round(d['ms_per_step'], 3), 'fresh', round((d.get('fresh_batches') or {}).get('value', 0)),
          'dropin', round((d.get('dropin') or {}).get('value', 0)), 'roof', round(d['roofline']['frac'], 3), d['roofline']['avg_launch_us'],
          'wgrad us', d['roofline'].get('wgrad', {}).get('avg_launch_us'), 'passes ms', d['roofline']['hbm']['all_passes']['ms_per_step'])
except Exception as e:
    print('side $side failed', e)
PY
  tail -3 $O/bench_side$side.err
done
