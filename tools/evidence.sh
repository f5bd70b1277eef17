#!/bin/bash
# Per-round evidence: bench lines of cfg 2 / 3 / 5 (PMC traffic + CPU baseline at cfg 2), rocprofv3 step profiles, stem launch
# times, deferred soak at tol 1e-5, bs = 1 latency.  Run from the repo root on the GPU box; writes gpurun_out/evidence/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/evidence
mkdir -p $O
R=$GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --config 3 --steps 20 --warmup 5 --no-pmc > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config 5 --steps 6 --warmup 2 --no-pmc > $O/bench_cfg5.json 2> $O/bench_cfg5.err
STEPS=12 WARM=3 TOP=70 bash tools/profile_bench.sh $O/cfg2 --no-pmc --no-fresh > $O/profile_cfg2.log 2>&1
STEPS=12 WARM=3 TOP=50 bash tools/profile_bench.sh $O/cfg3 --config 3 --no-pmc --no-fresh > $O/profile_cfg3.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st
rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python3 $R/tools/stem_time.py --run > /tmp/st.log 2>&1
cd $R
python3 tools/stem_time.py --report /tmp/st > $O/stem_time.txt
python tools/deferred_soak.py --steps 300 --config 3 > $O/deferred_soak_cfg3.txt 2>&1
python tools/latency_bs1.py > $O/latency_bs1.txt 2>&1
for f in bench_cfg2 bench_cfg3 bench_cfg5; do python -c "
import json
try:
    d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1])
    fb=d.get('fresh_batches') or {}
    print('$f', round(d['value'],1), round(d['ms_per_step'],3), 'fresh', round(fb.get('value',0)), 'dropin', round((d.get('dropin') or {}).get('value',0)), 'dead', d['config']['dead_steps_per_step'], 'retries', d['config']['retries'], 'roof', round(d['roofline']['frac'],3), d['roofline']['avg_launch_us'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))
except Exception as e:
    print('$f failed', e)
"; done
tail -3 $O/stem_time.txt; tail -2 $O/deferred_soak_cfg3.txt; tail -4 $O/latency_bs1.txt
cd /tmp && rm -rf /tmp/st8
NODE_TUNE_STEM_GNCB=8 rocprofv3 --kernel-trace --output-format csv -d /tmp/st8 -- python3 $R/tools/stem_time.py --run > /tmp/st8.log 2>&1
cd $R
python3 tools/stem_time.py --report /tmp/st8 > $O/stem_time_gncb8.txt
grep "gn_\|total" $O/stem_time.txt | tr '\n' ' '; echo; grep "gn_\|total" $O/stem_time_gncb8.txt | tr '\n' ' '; echo
