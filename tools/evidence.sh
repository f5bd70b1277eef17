#!/bin/bash
# Per-round evidence (round 6): the default bench line (cfg 2 with PMC traffic, CPU baseline, cfgs 3 / 5 as side runs with their
# rooflines), rocprofv3 step profiles of cfgs 2 / 3 / 5, the fp16-pair A/B on this box, counters + in-kernel timeline of the
# dominant kernel, the pure-traffic twin of its byte counts (tools/hbm_stream.hip), stem launch times, the bs = 1 census.  Run from the repo root on the GPU box; writes gpurun_out/evidence/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/evidence
mkdir -p $O
R=$GRAFT_REPO_ROOT
python bench.py > $O/bench_full.json 2> $O/bench_full.err
for f in 1 0; do
  NODE_TUNE_W4_F16=$f python bench.py --steps 30 --warmup 8 --no-roofline --no-cpu-baseline --no-other-configs --no-latency 2>/dev/null | tail -1 > $O/bench_cfg2_f16_$f.json
done
STEPS=12 WARM=3 TOP=70 bash tools/profile_bench.sh $O/cfg2 --no-pmc --no-fresh > $O/profile_cfg2.log 2>&1
STEPS=12 WARM=3 TOP=50 bash tools/profile_bench.sh $O/cfg3 --config 3 --no-pmc --no-fresh > $O/profile_cfg3.log 2>&1
STEPS=4 WARM=2 TOP=40 bash tools/profile_bench.sh $O/cfg5 --config 5 --no-pmc --no-fresh > $O/profile_cfg5.log 2>&1
bash tools/pmc_w4h.sh $O/pmc_w4h.txt > /dev/null 2>&1
bash tools/pmc_w4h.sh $O/pmc_w4h128_cfg5.txt k_w4_gemm128h 64,1024,16 > /dev/null 2>&1
python tools/w4_stamps.py > $O/w4h_stamps.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/hbm_stream.hip -o tools/hbm_stream 2>/dev/null && tools/hbm_stream > $O/hbm_stream.txt 2>&1
python tools/f16_check.py > $O/f16_check.txt 2>&1
python tools/f16_adjoint_ab.py > $O/f16_adjoint_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st
rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python3 $R/tools/stem_time.py --run > /tmp/st.log 2>&1
cd $R
python3 tools/stem_time.py --report /tmp/st > $O/stem_time.txt
python tools/census_bs1.py > $O/census_bs1.txt 2>&1
python tools/census_bs1.py --graphs >> $O/census_bs1.txt 2>&1
python -c "
import json
d=json.loads([l for l in open('$O/bench_full.json') if l.startswith('{')][-1])
print('cfg2', round(d['value']), round(d['ms_per_step'],3), 'dropin', round(d['dropin']['value']), 'roof', round(d['roofline']['frac'],3), d['roofline']['avg_launch_us'], 'traffic', d['roofline']['traffic'], 'cpu', d['cpu_baseline']['value'])
for k, v in d['other_configs'].items(): print('cfg', k, round(v['value'],1), v['ms_per_step'], 'dead', v['dead_steps_per_step'], 'roof', (v.get('roofline') or {}).get('frac'))
"
for f in 1 0; do python -c "import json; d=json.loads(open('$O/bench_cfg2_f16_$f.json').read()); print('F16=$f', round(d['value']), d['ms_per_step'])"; done
tail -3 $O/stem_time.txt; grep filters $O/census_bs1.txt
