cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04n
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_w4.py -q -m gpu -k "convolution_matches_fp64 or fixture or forward_and_vjp or split" > gpurun_out/r04n/tests.log 2>&1
tail -3 gpurun_out/r04n/tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/we
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/we -- python3 $R/tools/w4_time.py 40 128,256,8 > /tmp/we.log 2>&1
KS=$(find /tmp/we -name '*kernel_stats.csv' | head -1)
python3 -c "
import csv,sys
for r in csv.DictReader(open('$KS')):
    if 'gemm' in r['Name']: print(r['Name'][:40], float(r['AverageNs'])/1e3, 'us min', float(r['MinNs'])/1e3)"
cd $R
python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-fresh --no-dropin > gpurun_out/r04n/bench.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04n/bench.json') if l.startswith('{')][-1])
print('cfg2', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_us'])"
