cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_w4.py -q -m gpu -k "convolution_matches_fp64 or fixture or forward_and_vjp" > gpurun_out/r04s/tests.log 2>&1
tail -2 gpurun_out/r04s/tests.log
cd /tmp && export TMPDIR=/tmp
for rot in 0 1; do
rm -rf /tmp/wr_$rot
NODE_TUNE_W4_SHAREV=$rot rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wr_$rot -- python3 $R/tools/w4_time.py 40 128,256,8 > /tmp/wr_$rot.log 2>&1
KS=$(find /tmp/wr_$rot -name '*kernel_stats.csv' | head -1)
python3 -c "
import csv
for r in csv.DictReader(open('$KS')):
    if 'gemm' in r['Name']: print('SHAREV=$rot', r['Name'][:30], float(r['AverageNs'])/1e3, 'us min', float(r['MinNs'])/1e3)"
done
cd $R
for rot in 0 1 0 1; do
NODE_TUNE_W4_SHAREV=$rot python bench.py --steps 30 --warmup 5 --no-pmc --no-cpu-baseline --no-fresh --no-dropin > gpurun_out/r04s/bench_rot$rot.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04s/bench_rot$rot.json') if l.startswith('{')][-1])
print('SHAREV=$rot cfg2', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
