cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
R=$GRAFT_REPO_ROOT
NODE_TUNE_W4_UF32=1 timeout 600 python -m pytest tests/test_gpu_w4.py -q -m gpu -k "convolution_matches_fp64 or fixture or forward_and_vjp" > gpurun_out/r04i/tests_uf32.log 2>&1
tail -3 gpurun_out/r04i/tests_uf32.log
cd /tmp && export TMPDIR=/tmp
for uf in 0 1; do
rm -rf /tmp/wu_$uf
NODE_TUNE_W4_UF32=$uf rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wu_$uf -- python3 $R/tools/w4_time.py 40 128,256,8 > /tmp/wu_$uf.log 2>&1
KS=$(find /tmp/wu_$uf -name '*kernel_stats.csv' | head -1)
echo "UF32=$uf"; grep "k_w4_gemm" $KS | cut -d, -f1-5
done
cd $R
for uf in 0 1; do
NODE_TUNE_W4_UF32=$uf python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-fresh --no-dropin > gpurun_out/r04i/bench_uf$uf.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04i/bench_uf$uf.json') if l.startswith('{')][-1])
print('UF32=$uf cfg2', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
