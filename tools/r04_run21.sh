cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04t
timeout 900 python -m pytest tests/test_gpu_w4.py tests/test_gpu_stem.py -q -m gpu -k "wgrad or forward_and_vjp or fixture or whole_stem" > gpurun_out/r04t/tests.log 2>&1
tail -2 gpurun_out/r04t/tests.log
for sv in 0 1 0 1; do
NODE_TUNE_W4_SHAREV=$sv python bench.py --steps 30 --warmup 5 --no-pmc --no-cpu-baseline --no-fresh --no-dropin > gpurun_out/r04t/bench_sv$sv.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04t/bench_sv$sv.json') if l.startswith('{')][-1])
print('SHAREV=$sv cfg2', round(d['value']), d['ms_per_step'], 'gemm', round(d['roofline']['avg_launch_us'],2), 'wgrad', round(d['roofline']['wgrad']['avg_launch_us'],2))"
done
