cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k
timeout 600 python -m pytest tests/test_gpu_deferred.py -q -m gpu > gpurun_out/r04k/tests.log 2>&1; tail -3 gpurun_out/r04k/tests.log
for c in 3 2; do
python bench.py --config $c --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > gpurun_out/r04k/bench_cfg$c.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04k/bench_cfg$c.json') if l.startswith('{')][-1])
f=d['fresh_batches']
print('cfg$c', round(d['value']), d['ms_per_step'], 'settle', d['config']['settle_steps'], 'dead', d['config']['dead_steps_per_step'], 'retries', d['config']['retries'], '| fresh', round(f['value']), 'dead', f['dead_steps_per_step'], 'retries', f['retries'], '| dropin', round(d['dropin']['value']))"
done
python tools/deferred_soak.py --steps 300 --config 3 > gpurun_out/r04k/soak_cfg3.txt 2>&1; tail -2 gpurun_out/r04k/soak_cfg3.txt
