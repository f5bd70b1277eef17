cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
python bench.py --config 3 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > gpurun_out/r04j/bench_cfg3.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04j/bench_cfg3.json') if l.startswith('{')][-1])
print('cfg3', round(d['value']), d['ms_per_step'], 'fresh', d['fresh_batches'], 'dropin', d['dropin']['value'], 'dead', d['config']['dead_steps_per_step'], d['config']['per_rank'])"
STEPS=12 WARM=3 TOP=45 bash tools/profile_bench.sh gpurun_out/r04j/cfg3 --config 3 --no-pmc --no-fresh > gpurun_out/r04j/profile.log 2>&1
head -52 gpurun_out/r04j/cfg3_steps.txt
