cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stem.py -q -m gpu -x > gpurun_out/r04f/stem_tests.log 2>&1
tail -3 gpurun_out/r04f/stem_tests.log
python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > gpurun_out/r04f/bench_cfg2.json 2> gpurun_out/r04f/bench_cfg2.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04f/bench_cfg2.json') if l.startswith('{')][-1])
print('cfg2', d['value'], d['ms_per_step'], 'fresh', d['fresh_batches']['value'], 'dropin', d['dropin']['value'], 'dead', d['config']['dead_steps_per_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python3 $R/tools/stem_time.py --run > /tmp/st.log 2>&1 || tail -20 /tmp/st.log
cd $R
python3 tools/stem_time.py --report /tmp/st > gpurun_out/r04f/stem_time.txt
tail -3 gpurun_out/r04f/stem_time.txt; grep reduce gpurun_out/r04f/stem_time.txt
