cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04m
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stem.py -q -m gpu -s -k whole_stem > gpurun_out/r04m/stem_tests.log 2>&1
grep "stem (\|passed\|failed\|Error\|error" gpurun_out/r04m/stem_tests.log | tail -30
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python3 $R/tools/stem_time.py --run > /tmp/st.log 2>&1 || tail -20 /tmp/st.log
cd $R
python3 - <<'PY' > gpurun_out/r04m/stem_time.txt
import csv, glob
f = glob.glob('/tmp/st/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
# last iteration = from the last k_stem_prep on
idx = max(i for i, r in enumerate(rows) if 'k_stem_prep' in r[2])
last = rows[idx:]
tot = 0
for s, e, n in last:
    short = n.split('(')[0].replace('void ', '').replace('node::', '').replace('anonymous namespace)::','')
    print('%-40s %8.1f us' % (short[:40], (e - s) / 1e3)); tot += (e - s) / 1e3
print('total %.1f us over %d launches; span %.1f us' % (tot, len(last), (last[-1][1] - last[0][0]) / 1e3))
PY
cat gpurun_out/r04m/stem_time.txt
python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-fresh --no-dropin > gpurun_out/r04m/bench.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04m/bench.json') if l.startswith('{')][-1])
print('cfg2', round(d['value']), d['ms_per_step'])"
