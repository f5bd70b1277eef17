#!/bin/bash
# (diagnostics library: export NODE_HIP_DIAG=1 first) A/B of the half-height two-waves-per-SIMD component GEMM (NODE_TUNE_W4_HALF) on the cfg-2 / cfg-3 bench loops + its bit-identity test
cd ${GRAFT_REPO_ROOT:-.}
export NODE_HIP_DIAG=1     # k_w4_gemm32b lives in libnode_hip_diag.so (build.py --diag)
O=gpurun_out/w4_half
mkdir -p $O
timeout 600 python -m pytest "tests/test_gpu_w4.py::test_w4_gemm_work_assignments_are_bit_identical" tests/test_gpu_parity.py -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for half in 1 0 1 0; do
  NODE_TUNE_W4_HALF=$half python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-fresh > $O/bench_half$half.json 2> $O/bench_half$half.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open('$O/bench_half$half.json') if l.startswith('{')][-1])
    print('half $half', round(d['value'], 1), 'ms', round(d['ms_per_step'], 3), 'dropin', round((d.get('dropin') or {}).get('value', 0)), 'roof', round(d['roofline']['frac'], 3),
          'gemm us', round(d['roofline']['avg_launch_us'], 2), 'wgrad us', round(d['roofline'].get('wgrad', {}).get('avg_launch_us', 0), 1), 'passes ms', round(d['roofline']['hbm']['all_passes']['ms_per_step'], 3))
except Exception as e:
    print('half $half failed', e)
PY
done
for half in 1 0; do
  NODE_TUNE_W4_HALF=$half python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/bench_cfg3_half$half.json 2> $O/bench_cfg3_half$half.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open('$O/bench_cfg3_half$half.json') if l.startswith('{')][-1])
    print('cfg3 half $half', round(d['value'], 1), 'ms', round(d['ms_per_step'], 3), 'fresh', round((d.get('fresh_batches') or {}).get('value', 0)), 'dropin', round((d.get('dropin') or {}).get('value', 0)),
          'dead', d['config']['dead_steps_per_step'], 'retries', d['config']['retries'])
except Exception as e:
    print('cfg3 half $half failed', e)
PY
done
