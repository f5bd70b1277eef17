#!/bin/bash
# vectorised component loads / stores of the passes (W4S_VEC): parity, then A/B against the build without them, then the passes' limiter counters
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_w4.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_stem.py tests/test_gpu_head.py tests/test_gpu_train.py -q -m gpu -x > $O/tests.log 2>&1; tail -4 $O/tests.log | cut -c1-160
L=$GRAFT_REPO_ROOT/neural-ode-features_amd/csrc
for r in 1 2 3; do
  for v in libnode_hip.so libnode_hip_vec0.so; do
    NODE_HIP_LIB_AB=$L/$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-fresh --no-dropin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-24s %9.1f images/s  %.3f ms/step  gemm %.2f us  passes %.3f ms/step  frac %.3f' % ('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['hbm']['all_passes']['ms_per_step'], d['roofline']['hbm']['all_passes']['frac']))" | tee -a $O/ab_vec.txt
  done
done
timeout 900 bash tools/pmc_pass_limiter.sh $O/pmc_pass_limiter.txt > $O/pmc.log 2>&1; tail -5 $O/pmc.log | cut -c1-150
