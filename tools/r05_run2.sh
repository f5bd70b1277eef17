#!/bin/bash
# round 5, second GPU call: the tests that failed / are new, the drop-in time budget, the limiter counters of k_w4_gemm64b
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_head.py tests/test_gpu_generic.py tests/test_gpu_train.py tests/test_gpu_stem.py tests/test_gpu_golden.py "tests/test_gpu_w4.py::test_w4_wgrad128_matches_oracle" tests/test_gpu_parity.py -q -m gpu --durations=12 > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log
tail -40 $O/tests.log
NODE_HIP_DIAG=1 timeout 300 python -m pytest tests/test_diag_w4.py -q -m diag > $O/diag_tests.log 2>&1; tail -3 $O/diag_tests.log
timeout 300 python tools/dropin_time.py 40 > $O/dropin_time.txt 2>&1; cat $O/dropin_time.txt
timeout 900 bash tools/pmc_w4_limiter.sh $O/pmc_w4_limiter.txt > $O/pmc.log 2>&1; tail -70 $O/pmc.log
