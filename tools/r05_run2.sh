#!/bin/bash
# round 5, second GPU call: the tests that failed / are new, the drop-in time budget, the limiter counters of k_w4_gemm64b
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
timeout 1500 python -m pytest tests/ -q -m gpu --durations=25 > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log
tail -40 $O/tests.log
NODE_HIP_DIAG=1 timeout 300 python -m pytest tests/test_diag_w4.py -q -m diag > $O/diag_tests.log 2>&1; tail -3 $O/diag_tests.log
timeout 300 python tools/dropin_time.py 40 > $O/dropin_time.txt 2>&1; cat $O/dropin_time.txt
timeout 300 python tools/latency_bs1.py > $O/latency_bs1.txt 2>&1; tail -12 $O/latency_bs1.txt
NODE_TUNE_TINY=0 timeout 300 python tools/latency_bs1.py > $O/latency_bs1_tiny0.txt 2>&1; tail -6 $O/latency_bs1_tiny0.txt
timeout 900 bash tools/pmc_w4_limiter.sh $O/pmc_w4_limiter.txt > $O/pmc.log 2>&1; tail -70 $O/pmc.log
