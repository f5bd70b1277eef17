#!/bin/bash
# round 5 evidence: the whole -m gpu suite (durations), the diag suite, then tools/evidence.sh (bench lines, step profiles, soak, latency)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05final
mkdir -p $O
timeout 1500 python -m pytest tests/ -q -m gpu --durations=25 > $O/gpu_tests.log 2>&1
echo "pytest rc $?" >> $O/gpu_tests.log
tail -40 $O/gpu_tests.log | cut -c1-180
NODE_HIP_DIAG=1 timeout 300 python -m pytest tests/test_diag_w4.py -q -m diag > $O/diag_tests.log 2>&1; tail -2 $O/diag_tests.log
bash tools/evidence.sh > $O/evidence.log 2>&1
tail -30 $O/evidence.log | cut -c1-220
timeout 300 python tools/dropin_time.py 40 > gpurun_out/evidence/dropin_time.txt 2>&1; cat gpurun_out/evidence/dropin_time.txt
