#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/evidence
mkdir -p $O
STEPS=12 WARM=3 TOP=70 bash tools/profile_bench.sh $O/cfg2 --no-pmc --no-fresh > $O/profile_cfg2.log 2>&1; head -40 $O/cfg2_steps.txt | cut -c1-125
STEPS=12 WARM=3 TOP=50 bash tools/profile_bench.sh $O/cfg3 --config 3 --no-pmc --no-fresh > $O/profile_cfg3.log 2>&1; head -12 $O/cfg3_steps.txt | cut -c1-125
timeout 600 python -m pytest tests/test_gpu_generic.py -q -m gpu --durations=5 > $O/generic_tests.log 2>&1; tail -9 $O/generic_tests.log | cut -c1-150
