#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_generic.py tests/test_gpu_forced_tiles.py tests/test_gpu_solve.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_train.py "tests/test_gpu_w4.py::test_w4_solve_matches_f2_and_oracle_tolerance" tests/test_gpu_round2.py -k "not full_size" -q -m gpu --durations=10 > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log
tail -30 $O/tests.log | cut -c1-200
timeout 300 python tools/latency_bs1.py > $O/latency_bs1.txt 2>&1; tail -3 $O/latency_bs1.txt
timeout 300 python tools/latency_bs1.py --shape 1,64,8,8 > $O/latency_bs1_c64.txt 2>&1; tail -3 $O/latency_bs1_c64.txt
timeout 300 python tools/latency_bs1.py --shape 1,256,16,16 > $O/latency_bs1_16.txt 2>&1; tail -3 $O/latency_bs1_16.txt
NODE_TUNE_TINY=0 timeout 300 python tools/latency_bs1.py --shape 1,256,16,16 > $O/latency_bs1_16_tiny0.txt 2>&1; tail -3 $O/latency_bs1_16_tiny0.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -- python3 $GRAFT_REPO_ROOT/tools/latency_bs1.py --tols 1e-3 --iters 20 > /tmp/lt.log 2>&1
cd $GRAFT_REPO_ROOT
KS=$(find /tmp/lt -name '*kernel_stats.csv' | head -1); cp $KS $O/latency_kernel_stats.csv; head -12 $O/latency_kernel_stats.csv | cut -c1-160
