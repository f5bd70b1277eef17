#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_generic.py tests/test_gpu_forced_tiles.py tests/test_gpu_solve.py tests/test_gpu_parity.py tests/test_gpu_round2.py -k "not full_size and not 1e-05" -q -m gpu --durations=6 > $O/tests.log 2>&1
echo "pytest rc $?" >> $O/tests.log
tail -14 $O/tests.log | cut -c1-200
NODE_HIP_DIAG=1 timeout 300 python -m pytest tests/test_diag_w4.py -q -m diag > $O/diag_tests.log 2>&1; tail -2 $O/diag_tests.log
for sh in 1,256,8,8 1,64,8,8 1,256,16,16 4,256,8,8 1,64,7,7; do
  timeout 300 python tools/latency_bs1.py --shape $sh > $O/latency_$sh.txt 2>&1; tail -2 $O/latency_$sh.txt
done
NODE_TUNE_TINY=0 timeout 300 python tools/latency_bs1.py --shape 1,256,8,8 > $O/latency_tiny0.txt 2>&1; tail -2 $O/latency_tiny0.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -- python3 $GRAFT_REPO_ROOT/tools/latency_bs1.py --tols 1e-3 --iters 20 > /tmp/lt.log 2>&1
cd $GRAFT_REPO_ROOT
KS=$(find /tmp/lt -name '*kernel_stats.csv' | head -1); cp $KS $O/latency_kernel_stats.csv; head -8 $O/latency_kernel_stats.csv | cut -c1-160
