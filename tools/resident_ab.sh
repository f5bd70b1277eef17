#!/bin/bash
# the resident solve (kernels_tiny_solve.hip) on the box: its tests, then bs = 1 latency with it on / off
cd $GRAFT_REPO_ROOT
O=gpurun_out/resident
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q -s > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -30 $O/tests.log
for on in 1 0; do
  echo "== NODE_TUNE_TINY_RESIDENT=$on" >> $O/latency.txt
  NODE_TUNE_TINY_RESIDENT=$on timeout 300 python tools/latency_bs1.py --iters 200 >> $O/latency.txt 2>&1
  NODE_TUNE_TINY_RESIDENT=$on timeout 300 python tools/latency_bs1.py --iters 200 --shape 1,64,8,8 >> $O/latency.txt 2>&1
done
cat $O/latency.txt
