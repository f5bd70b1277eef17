#!/bin/bash
# the resident solve (kernels_tiny_solve.hip) on the box: its tests, bs = 1 latency with it on / off, the in-kernel timeline (diag library)
cd $GRAFT_REPO_ROOT
O=gpurun_out/resident
mkdir -p $O
rm -f $O/latency.txt
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -5 $O/tests.log
for on in 1 0; do
  echo "== NODE_TUNE_TINY_RESIDENT=$on" >> $O/latency.txt
  NODE_TUNE_TINY_RESIDENT=$on timeout 300 python tools/latency_bs1.py --iters 200 >> $O/latency.txt 2>&1
  NODE_TUNE_TINY_RESIDENT=$on timeout 300 python tools/latency_bs1.py --iters 200 --shape 1,64,8,8 >> $O/latency.txt 2>&1
done
grep -v amdgpu.ids $O/latency.txt
NODE_HIP_DIAG=1 NODE_TUNE_TINY_STAMPS=1 timeout 300 python tools/latency_bs1.py --iters 1 --tols 1e-3 > $O/stamps.txt 2>&1
grep "rounds" $O/stamps.txt; grep "^wg . conv 2[0-7]" $O/stamps.txt | sort -k6 -n | tail -12
