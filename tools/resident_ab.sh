#!/bin/bash
# the resident solve (kernels_tiny_solve.hip) on the box: its tests, bs = 1 latency with it on / off (-> profiles/r05_latency_bs1.txt),
# the in-kernel timeline from the diagnostics library (-> profiles/r05_resident_timeline.txt), hand-off latencies (-> profiles/r05_pingpong.txt)
cd $GRAFT_REPO_ROOT
O=gpurun_out/resident
mkdir -p $O
rm -f $O/latency.txt
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -3 $O/tests.log
for on in 1 0; do
  echo "== NODE_TUNE_TINY_RESIDENT=$on  ($([ $on = 1 ] && echo 'one resident launch per solve: k_tiny_solve' || echo 'two launches per evaluation: k_tiny_conv_gn, + the step-control launches'))" >> $O/latency.txt
  for shape in 1,256,8,8 1,64,8,8 2,256,8,8 1,128,7,7; do
    NODE_TUNE_TINY_RESIDENT=$on timeout 300 python tools/latency_bs1.py --iters 200 --shape $shape 2>&1 | grep -v amdgpu.ids >> $O/latency.txt
  done
done
cat $O/latency.txt
NODE_HIP_DIAG=1 NODE_TUNE_TINY_STAMPS=1 timeout 300 python tools/latency_bs1.py --iters 1 --tols 1e-3 2>&1 | grep -v amdgpu.ids > $O/stamps.txt
(echo "# NODE_HIP_DIAG=1 NODE_TUNE_TINY_STAMPS=1 python tools/latency_bs1.py --iters 1 --tols 1e-3   ([1,256,8,8]: 128 workgroups)"
 echo "# workgroup 0 = the reducer of channel block 0 (also its slice-0 worker), workgroup 1 = a worker (block 0, slice 1); constant 100 MHz clock,"
 echo "# offsets from the top of the convolution's iteration; conv 2k = first convolution of evaluation k, 2k + 1 = second (its epilogue"
 echo "# holds the next stage's combine + GroupNorm-1; every sixth one the step decision).  The stamps themselves cost ~0.1 us each."
 grep "rounds" $O/stamps.txt | sort | uniq | head -2
 grep "^wg . conv" $O/stamps.txt | sort -k6 -n | awk '!seen[$2 $4]++' | head -80) > $O/timeline.txt
tail -12 $O/timeline.txt
timeout 200 ./tools/pingpong > $O/pingpong.txt 2>&1; head -9 $O/pingpong.txt
