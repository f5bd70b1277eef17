cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04p
timeout 900 python -m pytest tests/test_gpu_round2.py -q -m gpu -s -k "w4_fp64_arbiter and 1e-05" > gpurun_out/r04p/arbiter.log 2>&1
grep "F(4x4\|theta tensor\|passed\|failed\|^E " gpurun_out/r04p/arbiter.log | tail -20
