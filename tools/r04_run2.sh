cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout 1200 python -m pytest tests/test_gpu_stem.py -q -m gpu -s -k "whole_stem or no_library" > gpurun_out/r04b/stem_tests.log 2>&1
grep "stem (\|  grad\|passed\|failed\|Error" gpurun_out/r04b/stem_tests.log | tail -70
