cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout 1500 python -m pytest tests/ -q -m gpu --durations=25 > gpurun_out/suite/gpu_tests.log 2>&1
tail -60 gpurun_out/suite/gpu_tests.log
