set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
python -m pytest tests/test_gpu_deferred.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04a/tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04a/bench_cfg2.json 2> gpurun_out/r04a/bench_cfg2.err
python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/r04a/bench_cfg3.json 2> gpurun_out/r04a/bench_cfg3.err
tail -5 gpurun_out/r04a/tests.log
