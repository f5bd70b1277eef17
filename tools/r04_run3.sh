cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout 900 python -m pytest tests/test_gpu_stem.py -q -m gpu -s > gpurun_out/r04c/stem_tests.log 2>&1
grep "stem (\|passed\|failed\|Error" gpurun_out/r04c/stem_tests.log | tail -12
python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > gpurun_out/r04c/bench_cfg2.json 2> gpurun_out/r04c/bench_cfg2.err
tail -3 gpurun_out/r04c/bench_cfg2.err
STEPS=12 WARM=3 TOP=70 bash tools/profile_bench.sh gpurun_out/r04c/cfg2 --no-pmc --no-fresh > gpurun_out/r04c/profile.log 2>&1
head -75 gpurun_out/r04c/cfg2_steps.txt
