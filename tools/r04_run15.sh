cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04o
STEPS=12 WARM=3 TOP=70 bash tools/profile_bench.sh gpurun_out/r04o/cfg2 --no-pmc --no-fresh > gpurun_out/r04o/profile.log 2>&1
head -64 gpurun_out/r04o/cfg2_steps.txt
