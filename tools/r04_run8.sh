cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
bash tools/pmc_w4.sh gpurun_out/r04h/pmc_w4_cfg2.json 128,256,8 > gpurun_out/r04h/pmc_w4_cfg2.txt 2>&1
cat gpurun_out/r04h/pmc_w4_cfg2.txt | tail -30
