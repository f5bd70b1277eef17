cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04l
python tools/dp_straggler.py --steps 50 --config 2 > gpurun_out/r04l/dp_straggler_cfg2.txt 2>&1; cat gpurun_out/r04l/dp_straggler_cfg2.txt
python tools/dp_straggler.py --steps 30 --config 3 > gpurun_out/r04l/dp_straggler_cfg3.txt 2>&1; cat gpurun_out/r04l/dp_straggler_cfg3.txt
