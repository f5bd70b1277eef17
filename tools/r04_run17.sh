cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04q
for pol in "0.15 16 8" "0.10 16 8" "0.05 8 8" "0.0 0 8" "0.25 32 8"; do
set -- $pol
echo "FRAGILE=$1 QUIET=$2 HIST=$3"
NODE_DEFERRED_FRAGILE=$1 NODE_DEFERRED_QUIET=$2 NODE_DEFERRED_HIST=$3 python tools/deferred_soak.py --steps 300 --config 3 2>&1 | grep "deferred" | cut -c1-260
done > gpurun_out/r04q/policy_cfg3.txt
cat gpurun_out/r04q/policy_cfg3.txt
python tools/deferred_soak.py --steps 300 --config 3 2>&1 | grep -v amdgpu > gpurun_out/r04q/soak_cfg3.txt; cat gpurun_out/r04q/soak_cfg3.txt | cut -c1-300
