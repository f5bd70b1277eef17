cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_stem.py tests/test_gpu_round2.py tests/test_gpu_w4.py -q -m gpu -k "stem or 32x32 or matches_f2" -s > gpurun_out/r04g/tests.log 2>&1
grep "32, 32\|16, 32\|32x32\|vs oracle\|passed\|failed\|Error\|error" gpurun_out/r04g/tests.log | tail -40
bash tools/w4_budget.sh gpurun_out/r04g/w4_budget.txt > /dev/null 2>&1
cat gpurun_out/r04g/w4_budget.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python3 $R/tools/stem_time.py --run > /tmp/st.log 2>&1 || tail -20 /tmp/st.log
cd $R
python3 tools/stem_time.py --report /tmp/st --per-iter 32 > gpurun_out/r04g/stem_time.txt
cat gpurun_out/r04g/stem_time.txt
