cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python3 $R/tools/stem_time.py --run > /tmp/st.log 2>&1 || tail -20 /tmp/st.log
cd $R
python3 tools/stem_time.py --report /tmp/st > gpurun_out/r04d/stem_time.txt
cat gpurun_out/r04d/stem_time.txt
