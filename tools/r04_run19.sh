cd $GRAFT_REPO_ROOT
python tools/host_time.py 2>&1 | grep -v amdgpu
