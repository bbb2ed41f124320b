cd $GRAFT_REPO_ROOT
timeout 600 python tools/time_adaptive.py 2>&1 | grep -v amdgpu.ids
