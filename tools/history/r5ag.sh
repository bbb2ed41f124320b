cd $GRAFT_REPO_ROOT
timeout 900 python tools/config_sweep.py 2>&1 | grep -v amdgpu.ids | tail -60
