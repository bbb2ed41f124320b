cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_container.py -m gpu -x -q -k "adaptive" 2>&1 | tail -3
timeout 600 python tools/time_adaptive.py 2>&1 | grep -v amdgpu.ids
