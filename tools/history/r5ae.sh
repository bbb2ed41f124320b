cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_slots.py tests/test_gpu_sized.py tests/test_gpu_stress.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-200
