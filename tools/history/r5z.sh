cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sized.py -m gpu -x -q -k "search or graph" 2>&1 | tail -12 | cut -c1-300
