cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_gpu or placement" 2>&1 | tail -2
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 build/multi_gpu 1 30 20 2>&1 | grep -E "placement|^\{" | cut -c1-400
