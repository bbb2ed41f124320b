cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5t
timeout 2400 python -m pytest tests/test_gpu_stress.py -m gpu -x -q > gpurun_out/r5t/stress.log 2>&1; echo "stress rc $?"; tail -15 gpurun_out/r5t/stress.log | cut -c1-600
