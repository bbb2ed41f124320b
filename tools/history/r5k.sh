cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_container.py -m gpu -x -q -k "adaptive" > gpurun_out/r5k/adaptive.log 2>&1; echo "adaptive rc $?"
tail -25 gpurun_out/r5k/adaptive.log | cut -c1-400
