cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
timeout 2400 python -m pytest tests/test_gpu_scale.py -m gpu -x -q > gpurun_out/r5q/scale.log 2>&1; echo "scale rc $?"; tail -12 gpurun_out/r5q/scale.log | cut -c1-300
