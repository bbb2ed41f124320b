cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r5i/gpu_all.log 2>&1; echo "gpu tests rc $?"
tail -15 gpurun_out/r5i/gpu_all.log
