set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 2400 python -m pytest tests/test_gpu_sized.py -m gpu -x -q > gpurun_out/r5b/sized.log 2>&1; echo "sized rc $?" 
tail -30 gpurun_out/r5b/sized.log
timeout 1200 python -m pytest tests/test_gpu_slots.py tests/test_dist_gloo.py -m gpu -x -q > gpurun_out/r5b/slots.log 2>&1; echo "slots rc $?"
tail -8 gpurun_out/r5b/slots.log
