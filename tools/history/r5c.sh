set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 2400 python -m pytest tests/test_gpu_sized.py -m gpu -x -q > gpurun_out/r5c/sized.log 2>&1; echo "sized rc $?" 
tail -30 gpurun_out/r5c/sized.log
timeout 1200 python -m pytest tests/test_dist_gloo.py -m gpu -x -q -k "default_command" > gpurun_out/r5c/bench_test.log 2>&1; echo "bench test rc $?"
tail -30 gpurun_out/r5c/bench_test.log | cut -c1-3000
