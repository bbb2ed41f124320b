cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
timeout 2400 python -m pytest tests/test_gpu_sized.py -m gpu -x -q > gpurun_out/r5f/sized.log 2>&1; echo "sized rc $?" 
tail -5 gpurun_out/r5f/sized.log
timeout 1200 python -m pytest tests/test_dist_gloo.py -m gpu -x -q -k "default_command" > gpurun_out/r5f/bench_test.log 2>&1; echo "bench test rc $?"
tail -5 gpurun_out/r5f/bench_test.log | cut -c1-2000
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r5f/details.json > gpurun_out/r5f/bench.out 2> gpurun_out/r5f/bench.err; echo "bench rc $?"
tail -c 3600 gpurun_out/r5f/bench.out
