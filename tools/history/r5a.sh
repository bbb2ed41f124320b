set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/test_dist_gloo.py -m gpu -x -q > gpurun_out/r5a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5a/pytest.log
tail -15 gpurun_out/r5a/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5a/bench.out 2> gpurun_out/r5a/bench.err; echo "bench rc $?"
cp bench_details.json gpurun_out/r5a/
wc -c gpurun_out/r5a/bench.out
tail -c 3600 gpurun_out/r5a/bench.out
