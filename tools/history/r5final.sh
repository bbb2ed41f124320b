cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5final
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r5final/gpu_all.log 2>&1; echo "gpu tests rc $?"
tail -3 gpurun_out/r5final/gpu_all.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r5final/details$i.json > gpurun_out/r5final/bench$i.out 2> gpurun_out/r5final/bench$i.err; echo "bench $i rc $? bytes $(tail -n 1 gpurun_out/r5final/bench$i.out | wc -c)"
done
