cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5r
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r5r/gpu_all.log 2>&1; echo "gpu tests rc $?"
tail -6 gpurun_out/r5r/gpu_all.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
