cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
timeout 1500 python -m pytest tests/test_gpu_sized.py -m gpu -x -q > gpurun_out/r5o/sized.log 2>&1; echo "sized rc $?"; tail -3 gpurun_out/r5o/sized.log
timeout 600 python tools/time_slots.py --configs c4 --rounds 3 2>&1 | grep "enc \|ok\|MISMATCH" | tee gpurun_out/r5o/c4.log
