cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5y
timeout 1500 python -m pytest tests/test_gpu_sized.py tests/test_gpu_stress.py -m gpu -x -q > gpurun_out/r5y/t.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r5y/t.log
timeout 600 python tools/time_slots.py --configs c4,word128,word256 --rounds 2 2>&1 | grep "enc slots\|enc tight\|ok \|MISMATCH" | tee gpurun_out/r5y/time.log
