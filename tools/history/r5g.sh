cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
timeout 2400 python -m pytest tests/test_gpu_sized.py -m gpu -x -q > gpurun_out/r5g/sized.log 2>&1; echo "sized rc $?" 
tail -5 gpurun_out/r5g/sized.log
timeout 900 python tools/time_slots.py --configs word,byte,c4,c2,word128,r64 --rounds 2 > gpurun_out/r5g/time_slots.log 2>&1; echo "time_slots rc $?"
grep -v "dec compact\|^$" gpurun_out/r5g/time_slots.log | tail -70
