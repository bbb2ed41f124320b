#!/bin/bash
# round 3, call 13: the 4096-symbol alias encoder with fused placement (mailbox in global memory): parity, then A/B
mkdir -p gpurun_out
(timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r03_13_tests.log 2>&1
tail -4 gpurun_out/r03_13_tests.log
{
for rep in 1 2; do
timeout -k 5 100 python tools/time_encode.py --configs c4 --rounds 1 --tag fused --fused 1
timeout -k 5 100 python tools/time_encode.py --configs c4 --rounds 1 --tag three-kernel --fused 0
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_13_c4enc.log
cat gpurun_out/r03_13_c4enc.log
