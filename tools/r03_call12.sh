#!/bin/bash
# round 3, call 12: config 2's lane decoder with 4-byte packed slot records (one gather per symbol): parity, then A/B
mkdir -p gpurun_out
(timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r03_12_tests.log 2>&1
tail -4 gpurun_out/r03_12_tests.log
M=$PWD/ryg_rans_amd/lib/libryg_rans_amd_measure.so
{
for rep in 1 2; do
RANS_AMD_LIB=$M timeout -k 5 100 python tools/time_decode.py --configs c2 --rounds 1 --mib 256
RANS_AMD_LIB=$M RANS_AMD_NO_R64_PACKED=1 timeout -k 5 100 python tools/time_decode.py --configs c2 --rounds 1 --mib 256
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_12_c2.log
cat gpurun_out/r03_12_c2.log
