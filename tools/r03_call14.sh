#!/bin/bash
# round 3, call 14: config 4 with 2, 3 and 4 chunks per wave (k_decode_dual / k_decode_multi with half-size windows)
mkdir -p gpurun_out
M=$PWD/ryg_rans_amd/lib/libryg_rans_amd_measure.so
{
for rep in 1 2; do
for c in 2 3 4; do
echo "## chunks per wave $c"
RANS_AMD_LIB=$M RANS_AMD_CHUNKS_PER_WAVE=$c timeout -k 5 100 python tools/time_decode.py --configs c4 --rounds 1
done; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_14_multi.log
cat gpurun_out/r03_14_multi.log
