#!/bin/bash
# round 3, call 9: word encoder with 8-byte records (ds_read_b64): parity, then timing (and the bound experiments again)
mkdir -p gpurun_out
(timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r03_9_tests.log 2>&1
tail -4 gpurun_out/r03_9_tests.log
M=$PWD/ryg_rans_amd/lib/libryg_rans_amd_measure.so
N=$PWD/build/libexp_nostore.so
{
for rep in 1 2; do
RANS_AMD_LIB=$M timeout -k 5 100 python tools/time_encode.py --tag alverson8 --rounds 1 --configs word
RANS_AMD_LIB=$M RANS_AMD_WORD_NO_SMALL=1 timeout -k 5 100 python tools/time_encode.py --tag roundup8 --rounds 1 --configs word
RANS_AMD_LIB=$M RANS_AMD_ENC_DEBUG=4 timeout -k 5 100 python tools/time_encode.py --tag rec0 --rounds 1 --configs word
RANS_AMD_LIB=$N timeout -k 5 100 python tools/time_encode.py --tag nostore --rounds 1 --configs word
RANS_AMD_LIB=$N RANS_AMD_ENC_DEBUG=4 timeout -k 5 100 python tools/time_encode.py --tag both --rounds 1 --configs word
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_9_enc.log
cat gpurun_out/r03_9_enc.log
