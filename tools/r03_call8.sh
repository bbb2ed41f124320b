#!/bin/bash
# round 3, call 8: where is the word encoder bound?  (a) every lane reads record 0 (no LDS bank conflicts), (b) the stream
# stores dropped, (c) both -- all wrong by construction, measure build only
mkdir -p gpurun_out
M=$PWD/ryg_rans_amd/lib/libryg_rans_amd_measure.so
N=$PWD/build/libexp_nostore.so
{
for rep in 1 2; do
RANS_AMD_LIB=$M timeout -k 5 100 python tools/time_encode.py --tag base --rounds 1 --configs word
RANS_AMD_LIB=$M RANS_AMD_ENC_DEBUG=4 timeout -k 5 100 python tools/time_encode.py --tag rec0 --rounds 1 --configs word
RANS_AMD_LIB=$N timeout -k 5 100 python tools/time_encode.py --tag nostore --rounds 1 --configs word
RANS_AMD_LIB=$N RANS_AMD_ENC_DEBUG=4 timeout -k 5 100 python tools/time_encode.py --tag both --rounds 1 --configs word
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_8_enc_bound.log
cat gpurun_out/r03_8_enc_bound.log
