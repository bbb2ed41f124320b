#!/bin/bash
# round 3, call 2: is the unaligned ds_read_u16 what makes k_decode_dual slow?  dual kernel with two ds_read_u8 per chunk
mkdir -p gpurun_out
L=$PWD/build/libexp_dual_u8.so
(RANS_AMD_LIB=$L RANS_AMD_BYTE_DUAL=1 timeout 300 python tools/time_decode.py --configs c4,byte,alias256 --rounds 2) > gpurun_out/r03_2_time_u8.log 2>&1
cat gpurun_out/r03_2_time_u8.log
