#!/bin/bash
# round 3, call 1: GPU suite with the two-chunks-per-wave alias decoder + interleaved A/B timing
mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r03_1_tests.log 2>&1
(timeout 300 python tools/time_decode.py --configs c4,alias256 --rounds 3) > gpurun_out/r03_1_time_alias.log 2>&1
(RANS_AMD_LIB=$PWD/ryg_rans_amd/lib/libryg_rans_amd_measure.so RANS_AMD_BYTE_DUAL=1 timeout 300 python tools/time_decode.py --configs byte --rounds 3) > gpurun_out/r03_1_time_byte.log 2>&1
tail -5 gpurun_out/r03_1_tests.log; cat gpurun_out/r03_1_time_alias.log gpurun_out/r03_1_time_byte.log
