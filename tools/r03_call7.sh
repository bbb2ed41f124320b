#!/bin/bash
# round 3, call 7: word encoder with ds_read_b128 records, SDWA shift count and Alverson reciprocals for small-frequency
# models; byte encoder with the SDWA shift -- parity first, then timing; bench line with 16 Ki and 32 Ki chunks
mkdir -p gpurun_out
(timeout -k 5 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r03_7_tests.log 2>&1
tail -4 gpurun_out/r03_7_tests.log
M=$PWD/ryg_rans_amd/lib/libryg_rans_amd_measure.so
{
for rep in 1 2; do
RANS_AMD_LIB=$M RANS_AMD_WORD_NO_SMALL=1 timeout -k 5 100 python tools/time_encode.py --tag roundup --rounds 1
RANS_AMD_LIB=$M timeout -k 5 100 python tools/time_encode.py --tag alverson --rounds 1
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_7_enc.log
cat gpurun_out/r03_7_enc.log
for rep in 1 2; do
for c in 32768 16384; do
timeout -k 5 200 python bench.py --no-configs --no-cpu-baseline --chunk $c 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('chunk', d['config']['chunk_syms'], 'kernel_ms', r['kernel_ms_avg'], 'frac', r['frac'], 'ms_per_step', d['ms_per_step'], d['bit_exact_roundtrip'])"
done; done | tee gpurun_out/r03_7_chunks.log
