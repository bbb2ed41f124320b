mkdir -p gpurun_out/r4e
M=ryg_rans_amd/lib/libryg_rans_amd_measure.so
{
for sb in 12 13 14; do
  RANS_AMD_LIB=$M RANS_AMD_NO_BYTE_FUSED=1 python tools/time_byte.py --sb $sb --tag two-gather
  RANS_AMD_LIB=$M RANS_AMD_BYTE_FUSED13=1 python tools/time_byte.py --sb $sb --tag fused
  RANS_AMD_LIB=$M RANS_AMD_NO_BYTE_FUSED=1 RANS_AMD_BYTE_FMT_OUT=1 python tools/time_byte.py --sb $sb --tag two-g+bytest
  RANS_AMD_LIB=$M RANS_AMD_BYTE_FUSED13=1 RANS_AMD_BYTE_FMT_OUT=1 python tools/time_byte.py --sb $sb --tag fused+bytest
done
} > gpurun_out/r4e/byte_variants.log 2>&1
grep -v amdgpu.ids gpurun_out/r4e/byte_variants.log
python -m pytest tests/test_gpu_parity.py -q -x -k "single_stream or chunked or tiny" 2>&1 | tail -3
