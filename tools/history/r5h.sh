cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
export RANS_AMD_LIB=$GRAFT_REPO_ROOT/ryg_rans_amd/lib/libryg_rans_amd_measure.so
for lds in 0 53000 81920 122880; do
  for chunk in 16384 32768; do
    echo "=== RANS_AMD_ENC_LDS_MIN=$lds chunk $chunk"
    RANS_AMD_ENC_LDS_MIN=$lds timeout 300 python tools/time_slots.py --configs word,byte --rounds 1 --chunk $chunk 2>&1 | grep "enc slots\|enc tight"
  done
done > gpurun_out/r5h/occupancy.log 2>&1
cat gpurun_out/r5h/occupancy.log
