# C2 decoder: occupancy scan (waves per CU) of both table forms, then counters of the two shipped shapes
mkdir -p gpurun_out/r4f
M=ryg_rans_amd/lib/libryg_rans_amd_measure.so
{
for w in 4 6 8 10 11; do
  echo "packed waves $w: $(RANS_AMD_LIB=$M RANS_AMD_R64_WAVES=$w python tools/time_lanes.py --steps 30 2>&1 | grep decode)"
done
for w in 4 6 8 10 12 14 16; do
  echo "two-gather waves $w: $(RANS_AMD_LIB=$M RANS_AMD_NO_R64_PACKED=1 RANS_AMD_R64_WAVES=$w python tools/time_lanes.py --steps 30 2>&1 | grep decode)"
done
} > gpurun_out/r4f/c2_occupancy.log 2>&1
cat gpurun_out/r4f/c2_occupancy.log
export PMC_CMD="python $GRAFT_REPO_ROOT/tools/time_lanes.py --steps 10"
bash tools/pmc_kernel.sh r4f_c2_packed k_decode_lanes_r64x2 > gpurun_out/r4f/pmc_packed.log 2>&1
export RANS_AMD_LIB=$GRAFT_REPO_ROOT/$M RANS_AMD_NO_R64_PACKED=1
bash tools/pmc_kernel.sh r4f_c2_twogather k_decode_lanes_r64x2 > gpurun_out/r4f/pmc_twogather.log 2>&1
unset RANS_AMD_LIB RANS_AMD_NO_R64_PACKED
cat gpurun_out/r4f_c2_packed_sq_summary.txt gpurun_out/r4f_c2_twogather_sq_summary.txt
