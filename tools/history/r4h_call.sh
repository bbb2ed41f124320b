# counters of the slot-layout encoders (word, byte, config 4) and of the fused byte decoder
mkdir -p gpurun_out/r4h
export PMC_CMD="python $GRAFT_REPO_ROOT/tools/time_slots.py --configs word --rounds 1 --launches 3"
bash tools/pmc_kernel.sh r4h_encw "k_encode<1, 1, 2>" > gpurun_out/r4h/encw.log 2>&1
export PMC_CMD="python $GRAFT_REPO_ROOT/tools/time_slots.py --configs byte --rounds 1 --launches 3"
bash tools/pmc_kernel.sh r4h_encb "k_encode<0, 1, 2>" > gpurun_out/r4h/encb.log 2>&1
export PMC_CMD="python $GRAFT_REPO_ROOT/tools/time_slots.py --configs c4 --rounds 1 --launches 3"
bash tools/pmc_kernel.sh r4h_enc4 "k_encode<5, 1, 2>" > gpurun_out/r4h/enc4.log 2>&1
cat gpurun_out/r4h_encw_sq_summary.txt gpurun_out/r4h_encb_sq_summary.txt gpurun_out/r4h_enc4_sq_summary.txt
find gpurun_out -name "*.db" -delete
