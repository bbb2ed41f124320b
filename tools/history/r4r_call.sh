# config 4 encoder: counters of the hand-written alias sub-step against the compiler's (previous build)
mkdir -p gpurun_out/r4r
export PMC_CMD="python $GRAFT_REPO_ROOT/tools/time_slots.py --configs c4 --rounds 1 --launches 3"
bash tools/pmc_kernel.sh r4r_new "k_encode<5, 1, 2>" > gpurun_out/r4r/new.log 2>&1
export RANS_AMD_LIB=$GRAFT_REPO_ROOT/build/libexp_prev.so
bash tools/pmc_kernel.sh r4r_prev "k_encode<5, 1, 2>" > gpurun_out/r4r/prev.log 2>&1
paste gpurun_out/r4r_prev_sq_summary.txt gpurun_out/r4r_new_sq_summary.txt | awk '{print $1, $3, $6}'
find gpurun_out -name "*.db" -delete
