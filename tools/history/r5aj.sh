cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
bash tools/pmc.sh r05 > $OUT/r05_pmc.log 2>&1
PMC_CMD="python $GRAFT_REPO_ROOT/tools/time_slots.py --configs word --rounds 1 --launches 3" bash tools/pmc_kernel.sh r05encw3 "k_encode<1, 1, 3>" > $OUT/r05encw3_pmc.log 2>&1
PMC_CMD="python $GRAFT_REPO_ROOT/tools/time_slots.py --configs word --rounds 1 --launches 3" bash tools/pmc_kernel.sh r05encw "k_encode<1, 1, 2>" > $OUT/r05encw_pmc.log 2>&1
PMC_CMD="python $GRAFT_REPO_ROOT/tools/time_slots.py --configs c4 --rounds 1 --launches 3" bash tools/pmc_kernel.sh r05encc43 "k_encode<5, 1, 3>" > $OUT/r05encc43_pmc.log 2>&1
PMC_CMD="python $GRAFT_REPO_ROOT/tools/time_slots.py --configs c4 --rounds 1 --launches 3" bash tools/pmc_kernel.sh r05encc4 "k_encode<5, 1, 2>" > $OUT/r05encc4_pmc.log 2>&1
cat $OUT/r05_sq_summary.txt | head -5
find $OUT -name "*.db" -delete
ls $OUT/*_sq_summary.txt | grep r05
