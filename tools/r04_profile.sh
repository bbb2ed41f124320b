#!/bin/bash
# round 4 profile set (run via gpurun): the bench line, rocprofv3 --kernel-trace --stats of the same command, FETCH / WRITE
# passes of the headline decoder, the same two passes over the encoders in both layouts (tools/time_slots.py), SQ / LDS / TA
# counters of the headline kernel, the slot-record byte decoder, k_decode<byte> and the slot-layout word and byte encoders
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $REPO
OUT=$REPO/gpurun_out
mkdir -p $OUT
bash tools/profile.sh r04 > $OUT/r04_profile.log 2>&1
bash tools/pmc.sh r04 > $OUT/r04_pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp
SLOTS="python $REPO/tools/time_slots.py --configs word,byte,c4,c2 --rounds 1 --launches 3"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/r04enc_pmc_fetch" -o pmc -- $SLOTS > "$OUT/r04enc_pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OUT/r04enc_pmc_write" -o pmc -- $SLOTS > "$OUT/r04enc_pmc_write.log" 2>&1
cd $REPO
PMC_CMD="python $REPO/tools/time_slots.py --configs word --rounds 1 --launches 3" bash tools/pmc_kernel.sh r04encw "k_encode<1, 1, 2>" > $OUT/r04encw_pmc.log 2>&1
PMC_CMD="python $REPO/tools/time_slots.py --configs byte --rounds 1 --launches 3" bash tools/pmc_kernel.sh r04encb "k_encode<0, 1, 2>" > $OUT/r04encb_pmc.log 2>&1
PMC_CMD="python $REPO/tools/time_byte.py --sb 12 --rounds 1" bash tools/pmc_kernel.sh r04bytef "k_decode<11" > $OUT/r04bytef_pmc.log 2>&1
PMC_CMD="python $REPO/tools/time_byte.py --sb 14 --rounds 1" bash tools/pmc_kernel.sh r04byte "k_decode<0" > $OUT/r04byte_pmc.log 2>&1
tail -3 $OUT/r04_profile.log
cat $OUT/r04_sq_summary.txt $OUT/r04encw_sq_summary.txt $OUT/r04encb_sq_summary.txt $OUT/r04bytef_sq_summary.txt $OUT/r04byte_sq_summary.txt
find $OUT -name "*.db" -delete; du -sh $OUT
