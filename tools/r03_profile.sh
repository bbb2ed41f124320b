#!/bin/bash
# round 3 profile set (run via gpurun): bench line, rocprofv3 --kernel-trace --stats of the same command, FETCH/WRITE passes,
# SQ/LDS counters of the headline kernel, of k_decode_dual<alias> (config 4) and of k_decode<byte>
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $REPO
bash tools/profile.sh r03 > gpurun_out/r03_profile.log 2>&1
bash tools/pmc.sh r03 > gpurun_out/r03_pmc.log 2>&1
PMC_CMD="python $REPO/tools/time_decode.py --configs c4 --rounds 1 --launches 3" bash tools/pmc_kernel.sh r03dual k_decode_dual > gpurun_out/r03dual_pmc.log 2>&1
PMC_CMD="python $REPO/tools/time_decode.py --configs byte --rounds 1 --launches 3" bash tools/pmc_kernel.sh r03byte "k_decode<" > gpurun_out/r03byte_pmc.log 2>&1
PMC_CMD="python $REPO/tools/time_encode.py --configs word --rounds 1 --launches 3" bash tools/pmc_kernel.sh r03enc "k_encode" > gpurun_out/r03enc_pmc.log 2>&1
tail -3 gpurun_out/r03_profile.log; cat gpurun_out/r03_sq_summary.txt gpurun_out/r03dual_sq_summary.txt gpurun_out/r03byte_sq_summary.txt gpurun_out/r03enc_sq_summary.txt
find gpurun_out -name "*.db" -delete; du -sh gpurun_out
