#!/bin/bash
# final profile set of round 2: bench line, rocprofv3 kernel stats, HBM traffic passes, SQ counters (headline + all kernels)
cd ${GRAFT_REPO_ROOT:-.}
timeout 420 bash tools/profile.sh r02 > gpurun_out/r02_profile.log 2>&1
timeout 240 bash tools/pmc.sh r02 > gpurun_out/r02_pmc.log 2>&1
timeout 420 bash tools/pmc_all.sh r02all > gpurun_out/r02_pmc_all.log 2>&1
tail -3 gpurun_out/r02_pmc_all.log
ls gpurun_out | head -40
du -sh gpurun_out
