#!/bin/bash
# final profile set of round 2: GPU tests + smoke, bench line, rocprofv3 kernel stats, HBM traffic passes, SQ counters
cd ${GRAFT_REPO_ROOT:-.}
( timeout 280 python -m pytest tests -m gpu -q --timeout 120 2>&1 | tail -4 ) > gpurun_out/r02_final_pytest.log
tail -2 gpurun_out/r02_final_pytest.log
( timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1 ) > gpurun_out/r02_final_smoke.log; cat gpurun_out/r02_final_smoke.log
timeout 420 bash tools/profile.sh r02 > gpurun_out/r02_profile.log 2>&1
timeout 240 bash tools/pmc.sh r02 > gpurun_out/r02_pmc.log 2>&1
timeout 420 bash tools/pmc_all.sh r02all > gpurun_out/r02_pmc_all.log 2>&1
tail -2 gpurun_out/r02_pmc_all.log | cut -c1-200
du -sh gpurun_out
