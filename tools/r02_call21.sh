#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
bash tools/profile.sh r02 > gpurun_out/r02_profile.log 2>&1
bash tools/pmc.sh r02 > gpurun_out/r02_pmc.log 2>&1
ls -la gpurun_out | tail -30
