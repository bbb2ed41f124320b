cd $GRAFT_REPO_ROOT
bash tools/r05_profile.sh > gpurun_out/r05_profile.log 2>&1
tail -5 gpurun_out/r05_profile.log | cut -c1-300
ls gpurun_out/r05p | head -50
