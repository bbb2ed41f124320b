cd $GRAFT_REPO_ROOT
export RANS_AMD_TRACE=/tmp/trace.txt RANS_AMD_LIB=$PWD/ryg_rans_amd/lib/libryg_rans_amd_measure.so
timeout 300 python tools/wave_tail.py 32768 2>&1 | grep -v amdgpu.ids
