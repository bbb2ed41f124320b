cd $GRAFT_REPO_ROOT
export RANS_AMD_LIB=$PWD/ryg_rans_amd/lib/libryg_rans_amd_measure.so RANS_AMD_TRACE=/tmp/trace.txt
for c in 32768 16384; do timeout 300 python tools/wave_tail.py $c 2>&1 | grep -v amdgpu.ids; done
