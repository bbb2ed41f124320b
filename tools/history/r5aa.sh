cd $GRAFT_REPO_ROOT
export RANS_AMD_TRACE=/tmp/trace.txt
for v in prio2 prio3; do echo "== $v"; RANS_AMD_LIB=$PWD/build/libexp_$v.so timeout 300 python tools/wave_tail.py 32768 2>&1 | grep -v amdgpu.ids; done
