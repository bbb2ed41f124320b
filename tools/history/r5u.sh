cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5u
echo "== shipped library"; timeout 300 python tools/time_slots.py --configs c4 --rounds 2 2>&1 | grep "enc slots\|enc tight\|ok\|MISMATCH"
echo "== stream stores of the unstaged byte-stream renormalisation DROPPED (output wrong by construction)"
RANS_AMD_LIB=$PWD/build/libexp_c4nostore.so timeout 300 python tools/time_slots.py --configs c4 --rounds 2 2>&1 | grep "enc slots\|enc tight\|ok\|MISMATCH\|failed"
