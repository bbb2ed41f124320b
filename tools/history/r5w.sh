cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5w
export BENCH_ARGS="--measure"
bash tools/ab_libs.sh gpurun_out/r5w/ab.txt 3 RANS_AMD_X=1:base prio
cat gpurun_out/r5w/ab.txt
export RANS_AMD_LIB=$PWD/build/libexp_priom.so RANS_AMD_TRACE=/tmp/trace.txt
timeout 300 python tools/wave_tail.py 32768 2>&1 | grep -v amdgpu.ids
