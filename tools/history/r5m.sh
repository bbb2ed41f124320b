cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
export BENCH_ARGS="--format r64 --ways 2 --chunk 512 --log2n 28 --measure"
bash tools/ab_libs.sh gpurun_out/r5m/ab.txt 4 RANS_AMD_X=1:base r64nopair
cat gpurun_out/r5m/ab.txt
