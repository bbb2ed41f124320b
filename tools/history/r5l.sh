cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5l
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_slots.py -m gpu -x -q -k "r64 or rans64 or lane or C2 or far or 1_gib" > gpurun_out/r5l/tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r5l/tests.log
export BENCH_ARGS="--format r64 --ways 2 --chunk 512 --log2n 28"
bash tools/ab_libs.sh gpurun_out/r5l/ab.txt 3 base r64nopair
cat gpurun_out/r5l/ab.txt
bash tools/traffic_kernel.sh r5l/pair --format r64 --ways 2 --chunk 512 --log2n 28 --no-configs --placement-candidates 1 --prewarm-ms 0 2>&1 | tail -4
RANS_AMD_LIB=$PWD/build/libexp_r64nopair.so bash tools/traffic_kernel.sh r5l/nopair --format r64 --ways 2 --chunk 512 --log2n 28 --no-configs --placement-candidates 1 --prewarm-ms 0 --measure 2>&1 | tail -4
find gpurun_out/r5l -name "*.db" -delete
