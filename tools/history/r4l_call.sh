# config 2 encoder: SDWA record addresses + no bad-symbol search for dense models, against the previous build on one box
mkdir -p gpurun_out/r4l
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slots.py -x -q -k "rans64 or r64 or lane or slot or frequency or model" > gpurun_out/r4l/tests.log 2>&1; tail -4 gpurun_out/r4l/tests.log
for r in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then export RANS_AMD_LIB=$GRAFT_REPO_ROOT/build/libexp_prev.so; else unset RANS_AMD_LIB; fi
    python tools/time_slots.py --configs c2 --rounds 2 --launches 20 2>&1 | grep -E "enc slots|dec|ok" | sed "s/^/$v /" | tee -a gpurun_out/r4l/ab.log
  done
done
