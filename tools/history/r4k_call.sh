# word / byte encoders: v_cmpx + word-unit pointer (word), mirrored lanes (byte) against the previous build, same box
mkdir -p gpurun_out/r4k
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slots.py -x -q -k "encode or slot or single_stream or ragged or byte or word" > gpurun_out/r4k/tests.log 2>&1; tail -4 gpurun_out/r4k/tests.log
for r in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then export RANS_AMD_LIB=$GRAFT_REPO_ROOT/build/libexp_prev.so; else unset RANS_AMD_LIB; fi
    python tools/time_slots.py --configs word,byte,word128 --rounds 2 --launches 20 2>&1 | grep -E "enc slots|ok" | sed "s/^/$v /" | tee -a gpurun_out/r4k/ab.log
  done
done
