mkdir -p gpurun_out/r4i
python -m pytest tests/test_gpu_parity.py tests/test_gpu_slots.py -x -q 2>&1 | tail -3
{
for i in 1 2; do
  RANS_AMD_LIB=$PWD/build/libexp_rec8.so python tools/time_slots.py --configs word,word128 --rounds 2 2>&1 | grep -E "enc|ok|MISMATCH" | sed 's/^/rec8  /'
  python tools/time_slots.py --configs word,word128 --rounds 2 2>&1 | grep -E "enc|ok|MISMATCH" | sed 's/^/rec16 /'
done
} > gpurun_out/r4i/word_rec16_ab.log 2>&1
cat gpurun_out/r4i/word_rec16_ab.log
