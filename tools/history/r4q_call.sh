# config 4 encoder: hand-written alias sub-step against the previous build (and, first, what its two store instructions cost)
mkdir -p gpurun_out/r4q
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slots.py tests/test_gpu_stress.py -x -q -k "alias or slot or stress or u16" > gpurun_out/r4q/tests.log 2>&1; tail -4 gpurun_out/r4q/tests.log
for v in prev new prev new; do
  if [ $v = new ]; then unset RANS_AMD_LIB; else export RANS_AMD_LIB=$GRAFT_REPO_ROOT/build/libexp_$v.so; fi
  python tools/time_slots.py --configs c4 --rounds 2 --launches 20 2>&1 | grep -E "enc|ok|MISMATCH" | sed "s/^/$v /" | tee -a gpurun_out/r4q/c4_asm.log
done
