#!/bin/bash
# round 3, call 6: scratch ring of the fused wave encoders -- parity at full size, then timing of ring sizes 1/2/3 against
# per-chunk slots; LDS slot-gather microbenchmark with its conflict counters
mkdir -p gpurun_out
(timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "ring or full_size or adaptive or dual" 2>&1 | tail -8) > gpurun_out/r03_6_tests.log 2>&1
tail -4 gpurun_out/r03_6_tests.log
M=$PWD/ryg_rans_amd/lib/libryg_rans_amd_measure.so
{
for rep in 1 2; do
RANS_AMD_LIB=$M RANS_AMD_ENC_NO_RING=1 timeout -k 5 100 python tools/time_encode.py --tag noring --rounds 1
RANS_AMD_LIB=$M timeout -k 5 100 python tools/time_encode.py --tag ring2 --rounds 1
RANS_AMD_LIB=$PWD/build/libexp_ring1.so timeout -k 5 100 python tools/time_encode.py --tag ring1 --rounds 1
RANS_AMD_LIB=$PWD/build/libexp_ring3.so timeout -k 5 100 python tools/time_encode.py --tag ring3 --rounds 1
done
RANS_AMD_LIB=$M RANS_AMD_ENC_NO_RING=1 timeout -k 5 100 python tools/time_encode.py --tag noring --rounds 1 --chunk 16384 --configs word
RANS_AMD_LIB=$M timeout -k 5 100 python tools/time_encode.py --tag ring2 --rounds 1 --chunk 16384 --configs word
RANS_AMD_LIB=$PWD/build/libexp_ring1.so timeout -k 5 100 python tools/time_encode.py --tag ring1 --rounds 1 --chunk 16384 --configs word
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_6_enc.log
cat gpurun_out/r03_6_enc.log
timeout -k 5 60 build/ubench_slot > gpurun_out/r03_6_ubench_slot.log 2>&1; cat gpurun_out/r03_6_ubench_slot.log
cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r03_6_ubench_pmc -o ub -- $GRAFT_REPO_ROOT/build/ubench_slot > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
ls -R gpurun_out/r03_6_ubench_pmc | head -20
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; nproc; lscpu | grep -E "Model name|Socket|Core|Thread" 
