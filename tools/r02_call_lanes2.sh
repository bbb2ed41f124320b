#!/bin/bash
mkdir -p gpurun_out/l2
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rans64_two_way or chunked_matches or lane_kernels or corruption" > gpurun_out/l2/tests.log 2>&1
tail -15 gpurun_out/l2/tests.log
for v in "" "RANS_AMD_DEBUG=4" "RANS_AMD_NO_R64X2=1"; do
  echo "== [$v]"; env $v timeout 200 python tools/time_lanes.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/l2/lanes.log 2>&1
env timeout 200 python tools/time_lanes.py --chunk 4096 2>&1 | grep -v amdgpu.ids >> gpurun_out/l2/lanes.log
env timeout 200 python tools/time_lanes.py --chunk 1024 2>&1 | grep -v amdgpu.ids >> gpurun_out/l2/lanes.log
cat gpurun_out/l2/lanes.log
