#!/bin/bash
mkdir -p gpurun_out/l4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rans64_two_way or chunked_matches or lane_kernels or corruption or unaligned or tiny" > gpurun_out/l4/tests.log 2>&1
tail -5 gpurun_out/l4/tests.log
for a in "--fmt r64 --ways 2" "--fmt word --ways 2 --sb 12" "--fmt byte --ways 2" "--fmt r64 --ways 1" "--fmt r64 --ways 8 --chunk 1024" "--fmt alias --ways 2 --sb 14"; do
  RANS_AMD_NO_R64X2=1 timeout 200 python tools/time_lanes.py $a 2>&1 | grep -v amdgpu.ids
done > gpurun_out/l4/lanes.log 2>&1
cat gpurun_out/l4/lanes.log
