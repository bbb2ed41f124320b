#!/bin/bash
mkdir -p gpurun_out/f1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/f1/tests.log 2>&1
tail -5 gpurun_out/f1/tests.log
for v in "" "RANS_AMD_ENCODE_UNFUSED=1"; do
  echo "== [$v]"
  env $v timeout 300 python tools/time_lanes.py --fmt word --ways 64 --chunk 32768 --log2n 30 --sb 12 --encode 2>&1 | grep -v amdgpu.ids
  env $v timeout 300 python tools/time_lanes.py --fmt byte --ways 64 --chunk 32768 --log2n 30 --sb 14 --encode 2>&1 | grep -v amdgpu.ids
  env $v timeout 300 python tools/time_lanes.py --fmt alias --ways 64 --chunk 32768 --log2n 30 --sb 16 --encode 2>&1 | grep -v amdgpu.ids
done > gpurun_out/f1/enc.log 2>&1
cat gpurun_out/f1/enc.log
