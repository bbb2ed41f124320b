#!/bin/bash
# one GPU call: 64-bit VALU / LDS census + memory-path experiments on the staged lane decoder (config 2)
mkdir -p gpurun_out/l1
timeout 200 build/ubench3 > gpurun_out/l1/ubench3.log 2>&1
for v in "" "RANS_AMD_DEBUG=2" "RANS_AMD_DEBUG=4" "RANS_AMD_DEBUG=6"; do
  echo "== [$v]"; env $v timeout 200 python tools/time_lanes.py --no-check
done > gpurun_out/l1/lanes.log 2>&1
env timeout 200 python tools/time_lanes.py --chunk 4096 >> gpurun_out/l1/lanes.log 2>&1
cat gpurun_out/l1/ubench3.log gpurun_out/l1/lanes.log
