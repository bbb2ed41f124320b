#!/bin/bash
mkdir -p gpurun_out/l3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rans64_two_way or chunked_matches or lane_kernels or corruption" > gpurun_out/l3/tests.log 2>&1
tail -5 gpurun_out/l3/tests.log
PMC_CMD="python $PWD/tools/time_lanes.py --steps 3 --no-check" bash tools/pmc_kernel.sh l3 k_decode_lanes_r64x2 > gpurun_out/l3/pmc.log 2>&1
tail -32 gpurun_out/l3/pmc.log
