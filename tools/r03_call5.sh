#!/bin/bash
# round 3, call 5: the scratch ring at small scale, fail-fast (every wait gives up after 30 s, and at once when another has)
mkdir -p gpurun_out
{
timeout -k 5 120 python tools/dbg_ring.py --log2n 26 --chunk 2048 --fmt word
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_5_ring.log
cat gpurun_out/r03_5_ring.log
