#!/bin/bash
# round 3, call 3: whole GPU suite (exhaustive full-size parity, native multi-GPU example, --force-dist) + the default bench line
mkdir -p gpurun_out
(time timeout 1200 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -40) > gpurun_out/r03_3_tests.log 2>&1
(time timeout 600 python bench.py) > gpurun_out/r03_3_bench.json 2> gpurun_out/r03_3_bench.err
tail -30 gpurun_out/r03_3_tests.log; tail -c 6000 gpurun_out/r03_3_bench.json; tail -5 gpurun_out/r03_3_bench.err
