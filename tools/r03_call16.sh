#!/bin/bash
# round 3, call 16: output placement trial in bench.py setup -- does the chosen buffer hold its advantage in the timed region?
mkdir -p gpurun_out
for rep in 1 2 3; do
for k in 4 1; do
timeout -k 5 200 python bench.py --no-configs --no-cpu-baseline --out-candidates $k 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('candidates', $k, 'kernel_ms', r['kernel_ms_avg'], 'frac', r['frac'], 'value', d['value'], d['output_placement'], d['bit_exact_roundtrip'])"
done; done | tee gpurun_out/r03_16_placement.log
