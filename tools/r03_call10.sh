#!/bin/bash
# round 3, call 10: 32 Ki against 16 Ki symbol chunks on the headline line (third box), interleaved
mkdir -p gpurun_out
for rep in 1 2 3; do
for c in 32768 16384; do
timeout -k 5 200 python bench.py --no-configs --no-cpu-baseline --chunk $c 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('chunk', d['config']['chunk_syms'], 'kernel_ms', r['kernel_ms_avg'], 'frac', r['frac'], 'ms_per_step', d['ms_per_step'], 'value', d['value'], d['bit_exact_roundtrip'])"
done; done | tee gpurun_out/r03_10_chunks.log
