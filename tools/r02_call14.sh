#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c14; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5"
pick='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(d["ms_per_step"], r["kernel_ms_avg"], r.get("wave_span_ms_avg"), r["frac"], d["bit_exact_roundtrip"])'
{
for o in 0 4096 65536 1048576 2097152 2101248 33554432 16777216; do
  echo "out+$o $(timeout 120 $B --debug-out-offset $o 2>/dev/null | python -c "$pick")"
done
for o in 4096 65536 1048576 2101248 16777216; do
  echo "cont+$o $(timeout 120 $B --debug-cont-offset $o 2>/dev/null | python -c "$pick")"
done
} > $OUT/offsets.log 2>&1
ls -la $OUT
