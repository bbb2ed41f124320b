#!/bin/bash
# tools/ab_libs.sh <out-file> <rounds> <variant>...   variant = "base" | "<libname>" (build/libexp_<libname>.so) |
# "ENV=VAL:<libname or base>"  -- interleaved A/B of library builds / env knobs on the bench workload (run via gpurun)
OUTF=$1; ROUNDS=$2; shift 2
B="python bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5 ${BENCH_ARGS:-}"
pick='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(d["ms_per_step"], r["kernel_ms_avg"], r.get("wave_span_ms_avg"), r["frac"], d["bit_exact_roundtrip"], r["kernel"], d.get("clocks",{}).get("sclk_hz_measured"), d.get("clocks",{}).get("per_simd_clocks_per_round_of_64"))'
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    envs=""; lib=$v
    if [[ "$v" == *:* ]]; then envs="${v%%:*}"; lib="${v##*:}"; fi
    if [ "$lib" = base ]; then libenv=""; else libenv="RANS_AMD_LIB=$PWD/build/libexp_$lib.so"; fi
    echo "$v $(env $envs $libenv timeout 120 $B 2>/dev/null | python -c "$pick")"
  done
done > $OUTF 2>&1
