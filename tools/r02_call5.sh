#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c5; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5"
pick='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(d["ms_per_step"], r["kernel_ms_avg"], r.get("wave_span_ms_avg"), r["frac"], d["bit_exact_roundtrip"], r["kernel"], d.get("clocks",{}).get("sclk_hz_measured"), d.get("clocks",{}).get("per_simd_clocks_per_round_of_64"), d.get("launch_ms"))'
{
for pw in 0 50 250 1000; do
  echo "prewarm $pw $(timeout 120 $B --prewarm-ms $pw --dump-launch-ms 2>/dev/null | python -c "$pick")"
done
echo "prewarm 0 steps 200 $(timeout 120 $B --prewarm-ms 0 --steps 200 --dump-launch-ms 2>/dev/null | python -c "$pick")"
for c in 16384 32768 65536; do
  echo "prewarm 250 chunk $c $(timeout 120 $B --chunk $c 2>/dev/null | python -c "$pick")"
  echo "prewarm 250 chunk $c nopipe $(RANS_AMD_NO_PIPE=1 timeout 120 $B --chunk $c 2>/dev/null | python -c "$pick")"
done
echo "r01 prewarm 250 $(RANS_AMD_LIB=$PWD/build/libexp_r01.so timeout 120 $B 2>/dev/null | python -c "$pick")"
echo "c1 prewarm 250 $(RANS_AMD_LIB=$PWD/build/libexp_c1.so timeout 120 $B 2>/dev/null | python -c "$pick")"
} > $OUT/prewarm.log 2>&1
ls -la $OUT
