#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c7; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5"
pick='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(d["ms_per_step"], r["kernel_ms_avg"], r.get("wave_span_ms_avg"), r["frac"], r["kernel"], d.get("clocks",{}).get("sclk_hz_measured"), d.get("clocks",{}).get("per_simd_clocks_per_round_of_64"), d.get("clocks",{}).get("instrumented_launch_ms"))'
{
for r in 1 2; do
echo "normal $(timeout 120 $B 2>/dev/null | python -c "$pick")"
echo "dropstores $(RANS_AMD_DEBUG=1 timeout 120 $B 2>/dev/null | python -c "$pick")"
echo "samechunk $(timeout 120 $B --debug-same-chunk 2>/dev/null | python -c "$pick")"
echo "samechunk+dropstores $(RANS_AMD_DEBUG=1 timeout 120 $B --debug-same-chunk 2>/dev/null | python -c "$pick")"
echo "r01 normal $(RANS_AMD_LIB=$PWD/build/libexp_r01.so timeout 120 $B 2>/dev/null | python -c "$pick")"
echo "r01 samechunk $(RANS_AMD_LIB=$PWD/build/libexp_r01.so timeout 120 $B --debug-same-chunk 2>/dev/null | python -c "$pick")"
done
} > $OUT/membound.log 2>&1
cd /tmp; rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|SQ_[A-Z_0-9]*\|TA_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u > $GRAFT_REPO_ROOT/$OUT/counters.txt
ls -la $GRAFT_REPO_ROOT/$OUT
