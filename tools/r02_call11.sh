#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c11; mkdir -p $OUT
bash tools/ab_libs.sh $OUT/ab.log 2 base m2 ahead20 ahead6
B="python bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5"
pick='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(d["ms_per_step"], r["kernel_ms_avg"], r.get("wave_span_ms_avg"), r["frac"], r["kernel"])'
for k in 64 512 4096; do
  echo "samechunk $k $(timeout 120 $B --debug-same-chunk $k 2>/dev/null | python -c "$pick")"
  echo "samechunk $k dropstores $(RANS_AMD_DEBUG=1 timeout 120 $B --debug-same-chunk $k 2>/dev/null | python -c "$pick")"
done > $OUT/reads.log 2>&1
ls -la $OUT
