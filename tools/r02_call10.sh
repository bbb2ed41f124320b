#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c10; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $OUT/pytest.log
bash tools/ab_libs.sh $OUT/ab.log 2 base RANS_AMD_NO_PIPE=1:base RANS_AMD_DEBUG=1:base c1
B="python bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5"
pick='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(d["ms_per_step"], r["kernel_ms_avg"], r.get("wave_span_ms_avg"), r["frac"], d["bit_exact_roundtrip"], r["kernel"], d["config"]["compressed_bytes_per_symbol"])'
for c in 32768 33024 32512 16384 16640 24576 49152; do
  echo "chunk $c $(timeout 120 $B --chunk $c 2>/dev/null | python -c "$pick")"
done > $OUT/chunks.log 2>&1
ls -la $OUT
