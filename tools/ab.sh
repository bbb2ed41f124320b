#!/bin/bash
# tools/ab.sh <lib-or-"base"> ... -- interleaved A/B of alternative library builds on the bench workload
# (run via gpurun).  Each variant is run ROUNDS times in rotation; prints ms/step per run.
ROUNDS=${ROUNDS:-3}
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset RANS_AMD_LIB; else export RANS_AMD_LIB=$PWD/build/libexp_$v.so; fi
    python bench.py --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], d['roofline']['frac'], d['bit_exact_roundtrip'])"
  done
done
