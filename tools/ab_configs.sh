#!/bin/bash
# tools/ab_configs.sh <out-file> <rounds> <variant>...  -- like ab_libs.sh, but prints every `configs` entry of the bench line
OUTF=$1; ROUNDS=$2; shift 2
B="python bench.py --no-cpu-baseline --steps 20 --warmup 5 ${BENCH_ARGS:-}"
pick='import json,sys
d=json.load(sys.stdin); r=d["roofline"]
print("headline", r["kernel_ms_avg"], r["frac"])
for c in d.get("configs",[]):
    if "error" in c: print("  ERROR", c); continue
    print("  %-44s dec %-22s %.4f ms %.4f | enc %.4f ms %.4f | ok %s" % (c["name"][:44], c["decode"]["kernel"], c["decode"]["ms_mean"], c["decode"]["frac"], c["encode"]["ms_mean"], c["encode"]["frac"], c["bit_exact_roundtrip"]))'
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    envs=""; lib=$v
    if [[ "$v" == *:* ]]; then envs="${v%%:*}"; lib="${v##*:}"; fi
    if [ "$lib" = base ]; then libenv=""; else libenv="RANS_AMD_LIB=$PWD/build/libexp_$lib.so"; fi
    echo "== $v"; env $envs $libenv timeout 300 $B 2>/dev/null | python -c "$pick"
  done
done > $OUTF 2>&1
