#!/bin/bash
# tools/ab_sweep.sh "<config filter>" <lib-or-"base"> ... -- interleaved A/B of library builds on one
# config_sweep line (run via gpurun); variants are build/libexp_<name>.so, "base" = the in-tree library.
FILTER=$1; shift
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset RANS_AMD_LIB; else export RANS_AMD_LIB=$PWD/build/libexp_$v.so; fi
    echo "$v $(python tools/config_sweep.py --only "$FILTER" 2>/dev/null | tail -1)"
  done
done
