#!/bin/bash
# tools/ab_lanes.sh <rounds> <time_lanes args> -- <lib variant>...   interleaved A/B of library builds on one lane config
ROUNDS=$1; shift
ARGS=""
while [ "$1" != "--" ]; do ARGS="$ARGS $1"; shift; done
shift
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    envs=""; lib=$v
    if [[ "$v" == *:* ]]; then envs="${v%%:*}"; lib="${v##*:}"; fi
    if [ "$lib" = base ]; then libenv=""; else libenv="RANS_AMD_LIB=$PWD/build/libexp_$lib.so"; fi
    echo "$v $(env $envs $libenv timeout 200 python tools/time_lanes.py $ARGS 2>&1 | grep -v amdgpu.ids)"
  done
done
