#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c12; mkdir -p $OUT
bash tools/ab_libs.sh $OUT/ab.log 3 base touch
BENCH_ARGS="--chunk 16384" bash tools/ab_libs.sh $OUT/ab16k.log 2 base touch
ls -la $OUT
