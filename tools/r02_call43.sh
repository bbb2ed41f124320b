#!/bin/bash
# A/B of fused-encoder build knobs (block size, copy depth) on the configs line
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c43; mkdir -p $OUT
timeout 400 bash tools/ab_configs.sh $OUT/ab.log 2 base t1024 d4 d16
grep -E "^==|C3 word|byte format" $OUT/ab.log | cut -c1-160
