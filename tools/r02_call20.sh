#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c20; mkdir -p $OUT
bash tools/ab_configs.sh $OUT/configs.log 2 base compactnt
ls -la $OUT
