#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c13; mkdir -p $OUT
bash tools/ab_configs.sh $OUT/configs.log 2 base lanesnt c1
ls -la $OUT
