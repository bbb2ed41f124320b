#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c9; mkdir -p $OUT
bash tools/ab_libs.sh $OUT/ab.log 2 base A B C D E F G
ls -la $OUT
