#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c8; mkdir -p $OUT
bash tools/ab_libs.sh $OUT/ab.log 2 base nt sc1 sc0sc1 ntsc1 sc0 r01
ls -la $OUT
