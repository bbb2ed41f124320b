#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c6; mkdir -p $OUT
bash tools/ab_libs.sh $OUT/ab.log 3 base RANS_AMD_NO_PIPE=1:base c1 r01 RANS_AMD_NO_SPAN=1:base "RANS_AMD_NO_SPAN=1 RANS_AMD_NO_PIPE=1:base" RANS_AMD_NO_PIPE=1:oldtemps RANS_AMD_NO_PIPE=1:oldopen
ls -la $OUT
