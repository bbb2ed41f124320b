#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c4; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log
bash tools/ab_libs.sh $OUT/ab.log 3 base RANS_AMD_NO_PIPE=1:base c1 a3p0 a3p1 a5p0 a6p0 RANS_AMD_NO_PIPE=1:a3p0 RANS_AMD_NO_PIPE=1:a3p1
ls -la $OUT
