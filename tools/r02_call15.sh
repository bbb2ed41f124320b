#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c15; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $OUT/pytest.log
ls -la $OUT
