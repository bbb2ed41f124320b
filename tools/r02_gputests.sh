#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/gt; mkdir -p $OUT
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $OUT/pytest.log
ls -la $OUT
