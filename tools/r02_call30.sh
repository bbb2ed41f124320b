#!/bin/bash
# full GPU test suite + the default bench line on HEAD (after the fused encoder placement)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c30; mkdir -p $OUT
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
