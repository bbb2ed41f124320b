#!/bin/bash
# A/B: encoders that help the copier at the end (base) vs not (nohelp), whole configs line; lane tests on the current build
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c33; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) > $OUT/pytest.log
tail -2 $OUT/pytest.log
bash tools/ab_configs.sh $OUT/ab.log 2 base nohelp
cat $OUT/ab.log
